// smr_fused_compose.h — wave B of the hot path: k_compose_output (included by smr_fused.hip only).
//
// LayoutShader::render + RgbaToYuvConverter / RgbaToNv12Converter in one launch.  A 256-thread workgroup
// owns a 128x16 pixel tile.  The layout list (copied once into LDS) is classified per tile with the exact
// "solid region" test of smr_layout_dev.h (one thread per layout): touch / solid / start = last solid layer whose
// base value is opaque — every earlier layer is overwritten by it (dst * (1 - 1) == 0 exactly).
//   * copy tiles — the start layer is a 1:1 texel-aligned opaque texture (the resampled video tile) or an opaque
//     colour and nothing above it touches the tile: each thread moves a 4x2 pixel block straight from the
//     source texels to Y'CbCr.  No tables, no blending arithmetic.
//   * general tiles — one pixel per thread, eight sweeps of the tile: per-pixel start search (topmost opaque
//     solid layer at that pixel), then the same per-pixel code as the general compositor; results go to an
//     8 KB RGBA8 tile in LDS, from which the 4x2 Y'CbCr conversion runs.  (One inlined copy of the compositor
//     keeps the kernel inside the instruction cache — eight unrolled copies did not.)
#pragma once

#include <type_traits>

#include "smr_convert_dev.h"
#include "smr_layout_dev.h"

namespace {

// row * pitch + offset as one full-rate v_mad_u32_u24 (rows and pitches are below 2^24, a surface below 4 GiB); the 64-bit form is a
// quarter-rate 32 x 32 multiply per address
__device__ __forceinline__ u32 b_off(int row, u32 pitch, int col_bytes) { return (u32)__umul24((u32)row, pitch) + (u32)col_bytes; }

constexpr int B_TILE_W = 128, B_TILE_H = 16;  // pixels; 32 x 8 threads, one 4x2 pixel block each
constexpr int B_MAX_LAYOUTS = 48;             // LDS-resident layout list (larger lists take the general compositor)
constexpr int B_MAX_MASKS = 96;
static_assert(sizeof(DevLayout) % 16 == 0 && sizeof(DevMask) % 16 == 0, "LDS copies move 16 B words");

// How many workgroups the compositor's band list gets.  The tiles that need compositing are latency-bound (layer records through
// scalar registers, one pixel per thread); started last they would be the tail of the kernel, so the first workgroups of the grid
// take them band by band from the classifier's list (TileList).  The host sizes that part of the grid from the list's length once
// it has come back from the device; for a list that is new this frame (a scene in transition) from this prediction — an upper
// bound: a workgroup beyond the list's end returns at once, a listed tile beyond the prediction is composited by the workgroup that
// owns it, band after band.
// A tile that needs compositing is rendered band by band, a workgroup per band, each starting from the classifier's per-band record (touching
// layers, start layer) instead of classifying again; how many bands a tile is cut into is the classifier's decision (B_AREA_BANDS / B_EDGE_BANDS
// below).  Round 3 measured the first compositor with four bands for every tile at 23.2 us on configs[2] and 48.3 us on configs[4], with eight at
// 25.0 / 57.1 us (profiles/r03_compose_ab.txt).
#ifndef SMR_COMPOSE_MIN_WAVES
#define SMR_COMPOSE_MIN_WAVES 6
#endif
// waves per SIMD the register allocation must leave room for.  Round 3 measured the first compositor (~106 VGPRs unconstrained) at 5 / 6 waves
// (96 / 80 VGPRs, scratch in the compositing path): slower ALONE on configs[2], faster on configs[4] (profiles/r03_compose_occupancy.txt).
// Round 6: with two frames in flight the kernel runs beside the other lane's resampler, whose two waves per SIMD leave 160 registers — at 96
// the frame gained 7 %.  The list compositor (compose_full below) fits 79 registers without scratch: six waves, and two of them beside the
// resampler — configs[4] + 3 % over five waves (84 registers), configs[2] unchanged (profiles/r06_sensitivity.txt section 9).
constexpr int B_MIN_WAVES = SMR_COMPOSE_MIN_WAVES;
constexpr int B_BAND_ROWS = SMR_COMPOSE_BAND_ROWS;       // rows of the workgroup's LDS pixel state: a band is at most this tall
static_assert(B_TILE_H % B_BAND_ROWS == 0 && B_BAND_ROWS >= 2, "a band fits the LDS pixel state");

// Host: tiles likely to need compositing -> their number.  A layer that can never be solid over a tile (translucent colour, texture
// with an alpha channel, rotated quad) marks its whole pixel box; every other layer marks the corner squares of its rounded rect and
// of its masks, and the bands along its edges where border / blur / fractional coordinates keep the fragment from being the base
// value.  `bitmap`: scratch, one bit per tile.
inline u32 compose_predict(const PackedLayouts &p, int tiles_x, int tiles_y, std::vector<u32> &bitmap) {
    const int tiles = tiles_x * tiles_y, words = (tiles + 31) / 32;
    bitmap.assign((size_t)words, 0u);
    auto mark = [&](float x0, float y0, float x1, float y1) {  // pixel-space box, clipped to the tile grid
        int tx0 = (int)floorf(x0 / (float)B_TILE_W), tx1 = (int)floorf(x1 / (float)B_TILE_W);
        int ty0 = (int)floorf(y0 / (float)B_TILE_H), ty1 = (int)floorf(y1 / (float)B_TILE_H);
        tx0 = tx0 < 0 ? 0 : tx0; ty0 = ty0 < 0 ? 0 : ty0;
        tx1 = tx1 >= tiles_x ? tiles_x - 1 : tx1; ty1 = ty1 >= tiles_y ? tiles_y - 1 : ty1;
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) {
                const int t = ty * tiles_x + tx;
                bitmap[t >> 5] |= 1u << (t & 31);
            }
    };
    auto corners = [&](float left, float top, float w, float h, float c, const DevLayout &L) {
        if (!(c > 0.0f)) return;
        const float bx0 = (float)L.bx0, by0 = (float)L.by0, bx1 = (float)L.bx1 - 1.0f, by1 = (float)L.by1 - 1.0f;  // only where L draws
        const float xs[2][2] = {{left, left + c}, {left + w - c, left + w}}, ys[2][2] = {{top, top + c}, {top + h - c, top + h}};
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                const float x0 = fmaxf(xs[a][0], bx0), x1 = fminf(xs[a][1], bx1), y0 = fmaxf(ys[b][0], by0), y1 = fminf(ys[b][1], by1);
                if (x0 <= x1 && y0 <= y1) mark(x0, y0, x1, y1);
            }
    };
    for (int i = 0; i < p.n; i++) {
        const DevLayout &L = p.host_layouts[i];
        if (L.bx1 <= L.bx0 || L.by1 <= L.by0) continue;
        // (an opaque texture at a fractional position or another scale is sampled, not copied — TC_SAMPLED — but solid all the same)
        const bool solid_capable = (L.flags & DL_UNROTATED) && (L.type == 0 ? L.src_kind == 2 : (L.flags & DL_COLOR_OPAQUE) != 0);
        if (!solid_capable) {
            mark((float)L.bx0, (float)L.by0, (float)L.bx1 - 1.0f, (float)L.by1 - 1.0f);
            continue;
        }
        corners(L.left, L.top, L.width, L.height, L.corner, L);
        for (u32 m = 0; m < L.masks_len; m++) {
            const DevMask &K = p.host_masks[L.masks_off + m];
            corners(K.left, K.top, K.width, K.height, K.corner, L);
            // a mask edge that cuts through the layer leaves a band of partially covered / uncovered pixels
            if (K.left > L.left || K.top > L.top || K.left + K.width < L.left + L.width || K.top + K.height < L.top + L.height)
                mark((float)L.bx0, (float)L.by0, (float)L.bx1 - 1.0f, (float)L.by1 - 1.0f);
        }
        if (L.inset != 0.5f) {  // border, blur or fractional coordinates: bands along the four edges
            const float d = ceilf(L.inset);
            const float x0 = (float)L.bx0, y0 = (float)L.by0, x1 = (float)L.bx1 - 1.0f, y1 = (float)L.by1 - 1.0f;
            mark(x0, y0, x1, fminf(y0 + d, y1)); mark(x0, fmaxf(y1 - d, y0), x1, y1);
            mark(x0, y0, fminf(x0 + d, x1), y1); mark(fmaxf(x1 - d, x0), y0, x1, y1);
        }
    }
    u32 nf = 0;
    for (int w = 0; w < words; w++) nf += (u32)__builtin_popcount(bitmap[w]);
    return nf;
}

__device__ __forceinline__ void classify_layouts(u32 *s_touch, u32 *s_solid, int *s_start, const DevLayout *__restrict__ layouts,
                                                 const DevMask *__restrict__ masks, int n, int x0, int y0, int x1, int y1, int tid,
                                                 int nthreads) {
    if (tid < MAX_LAYOUT_WORDS) { s_touch[tid] = 0; s_solid[tid] = 0; }
    if (tid == 0) *s_start = -1;
    __syncthreads();
    const float cx0 = (float)x0 + 0.5f, cx1 = (float)x1 - 0.5f, cy0 = (float)y0 + 0.5f, cy1 = (float)y1 - 0.5f;
    for (int i = tid; i < n; i += nthreads) {
        const DevLayout &L = layouts[i];
        if (!(L.bx0 < x1 && L.bx1 > x0 && L.by0 < y1 && L.by1 > y0)) continue;
        atomicOr(&s_touch[i >> 5], 1u << (i & 31));
        if (!layout_solid_box(L, masks, cx0, cy0, cx1, cy1)) continue;
        atomicOr(&s_solid[i >> 5], 1u << (i & 31));
        if (layout_base_opaque(L)) atomicMax(s_start, i);
    }
    __syncthreads();
}

// What a tile needs beyond its start layer (after classify_layouts; the caller zeroes *s_general and syncs afterwards):
// bit 0 — some touched layer at or above the start needs blending arithmetic, 2 — the start layer is an opaque texture that is not
// a 1:1 blit.  0 = copy tile.
__device__ __forceinline__ void tile_needs(const u32 *s_touch, int start, int *s_general, const DevLayout *__restrict__ layouts, int n, int tid,
                                           int nthreads) {
    for (int i = tid; i < n; i += nthreads) {
        if (!((s_touch[i >> 5] >> (i & 31)) & 1u) || i < start) continue;
        const DevLayout &L = layouts[i];
        const bool copy_layer = i == start && (L.type != 0 || (L.flags & DL_ALIGNED));
        const bool sample_layer = i == start && L.type == 0 && !(L.flags & DL_ALIGNED);  // opaque texture, solid over the tile, not a 1:1 blit
        if (sample_layer) atomicOr(s_general, 2);
        else if (!copy_layer) atomicOr(s_general, 1);
        // 4: a layer that can blend over an area (translucent, an alpha channel, rotated, a shadow): the tile's list of pixels to composite may
        // be most of the tile.  Without it the layers are opaque and unrotated and only their edges and corners blend.
        if (!(layout_base_opaque(L) && (L.flags & DL_UNROTATED))) atomicOr(s_general, 4);
    }
}

// The class of every 128x16 tile, computed once per layout list (smr_render_layouts keeps it while the list repeats — a scene at
// rest — and recomputes it when the list changes): what k_compose_output needs to know to finish a copy tile without reading the
// layout list at all.  Same classification code as the compositor's own (classify_layouts / tile_needs), one 64-thread workgroup
// per tile.
//   TC_CLEAR    no opaque layer and nothing touching: transparent black
//   TC_COLOUR   an opaque colour, nothing above: px = its encoded bytes
//   TC_TEXTURE  a 1:1 blit of an opaque texture, nothing above: base / pitch = the texel under the tile's top-left pixel
//   TC_FULL     anything else: the compositor classifies and composites the tile itself; px = the tile's index in the list of
//               such tiles (TileList), which the compositor's first workgroups take band by band
//   TC_SKIP     wave A writes the tile's Y'CbCr (direct output)
//   TC_SAMPLED  the whole tile lies in the solid region of one opaque texture layer that is not a 1:1 blit (a video tile at a
//               fractional position or another scale: every tile of a grid in mid-transition), nothing above: px = that layer
//   TC_SELECT   every layer that touches the tile (from the start layer up; at most four) is an opaque 1:1 texture or an opaque colour
//               and SOLID wherever it touches it — the seam between two video tiles of a grid, a tile's edge on the background:
//               a pixel is a copy from the topmost of them whose pixel box holds it (dst * (1 - 1) == 0 wipes what is below, and a
//               layer draws nothing outside its box), no blending arithmetic.  px = the layers, topmost first, a byte each, 0xff = none
enum { TC_CLEAR = 0, TC_COLOUR = 1, TC_TEXTURE = 2, TC_FULL = 3, TC_SKIP = 4, TC_SAMPLED = 5, TC_SELECT = 6 };
struct alignas(16) TileClass {
    const u8 *base;
    u32 pitch_or_px;
    u32 kind;
};
// A tile that needs compositing, with what the classification found out about it (the compositor's workgroups start from this
// record instead of classifying their band again): the touching layers, the last opaque solid layer, tile_needs' bits.
struct TileFull {
    u32 tile;
    int start;
    u32 general;
    u32 band;  // first row of the band within the tile | rows << 8
    u32 touch[MAX_LAYOUT_WORDS];
};
// Bands — list entries, workgroups — of a tile that needs compositing.  A workgroup's time is its round trips to memory (the layout list, the
// start values, a trip per layer of the listed pixels) while its listed pixels fit a pass or two; a tile whose layers are all opaque and unrotated
// blends along their edges and corners only — a few dozen listed pixels — and is two workgroups' (configs[4] in motion: 993 such tiles; root
// compositor 52.5 -> 47.8 us).  A tile under a layer that blends over an area (a translucent label, text, a texture with an alpha channel) lists
// most of its pixels: four shorter bands, four workgroups (configs[2]: alone 17.9 us; 25 us with two bands for every tile).
constexpr int B_AREA_BANDS = 4, B_EDGE_BANDS = B_TILE_H / B_BAND_ROWS;
static_assert(B_TILE_H / B_AREA_BANDS <= B_BAND_ROWS && B_EDGE_BANDS <= B_AREA_BANDS, "a band fits the workgroup's pixel state");
constexpr int B_LIST_COUNTERS = 64;
struct TileList {
    u32 count[B_LIST_COUNTERS];  // the list's length, one counter per classification (a ring: the host zeroes all of them once per
                                 // B_LIST_COUNTERS classifications instead of one before every k_classify_tiles — a fill kernel on the stream
                                 // of every frame of a scene in motion)
    TileFull e[1];               // really one per tile
};
// Direct output: direct[tile] = the start layer when the tile is a copy tile of a texture layer in `direct_layers` (the tiles wave A
// resamples in the same call, at even output positions) — wave A then writes that tile's Y'CbCr itself, the RGBA8 bytes of those
// pixels never exist in memory — else 0xff.
constexpr u32 B_CLASS_NONE = 0xffu;
constexpr int B_CLASSIFY_TILES = 16;  // tiles per workgroup of k_classify_tiles, one wave each: the tiles that need compositing take their
                                      // list slots with ONE atomic per workgroup (a thousand atomics on one counter were most of the
                                      // kernel: 15.3 -> 7.7 us for a 4K output in transition)
__global__ __launch_bounds__(64 * B_CLASSIFY_TILES) void k_classify_tiles(const DevLayout *__restrict__ layouts, const DevMask *__restrict__ masks, int n,
                                                                          int W, int H, int tiles_x, int tiles, unsigned long long direct_layers,
                                                                          TileClass *__restrict__ tc, u8 *__restrict__ direct, TileList *__restrict__ full,
                                                                          int allow_select, int counter) {
    __shared__ u32 s_touch_w[B_CLASSIFY_TILES][MAX_LAYOUT_WORDS], s_solid_w[B_CLASSIFY_TILES][MAX_LAYOUT_WORDS];
    __shared__ int s_start_w[B_CLASSIFY_TILES], s_general_w[B_CLASSIFY_TILES], s_slot_w[B_CLASSIFY_TILES];
    __shared__ TileClass s_class_w[B_CLASSIFY_TILES];
    __shared__ u32 s_direct_w[B_CLASSIFY_TILES];
    // (every wave runs the same barriers: a wave past the last tile classifies the last tile again and writes nothing)
    const int wave = threadIdx.x >> 6, tid = threadIdx.x & 63;
    const bool live = (int)blockIdx.x * B_CLASSIFY_TILES + wave < tiles;
    const int tile = min((int)blockIdx.x * B_CLASSIFY_TILES + wave, tiles - 1);
    u32 *s_touch = s_touch_w[wave], *s_solid = s_solid_w[wave];
    int &s_start = s_start_w[wave], &s_general = s_general_w[wave], &s_slot = s_slot_w[wave];
    const int tile_y = tile / tiles_x;
    const int tx0 = (tile - tile_y * tiles_x) * B_TILE_W, ty0 = tile_y * B_TILE_H;
    if (tid == 0) s_general = 0;
    classify_layouts(s_touch, s_solid, &s_start, layouts, masks, n, tx0, ty0, min(tx0 + B_TILE_W, W), min(ty0 + B_TILE_H, H), tid, 64);
    const int start = s_start;
    tile_needs(s_touch, start, &s_general, layouts, n, tid, 64);
    __syncthreads();
    if (tid == 0) {
        TileClass c;
        c.base = nullptr; c.pitch_or_px = 0u; c.kind = TC_FULL;
        u32 d = B_CLASS_NONE;
        if ((s_general & 3) == 1 && allow_select) {  // some layer above the start: a tile of plain copies all the same?  (serial: a handful of layers)
            const int x1 = min(tx0 + B_TILE_W, W), y1 = min(ty0 + B_TILE_H, H);
            u32 ids = 0xffffffffu;
            int cnt = 0;
            bool ok = true;
            for (int wi = (n + 31) / 32 - 1; ok && wi >= 0; wi--) {
                u32 bits = s_touch[wi];
                while (ok && bits) {
                    const int b = 31 - __clz(bits);
                    bits &= ~(1u << b);
                    const int i = wi * 32 + b;
                    if (i < start) continue;
                    const DevLayout &L = layouts[i];
                    const int ax0 = max(L.bx0, tx0), ay0 = max(L.by0, ty0), ax1 = min(L.bx1, x1), ay1 = min(L.by1, y1);  // (touching: not empty)
                    ok = cnt < 4 && i < 255 && layout_base_opaque(L) && (L.type != 0 || (L.flags & DL_ALIGNED)) &&
                         layout_solid_box(L, masks, (float)ax0 + 0.5f, (float)ay0 + 0.5f, (float)ax1 - 0.5f, (float)ay1 - 0.5f);
                    if (ok) {
                        ids = (ids & ~(0xffu << (8 * cnt))) | ((u32)i << (8 * cnt));
                        cnt++;
                    }
                }
            }
            if (ok && cnt > 0) {
                c.kind = TC_SELECT;
                c.pitch_or_px = ids;
            }
        } else if ((s_general & 3) == 2 && start >= 0) {
            c.kind = TC_SAMPLED;
            c.pitch_or_px = (u32)start;
        } else if ((s_general & 3) == 0) {
            if (start < 0) {
                c.kind = TC_CLEAR;
            } else if (layouts[start].type != 0) {
                c.kind = TC_COLOUR;
                c.pitch_or_px = layouts[start].solid_px;
            } else {
                const DevLayout &L = layouts[start];
                c.kind = TC_TEXTURE;
                c.base = L.src.ptr + (ptrdiff_t)(ty0 - L.iy) * (ptrdiff_t)L.src.pitch + (ptrdiff_t)(tx0 - L.ix) * 4;
                c.pitch_or_px = (u32)L.src.pitch;
                if (start < 64 && ((direct_layers >> start) & 1ull) && tx0 + B_TILE_W <= W && ty0 + B_TILE_H <= H) {
                    c.kind = TC_SKIP;
                    d = (u32)start;
                }
            }
        }
        // (a request; the slots themselves below)  A tile that needs compositing is B_AREA_BANDS bands' work when one of its layers blends over an
        // area, B_EDGE_BANDS longer ones when all of them are opaque and unrotated: one list entry — one workgroup of the compositor — per band
        s_slot = (live && c.kind == TC_FULL) ? ((s_general & 4) ? B_AREA_BANDS : B_EDGE_BANDS) : 0;
        s_class_w[wave] = c;
        s_direct_w[wave] = d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int want = 0;
        for (int w = 0; w < B_CLASSIFY_TILES; w++) want += s_slot_w[w];
        const u32 base = want ? atomicAdd(&full->count[counter], (u32)want) : 0u;
        int at = 0;
        for (int w = 0; w < B_CLASSIFY_TILES; w++) {
            const int mine = s_slot_w[w] ? (int)base + at : -1;
            at += s_slot_w[w];
            s_slot_w[w] = mine;
        }
    }
    __syncthreads();
    const int bands = (s_general & 4) ? B_AREA_BANDS : B_EDGE_BANDS;
    if (tid == 0 && live) {
        TileClass c = s_class_w[wave];
        if (c.kind == TC_FULL) { c.pitch_or_px = (u32)s_slot; c.base = (const u8 *)(uintptr_t)bands; }  // (first entry, number of entries)
        tc[tile] = c;
        direct[tile] = (u8)s_direct_w[wave];
    }
    if (s_slot >= 0) {  // (uniform per wave) the records the compositor's bands of this tile start from
        const int rows = B_TILE_H / bands;
        for (int j = 0; j < bands; j++) {
            TileFull &E = full->e[s_slot + j];
            if (tid < MAX_LAYOUT_WORDS) E.touch[tid] = tid < ((n + 31) >> 5) ? s_touch[tid] : 0u;
            if (tid == 0) { E.tile = (u32)tile; E.start = start; E.general = (u32)s_general; E.band = (u32)(j * rows) | ((u32)rows << 8); }
        }
    }
}

// Wave-uniform copy of a record (LDS or global) into scalar registers: the dword loads are issued back to back
// (one wait), v_readfirstlane moves them to SGPRs, and every later use is a scalar operand / a uniform branch.
template <typename T>
__device__ __forceinline__ T load_uniform(const T *p) {
    static_assert(sizeof(T) % 4 == 0, "dword records only");
    T out;  // (through memcpy both ways: the record's fields are floats and ints, not u32 objects)
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); i++) {
        u32 w;
        __builtin_memcpy(&w, (const char *)p + 4 * i, 4);
        w = __builtin_amdgcn_readfirstlane(w);
        __builtin_memcpy((char *)&out + 4 * i, &w, 4);
    }
    return out;
}

__device__ __forceinline__ float4 decode_texel(u32 raw, int srgb, const float *__restrict__ dec) {
    const u32 r = raw & 0xff, g = (raw >> 8) & 0xff, b = (raw >> 16) & 0xff, a = raw >> 24;
    float4 o;
    if (srgb) { o.x = dec[r]; o.y = dec[g]; o.z = dec[b]; }
    else { o.x = (float)r / 255.0f; o.y = (float)g / 255.0f; o.z = (float)b / 255.0f; }
    o.w = div_cr((float)a, 255.0f, 1.0f / 255.0f);
    return o;
}

// composite_layout() for one pixel with the exact shortcuts the fused kernel can take:
//   * an opaque colour in its solid region stores its pre-encoded bytes (dst * 0 vanishes);
//   * an aligned opaque texel whose fragment came out equal to the sample (coverage, border and every mask evaluated
//     to exactly 1) stores its own bytes: encode(decode(b)) == b;
//   * `is_base`: the caller established that this layer is opaque and solid at the pixel.
__device__ __forceinline__ u32 aligned_texel(const DevLayout &L, int px, int py) {
    return g_ld_u32(L.src.ptr + b_off(clampi(py - L.iy, 0, L.tex_h - 1), L.src.pitch, clampi(px - L.ix, 0, L.tex_w - 1) * 4));  // (a global load: the layer record came through LDS / scalar registers)
}

// `raw_pre`: the texel of an aligned texture layer, fetched by the caller ahead of the arithmetic (aligned_texel)
__device__ __forceinline__ u32 compose_px(u32 a, const DevLayout &L, const DevMask *__restrict__ masks, int px, int py, bool is_base,
                                          u32 raw_pre, int srgb, const float *__restrict__ dec, const float *__restrict__ thr) {
    float fx, fy, lx, ly;
    if (!layout_covers(L, px, py, fx, fy, lx, ly)) return a;
    const bool solid = is_base || layout_solid_box(L, masks, fx, fy, fx, fy);
    if (L.type != 0) {
        if (solid && (L.flags & DL_COLOR_OPAQUE)) return L.solid_px;
        const float4 frag = solid ? make_float4(L.color[0], L.color[1], L.color[2], L.color[3])
                                  : layout_fragment(L, masks, fx, fy, lx, ly, make_float4(0.f, 0.f, 0.f, 0.f));
        return blend_store(a, frag, srgb, dec, thr);
    }
    const bool aligned = (L.flags & DL_ALIGNED) && L.src_kind != 0;
    u32 raw = 0u;
    float4 sample;
    if (aligned) {
        raw = raw_pre;
        sample = decode_texel(raw, srgb, dec);
    } else {
        sample = layout_texture_sample(L, px, py, lx, ly, srgb, dec);
    }
    const float4 frag = solid ? sample : layout_fragment(L, masks, fx, fy, lx, ly, sample);
    if (aligned && (raw >> 24) == 255u && frag.x == sample.x && frag.y == sample.y && frag.z == sample.z && frag.w == sample.w) return raw;
    return blend_store(a, frag, srgb, dec, thr);
}

template <int NV>
__device__ __forceinline__ void store_yuv_block(const u32 (&acc)[8], int px0, int py0, int W, const SurfView &yp, const SurfView &up, const SurfView &vp) {
    // (W is even: the last block of a row holds four or two pixels)
    const bool half = px0 + 3 >= W;
    if (NV == 2) {  // an RGBA8 node texture (LayoutNode::render into a NodeTexture): the composited bytes as they are
        u8 *o = yp.ptr + b_off(py0, yp.pitch, px0 * 4);
        if (half) {
            *(uint2 *)o = make_uint2(acc[0], acc[1]);
            *(uint2 *)(o + yp.pitch) = make_uint2(acc[4], acc[5]);
        } else {
            *(uint4 *)o = make_uint4(acc[0], acc[1], acc[2], acc[3]);
            *(uint4 *)(o + yp.pitch) = make_uint4(acc[4], acc[5], acc[6], acc[7]);
        }
        return;
    }
    // RGBA -> Y'CbCr on the raw (gamma-encoded) bytes: rgba_to_yuv.wgsl:26-54 / rgba_to_nv12.wgsl:24-52 — the bytes yuv_component() /
    // unorm8() compute, through the fast path of smr_yuv_fast.h: three FMAs per value on the bytes as floats, the reference sequence itself
    // (unorm_of_byte, yuv_byte) only where the guard flag asks for it (smr_convert_dev.h: yuv_luma_byte, yuv_chroma_bytes)
    float cr[8], cg[8], cb[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        cr[k] = (float)(acc[k] & 0xffu);
        cg[k] = (float)((acc[k] >> 8) & 0xffu);
        cb[k] = (float)((acc[k] >> 16) & 0xffu);
    }
    u32 yrow0 = 0, yrow1 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        yrow0 |= yuv_luma_byte(acc[k], cr[k], cg[k], cb[k]) << (8 * k);
        yrow1 |= yuv_luma_byte(acc[4 + k], cr[4 + k], cg[4 + k], cb[4 + k]) << (8 * k);
    }
    if (half) {
        *(u16 *)(yp.ptr + b_off(py0, yp.pitch, px0)) = (u16)yrow0;
        *(u16 *)(yp.ptr + b_off(py0 + 1, yp.pitch, px0)) = (u16)yrow1;
    } else {
        *(u32 *)(yp.ptr + b_off(py0, yp.pitch, px0)) = yrow0;
        *(u32 *)(yp.ptr + b_off(py0 + 1, yp.pitch, px0)) = yrow1;
    }
    // chroma: the bilinear tap at the chroma texel centre = weights (1/2, 1/2) x (1/2, 1/2):
    //     (a * .5 + b * .5) * .5 + (c * .5 + d * .5) * .5  ==  ((a + b) + (c + d)) * .25   bit for bit
    // (a power of two scales exactly and commutes with rounding — the operands are 0 or >= 1/255, nowhere near the subnormals);
    // the fast path takes the block's byte sums (exact in f32)
    u32 uv[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int a = 2 * j, b = 2 * j + 1, c = 4 + 2 * j, d = 4 + 2 * j + 1;
        const u32 q = yuv_chroma_bytes(acc[a], acc[b], acc[c], acc[d], (cr[a] + cr[b]) + (cr[c] + cr[d]), (cg[a] + cg[b]) + (cg[c] + cg[d]),
                                       (cb[a] + cb[b]) + (cb[c] + cb[d]));
        uv[j][0] = q & 0xffu;
        uv[j][1] = q >> 8;
    }
    const int cx = px0 >> 1, cy = py0 >> 1;
    if (NV == 0) {
        if (half) {
            up.ptr[b_off(cy, up.pitch, cx)] = (u8)uv[0][0];
            vp.ptr[b_off(cy, vp.pitch, cx)] = (u8)uv[0][1];
        } else {
            *(u16 *)(up.ptr + b_off(cy, up.pitch, cx)) = (u16)(uv[0][0] | (uv[1][0] << 8));
            *(u16 *)(vp.ptr + b_off(cy, vp.pitch, cx)) = (u16)(uv[0][1] | (uv[1][1] << 8));
        }
    } else if (half) {
        *(u16 *)(up.ptr + b_off(cy, up.pitch, cx * 2)) = (u16)(uv[0][0] | (uv[0][1] << 8));
    } else {
        *(u32 *)(up.ptr + b_off(cy, up.pitch, cx * 2)) = uv[0][0] | (uv[0][1] << 8) | (uv[1][0] << 16) | (uv[1][1] << 24);
    }
}

// One LDS atomic per wave hands the wave's flagged pixels their slots on the workgroup's list (uniform call).
__device__ __forceinline__ u32 list_slot(u32 *count, bool want) {
#ifdef SMR_EMU
    return want ? atomicAdd(count, 1u) : 0u;  // (the lane emulator has no wave votes)
#else
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return 0u;
    const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));  // flagged lanes below this one
    u32 base = 0u;
    if (want && rank == 0u) base = atomicAdd(count, (u32)__popcll(m));
    return (u32)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m)) + rank;
#endif
}
__device__ __forceinline__ bool wave_any(bool v) {
#ifdef SMR_EMU
    return true;
#else
    return __ballot(v) != 0ull;
#endif
}

// Rows [band, band + rh) of tile `tile`, composited by the whole workgroup (uniform call; may be called again).
// NV = 0: planar Y,U,V (4:2:0); NV = 1: NV12 (Y + interleaved UV); NV = 2: RGBA8 surface (in `yp`)
// BIG: the layout list is longer than the LDS copy (B_MAX_LAYOUTS / B_MAX_MASKS) and is read where it lies in memory
// `pre`: the tile's record from k_classify_tiles — touching layers, start layer and tile_needs' bits of the WHOLE tile, valid for every
// band of it (a layer solid over the tile is solid over the band; a superset of the band's touching layers composites to the same
// pixels) — so a band's workgroup does not classify again.
//
// Three steps, one pixel per thread and sweep (a wave covers 64 consecutive pixels of a row), layer records through scalar registers:
//   A  every pixel's START: the topmost touched layer that is opaque, unrotated and solid at the pixel (else the tile's own start layer),
//      and that layer's value there — an opaque colour's bytes, a 1:1 texel, a filtered sample of a texture at a fractional position:
//      what compositing from the start layer upwards begins with (dst * (1 - 1) wipes what is below).  The layers are walked first and
//      each pixel keeps where its value lies (address, weights); then ALL fetches go out together: one round trip whatever the number
//      of layers (the earlier kernel composited layer after layer, sweep after sweep: a round trip for each).
//   B  the pixels that some layer above their start touches (the layers' pixel boxes) go on a list in LDS — the edge of a video tile,
//      a rounded corner, a label: a few dozen of a band's 512 pixels in a grid in motion.
//   C  the listed pixels are composited upwards from their start value, a thread per listed pixel, by compose_px — the one copy of the
//      general per-pixel code — and written back; the band is then converted and stored as 4 x 2 blocks.
template <int NV, bool BIG>
__device__ __forceinline__ void compose_full(const TileFull *__restrict__ pre, int band, int rh, const SurfView &yp, const SurfView &up, const SurfView &vp, int W, int H,
                                             const DevLayout *__restrict__ layouts_g, const DevMask *__restrict__ masks_g, int n, int n_masks,
                                             int srgb_and_ablate, const float *__restrict__ tables, int tiles_x, float *s_tab) {
    const int tile = (int)pre->tile;
    constexpr int BAND_PX = B_TILE_W * B_BAND_ROWS;
    constexpr int SWEEPS = 2;  // pixels per thread of steps A and B at a time (rows y and y + 2 of a four-row slab): the band is walked slab by slab
    static_assert(BAND_PX % 512 == 0 && BAND_PX <= 2048, "whole slabs; a list entry keeps the pixel's index in IDX_BITS bits");
    // a list entry: pixel index | (start layer + 1) << IDX_BITS — 16 bits while the list's layers fit the LDS copy, 32 for the build that reads the list in place
    constexpr int IDX_BITS = BIG ? 11 : (BAND_PX <= 1024 ? 10 : 11);
    constexpr bool WIDE = BIG || ((B_MAX_LAYOUTS + 1) > (1 << (16 - IDX_BITS)));
    typedef typename std::conditional<WIDE, u32, u16>::type entry_t;
    __shared__ u32 s_touch[MAX_LAYOUT_WORDS];
    __shared__ u32 s_px[BAND_PX];         // the band's RGBA8: every pixel's start value, then the listed pixels composited
    __shared__ entry_t s_list[BAND_PX];   // pixels some layer above their start touches
    __shared__ u32 s_count;
    // the whole layout list lives in LDS for the lifetime of the workgroup: one coalesced copy instead of a
    // dependent scalar-memory round trip per field per layer per wave
    __shared__ __attribute__((aligned(16))) DevLayout s_lay[B_MAX_LAYOUTS];
    __shared__ __attribute__((aligned(16))) DevMask s_mask[B_MAX_MASKS];
    const int tid = threadIdx.x;
    u32 acc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // this thread's 4x2 output block, [row][col]: acc[r * 4 + c]
    const int tile_y = tile / tiles_x;
    const int tx0 = (tile - tile_y * tiles_x) * B_TILE_W, ty0 = tile_y * B_TILE_H + band;
    const int px0 = tx0 + 4 * (tid & 31), py0 = ty0 + 2 * (tid >> 5);  // this thread's 4x2 output block
    __syncthreads();  // (a previous call's readers of the shared tile are done)
    // (one round trip: every load of the list, the masks and the tables is issued before the first store — as loops with a load and a
    //  store per trip they were eight dependent trips to the L2)
    {
        static_assert(SMR_TABLE_FLOATS % 4 == 0 && SMR_TABLE_FLOATS / 4 <= 256, "one 16 B word of the tables per thread");
        constexpr int LW = (B_MAX_LAYOUTS * (int)(sizeof(DevLayout) / 16) + 255) / 256, MW = (B_MAX_MASKS * (int)(sizeof(DevMask) / 16) + 255) / 256;
        const uint4 *gl = (const uint4 *)layouts_g, *gm = (const uint4 *)masks_g, *gt = (const uint4 *)tables;
        const int nl = BIG ? 0 : n * (int)(sizeof(DevLayout) / 16), nm = BIG ? 0 : n_masks * (int)(sizeof(DevMask) / 16);
        uint4 wl[LW], wm[MW], wt = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int k = 0; k < LW; k++) {
            wl[k] = wt;
            if (tid + 256 * k < nl) wl[k] = gl[tid + 256 * k];
        }
#pragma unroll
        for (int k = 0; k < MW; k++) {
            wm[k] = wt;
            if (tid + 256 * k < nm) wm[k] = gm[tid + 256 * k];
        }
        if (tid < SMR_TABLE_FLOATS / 4) wt = gt[tid];
        u32 touch_w = 0u;
        if (tid < MAX_LAYOUT_WORDS) touch_w = pre->touch[tid];
#pragma unroll
        for (int k = 0; k < LW; k++)
            if (tid + 256 * k < nl) ((uint4 *)s_lay)[tid + 256 * k] = wl[k];
#pragma unroll
        for (int k = 0; k < MW; k++)
            if (tid + 256 * k < nm) ((uint4 *)s_mask)[tid + 256 * k] = wm[k];
        if (tid < SMR_TABLE_FLOATS / 4) ((uint4 *)s_tab)[tid] = wt;
        if (tid < MAX_LAYOUT_WORDS) s_touch[tid] = touch_w;
        if (tid == 0) s_count = 0u;
    }
    const DevLayout *layouts = BIG ? layouts_g : s_lay;
    const DevMask *masks = BIG ? masks_g : s_mask;
    const int srgb = srgb_and_ablate & 1;
    const int ablate = srgb_and_ablate >> 8;  // profiling only (SMR_ABLATE bits 8..): 1 dispatch only, 2 list only, 4 no per-pixel start, 8 start values only, 16 no compositing
    const int start = pre->start;
    __syncthreads();
    if (ablate & (1 | 2 | 16)) return;
    if (ty0 >= H) return;
    const int words = (n + 31) >> 5;
    const bool no_block = px0 >= W || py0 >= H || 2 * (tid >> 5) >= rh;      // (W, H even: a block has four or two columns, always two rows)
    const int nsweeps = (B_TILE_W * rh) / 256;
    const float *dec = s_tab, *thr = s_tab + 256;
    const int sx = tx0 + (tid & (B_TILE_W - 1));  // this thread's pixels: column sx, rows ty0 + (tid >> 7) + 2 * sweep

    // ---- A. start layer and start value of every pixel, B. the list — a slab of four rows at a time
#pragma unroll 1
    for (int slab = 0; slab < nsweeps; slab += SWEEPS) {
    enum { K_NONE = 0, K_COLOUR = 1, K_TEXEL = 2, K_QUAD = 3 };
    int sp[SWEEPS];          // the layer the pixel starts from (-1: none, a cleared pixel)
    u32 kind[SWEEPS];        // where its start value lies
    const u8 *sbase[SWEEPS];
    u32 so[SWEEPS][4];       // K_COLOUR: [0] = the encoded bytes; K_TEXEL: [0] = byte offset; K_QUAD: the footprint's four byte offsets
    float swx[SWEEPS], swy[SWEEPS];
#pragma unroll
    for (int s = 0; s < SWEEPS; s++) { sp[s] = -1; kind[s] = K_NONE; sbase[s] = nullptr; so[s][0] = so[s][1] = so[s][2] = so[s][3] = 0u; swx[s] = swy[s] = 0.0f; }
    const auto claim = [&](const DevLayout &L, int li, int s, int py) {
        sp[s] = li;
        if (L.type != 0) {
            kind[s] = K_COLOUR;
            so[s][0] = L.solid_px;
        } else if (L.flags & DL_ALIGNED) {
            kind[s] = K_TEXEL;
            sbase[s] = L.src.ptr;
            so[s][0] = (u32)b_off(clampi(py - L.iy, 0, L.tex_h - 1), L.src.pitch, clampi(sx - L.ix, 0, L.tex_w - 1) * 4);
        } else {
            kind[s] = K_QUAD;
            sbase[s] = L.src.ptr;
            const SampledAxis X = sampled_axis_x(L, sx), Y = sampled_axis_y(L, py);
            so[s][0] = Y.o0 + X.o0; so[s][1] = Y.o0 + X.o1; so[s][2] = Y.o1 + X.o0; so[s][3] = Y.o1 + X.o1;
            swx[s] = X.f; swy[s] = Y.f;
        }
    };
    // (topmost first: a pixel is claimed once)
    for (int wi = (ablate & 4) ? -1 : words - 1; wi >= (start < 0 ? 0 : (start >> 5)); wi--) {  // (4: profiling, the tile's start layer only)
        u32 bits = __builtin_amdgcn_readfirstlane(s_touch[wi]);
        if (start >= 0 && wi == (start >> 5)) bits &= ~((2u << (start & 31)) - 1u);  // strictly above start
        while (bits) {
            const int b = 31 - __builtin_clz(bits);
            bits &= ~(1u << b);
            const int li = (wi << 5) + b;
            const DevLayout L = load_uniform(&layouts[li]);
            if (!layout_base_opaque(L) || !(L.flags & DL_UNROTATED)) continue;
#pragma unroll
            for (int s = 0; s < SWEEPS; s++) {
                const int py = ty0 + (tid >> 7) + 2 * (slab + s);
                const float fx = (float)sx + 0.5f, fy = (float)py + 0.5f;
                if (slab + s < nsweeps && kind[s] == K_NONE && layout_solid_box(L, masks, fx, fy, fx, fy)) claim(L, li, s, py);
            }
        }
    }
    if (start >= 0) {  // (opaque, unrotated and solid over the whole tile: classify_layouts)
        const DevLayout L = load_uniform(&layouts[start]);
#pragma unroll
        for (int s = 0; s < SWEEPS; s++)
            if (slab + s < nsweeps && kind[s] == K_NONE) claim(L, start, s, ty0 + (tid >> 7) + 2 * (slab + s));
    }
    u32 st[SWEEPS][4];
#pragma unroll
    for (int s = 0; s < SWEEPS; s++) {  // (every fetch of the band before the first use)
        st[s][0] = st[s][1] = st[s][2] = st[s][3] = 0u;
        if (kind[s] >= K_TEXEL) st[s][0] = g_ld_u32(sbase[s] + so[s][0]);
        if (kind[s] == K_QUAD) { st[s][1] = g_ld_u32(sbase[s] + so[s][1]); st[s][2] = g_ld_u32(sbase[s] + so[s][2]); st[s][3] = g_ld_u32(sbase[s] + so[s][3]); }
    }
    u32 val[SWEEPS];
#pragma unroll
    for (int s = 0; s < SWEEPS; s++) {
        // (compose_px's results for a layer that is opaque and solid at the pixel: an opaque colour stores its bytes; an opaque 1:1 texel
        //  stores its own bytes, encode(decode(b)) == b; an opaque sample is filtered and encoded: composite_sampled_opaque)
        val[s] = kind[s] == K_COLOUR ? so[s][0] : kind[s] == K_TEXEL ? st[s][0] : 0u;
        if (kind[s] == K_QUAD) val[s] = filter_opaque_quad(st[s][0], st[s][1], st[s][2], st[s][3], swx[s], swy[s], srgb, dec, thr);
    }

    // ---- B. the pixels a layer above their start touches
    bool listed[SWEEPS];
#pragma unroll
    for (int s = 0; s < SWEEPS; s++) listed[s] = false;
    if (!(ablate & 8))
        for (int wi = start < 0 ? 0 : (start >> 5); wi < words; wi++) {
            u32 bits = __builtin_amdgcn_readfirstlane(s_touch[wi]);
            if (start >= 0 && wi == (start >> 5)) bits &= ~((2u << (start & 31)) - 1u);
            while (bits) {
                const int li = (wi << 5) + __builtin_ctz(bits);
                bits &= bits - 1;
                const DevLayout *Lp = &layouts[li];
                const int bx0 = __builtin_amdgcn_readfirstlane(Lp->bx0), bx1 = __builtin_amdgcn_readfirstlane(Lp->bx1);
                const int by0 = __builtin_amdgcn_readfirstlane(Lp->by0), by1 = __builtin_amdgcn_readfirstlane(Lp->by1);
#pragma unroll
                for (int s = 0; s < SWEEPS; s++) {
                    const int py = ty0 + (tid >> 7) + 2 * (slab + s);
                    if (li > sp[s] && sx >= bx0 && sx < bx1 && py >= by0 && py < by1) listed[s] = true;
                }
            }
        }
#pragma unroll
    for (int s = 0; s < SWEEPS; s++) {
        if (slab + s >= nsweeps) break;  // (uniform)
        const int idx = (slab + s) * 256 + tid, py = ty0 + (tid >> 7) + 2 * (slab + s);
        s_px[idx] = val[s];
        const bool want = listed[s] && sx < W && py < H;
        const u32 slot = list_slot(&s_count, want);
        if (want) s_list[slot] = (entry_t)((u32)idx | ((u32)(sp[s] + 1) << IDX_BITS));
    }
    }
    __syncthreads();

    // ---- C. the listed pixels, composited upwards from their start value
    const u32 total = s_count;
#pragma unroll 1
    for (u32 k0 = 0; k0 < total; k0 += 256u) {
        const bool act = k0 + (u32)tid < total;
        const u32 e = act ? (u32)s_list[k0 + tid] : 0u;
        const int idx = (int)(e & ((1u << IDX_BITS) - 1u)), from = (int)(e >> IDX_BITS) - 1;
        const int px = tx0 + (idx & (B_TILE_W - 1)), py = ty0 + (idx >> 7);
        u32 a = s_px[idx];
        for (int wi = start < 0 ? 0 : (start >> 5); wi < words; wi++) {
            u32 bits = __builtin_amdgcn_readfirstlane(s_touch[wi]);
            if (start >= 0 && wi == (start >> 5)) bits &= ~((2u << (start & 31)) - 1u);
            while (bits) {
                const int li = (wi << 5) + __builtin_ctz(bits);
                bits &= bits - 1;
                const DevLayout *Lp = &layouts[li];
                const int bx0 = __builtin_amdgcn_readfirstlane(Lp->bx0), bx1 = __builtin_amdgcn_readfirstlane(Lp->bx1);
                const int by0 = __builtin_amdgcn_readfirstlane(Lp->by0), by1 = __builtin_amdgcn_readfirstlane(Lp->by1);
                const bool need = act && li > from && px >= bx0 && px < bx1 && py >= by0 && py < by1;
                if (!wave_any(need)) continue;  // (none of this wave's pixels: the layer's record stays where it is)
                const DevLayout L = load_uniform(Lp);
                u32 raw = 0u;
                if (need && L.type == 0 && (L.flags & DL_ALIGNED) && L.src_kind != 0) raw = aligned_texel(L, px, py);
                if (need) a = compose_px(a, L, masks, px, py, false, raw, srgb, dec, thr);
            }
        }
        if (act) s_px[idx] = a;
    }
    __syncthreads();
    if (!no_block) {
        const int lx0 = px0 - tx0, ly0 = py0 - ty0;
        const uint4 t0 = *(const uint4 *)&s_px[ly0 * B_TILE_W + lx0], t1 = *(const uint4 *)&s_px[(ly0 + 1) * B_TILE_W + lx0];
        acc[0] = t0.x; acc[1] = t0.y; acc[2] = t0.z; acc[3] = t0.w;
        acc[4] = t1.x; acc[5] = t1.y; acc[6] = t1.z; acc[7] = t1.w;
        store_yuv_block<NV>(acc, px0, py0, W, yp, up, vp);
    }
}

// Copy tiles per workgroup of the kernel's second part (their records, then all their texels, are fetched back to back).
// Measured on configs[2]: 1 -> 29 us, 4 -> 37 us for the whole kernel — the conversion arithmetic of four tiles in one workgroup
// outweighs the launches saved.  Round 4, same kernel three rounds later (tools/r04_run16.sh, -DSMR_COPY_TILES=2): 22.5 -> 29.9 us on configs[2],
// 45.3 -> 53.1 us on configs[4]: still one.
#ifndef SMR_COPY_TILES
#define SMR_COPY_TILES 1
#endif
constexpr int B_COPY_TILES = SMR_COPY_TILES;
static_assert(B_COPY_TILES == 1 || B_COPY_TILES == 2, "one or two tiles per workgroup");

template <int NV, bool BIG>
__global__ __launch_bounds__(256, B_MIN_WAVES) void k_compose_output(SurfView yp, SurfView up, SurfView vp, int W, int H,
                                                        const DevLayout *__restrict__ layouts_g, const DevMask *__restrict__ masks_g,
                                                        int n, int n_masks, int srgb_and_ablate, const float *__restrict__ tables,
                                                        int tiles_x, int tiles, const TileClass *__restrict__ tc, const TileList *__restrict__ full,
                                                        int n_banded, int counter) {
    const int tid = threadIdx.x;
#if defined(SMR_PRIO_COMPOSE) && !defined(SMR_EMU)
    __builtin_amdgcn_s_setprio(SMR_PRIO_COMPOSE);
#endif
    __shared__ __attribute__((aligned(16))) float s_tab[SMR_TABLE_FLOATS];  // decode / encode tables (whoever needs them loads them)
    // The first `n_banded` workgroups take the bands of the tiles that need compositing, an entry of the TileList each (the host sized the
    // grid from the list's length when it knows it, from its own prediction otherwise) — they are latency-bound and would be the tail of
    // the kernel.  Every other workgroup takes B_COPY_TILES consecutive tiles of the row-major order.
    // (the compositing path is most of the kernel's code: it is instantiated once, at the end, and both ways into it — a band of a
    //  listed tile; a tile the list had no room for — only choose its arguments)
    const int rest = (int)blockIdx.x - n_banded;
    const TileFull *full_entry = nullptr, *full_entry2 = nullptr;
    int full_entries = 1, full_entries2 = 1;
    if (rest < 0) {
        if (blockIdx.x >= full->count[counter]) return;
        full_entry = &full->e[blockIdx.x];
    } else {
    // ---- copy tiles, straight from their class records (k_classify_tiles): no layout list, no classification, no barrier
    const int t0 = rest * B_COPY_TILES;
    const int bx = 4 * (tid & 31), by = 2 * (tid >> 5);
    TileClass c[B_COPY_TILES];
#pragma unroll
    for (int k = 0; k < B_COPY_TILES; k++) {
        c[k] = tc[min(t0 + k, tiles - 1)];
        if (t0 + k >= tiles) c[k].kind = TC_SKIP;
    }
    u32 acc[B_COPY_TILES][8];
    bool on[B_COPY_TILES];
    bool aligned = true;  // (uniform) every texture tile of the group can be read 16 bytes at a time
#pragma unroll
    for (int k = 0; k < B_COPY_TILES; k++) {
        const int tile = t0 + k, ty = tile / tiles_x;
        on[k] = c[k].kind <= TC_TEXTURE && !((srgb_and_ablate >> 8) & 64) && (tile - ty * tiles_x) * B_TILE_W + bx < W && ty * B_TILE_H + by < H;
        if (c[k].kind == TC_TEXTURE && ((((uintptr_t)c[k].base) & 15) != 0 || (c[k].pitch_or_px & 15) != 0)) aligned = false;
        if ((W & 3) && (tile - ty * tiles_x) * B_TILE_W + B_TILE_W > W) aligned = false;  // the row's last block is two pixels wide
    }
    if (aligned) {
        // straight-line: the texel loads of all the group's tiles are in flight together (a lane with nothing to read reads the
        // tables instead — no branch between the loads); 1:1 blit of an opaque texture: decode -> encode is the identity
        uint4 ra[B_COPY_TILES], rb[B_COPY_TILES];
#pragma unroll
        for (int k = 0; k < B_COPY_TILES; k++) {
            const bool tex = on[k] && c[k].kind == TC_TEXTURE;
            const u8 *r0 = tex ? c[k].base + b_off(by, c[k].pitch_or_px, bx * 4) : (const u8 *)tables;
            const u8 *r1 = tex ? r0 + c[k].pitch_or_px : (const u8 *)tables;
            ra[k] = g_ld_u32x4(r0);  // (global loads: a base from a record in memory is a generic pointer to the compiler — smr_internal.h)
            rb[k] = g_ld_u32x4(r1);
        }
#pragma unroll
        for (int k = 0; k < B_COPY_TILES; k++) {
            const bool tex = c[k].kind == TC_TEXTURE;
            const u32 fill = c[k].kind == TC_COLOUR ? c[k].pitch_or_px : 0u;
            acc[k][0] = tex ? ra[k].x : fill; acc[k][1] = tex ? ra[k].y : fill; acc[k][2] = tex ? ra[k].z : fill; acc[k][3] = tex ? ra[k].w : fill;
            acc[k][4] = tex ? rb[k].x : fill; acc[k][5] = tex ? rb[k].y : fill; acc[k][6] = tex ? rb[k].z : fill; acc[k][7] = tex ? rb[k].w : fill;
        }
    } else {
#pragma unroll
        for (int k = 0; k < B_COPY_TILES; k++) {
            const u32 fill = c[k].kind == TC_COLOUR ? c[k].pitch_or_px : 0u;
#pragma unroll
            for (int q = 0; q < 8; q++) acc[k][q] = fill;
            if (on[k] && c[k].kind == TC_TEXTURE) {
                const u8 *r0 = c[k].base + b_off(by, c[k].pitch_or_px, bx * 4), *r1 = r0 + c[k].pitch_or_px;
                const int tile = t0 + k, cols = W - ((tile - (tile / tiles_x) * tiles_x) * B_TILE_W + bx);  // >= 2
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (q < cols) { acc[k][q] = g_ld_u32(r0 + 4 * q); acc[k][4 + q] = g_ld_u32(r1 + 4 * q); }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < B_COPY_TILES; k++) {
        const int tile = t0 + k, ty = tile / tiles_x;
        if (on[k]) store_yuv_block<NV>(acc[k], (tile - ty * tiles_x) * B_TILE_W + bx, ty * B_TILE_H + by, W, yp, up, vp);
    }
    // sampled tiles: one layer's record straight from the list in memory, eight independent pixels per thread
#pragma unroll 1
    for (int k = 0; k < B_COPY_TILES; k++) {
        if (c[k].kind != TC_SAMPLED || ((srgb_and_ablate >> 8) & 32)) continue;  // (32: profiling, no sampled tiles)
        __syncthreads();
        for (int i = tid; i < SMR_TABLE_FLOATS; i += 256) s_tab[i] = tables[i];
        __syncthreads();
        const DevLayout L = load_uniform(&layouts_g[c[k].pitch_or_px]);
        const int tile = t0 + k, ty = tile / tiles_x;
        const int px0 = (tile - ty * tiles_x) * B_TILE_W + bx, py0 = ty * B_TILE_H + by;
        if (px0 < W && py0 < H) {
            u32 a[8];
            composite_sampled_opaque_block(L, px0, py0, srgb_and_ablate & 1, s_tab, s_tab + 256, a);  // (column / row halves of the sample positions once each)
            store_yuv_block<NV>(a, px0, py0, W, yp, up, vp);
        }
    }
    // select tiles: every pixel a copy from the topmost listed layer whose pixel box holds it (records through scalar loads: the layer
    // numbers are the workgroup's)
#pragma unroll 1
    for (int k = 0; k < B_COPY_TILES; k++) {
        if (c[k].kind != TC_SELECT) continue;
        const int tile = t0 + k, ty = tile / tiles_x;
        const int px0 = (tile - ty * tiles_x) * B_TILE_W + bx, py0 = ty * B_TILE_H + by;
        u32 a[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        u32 open_px = 0xffu;  // pixels no layer has claimed yet
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
            const u32 li = (c[k].pitch_or_px >> (8 * q)) & 0xffu;
            if (li == 0xffu) break;  // (uniform)
            const DevLayout &L = layouts_g[li];
            const int lx0 = L.bx0, ly0 = L.by0, lx1 = L.bx1, ly1 = L.by1;
            const bool colour = L.type != 0;
            const u32 solid = L.solid_px;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int px = px0 + (i & 3), py = py0 + (i >> 2);
                if (((open_px >> i) & 1u) && px >= lx0 && px < lx1 && py >= ly0 && py < ly1) {
                    a[i] = colour ? solid : aligned_texel(L, px, py);
                    open_px &= ~(1u << i);
                }
            }
        }
        if (px0 < W && py0 < H) store_yuv_block<NV>(a, px0, py0, W, yp, up, vp);
    }
    // a tile that needs compositing and whose bands found no room (or not all of them) among the first workgroups (the host's bound was
    // short): here, band after band
    if (c[0].kind == TC_FULL && (int)c[0].pitch_or_px + (int)(uintptr_t)c[0].base > n_banded) {
        full_entry = &full->e[c[0].pitch_or_px];
        full_entries = (int)(uintptr_t)c[0].base;
    }
    if (B_COPY_TILES > 1 && c[B_COPY_TILES - 1].kind == TC_FULL && (int)c[B_COPY_TILES - 1].pitch_or_px + (int)(uintptr_t)c[B_COPY_TILES - 1].base > n_banded) {
        const TileFull *e = &full->e[c[B_COPY_TILES - 1].pitch_or_px];
        if (full_entry) { full_entry2 = e; full_entries2 = (int)(uintptr_t)c[B_COPY_TILES - 1].base; }
        else { full_entry = e; full_entries = (int)(uintptr_t)c[B_COPY_TILES - 1].base; }
    }
    }
    // (uniform; one band of a listed tile, or the bands of a tile — of each tile of the group — that the list's bound left out)
#ifdef SMR_COMPOSE_NO_FULL  // profiling builds only: the compositing path compiled out (what the copy paths alone need in registers and time)
    if (full_entry || full_entry2) return;
#endif
#pragma unroll 1
    for (int q = 0; q < 2; q++) {
        const TileFull *e = q == 0 ? full_entry : full_entry2;
        if (!e) break;
        const int entries = q == 0 ? full_entries : full_entries2;
#pragma unroll 1
        for (int j = 0; j < entries; j++) {
            if (rest >= 0 && (int)(e - full->e) + j < n_banded) continue;  // (that band has a workgroup of its own)
            const u32 bw = __builtin_amdgcn_readfirstlane(e[j].band);
            compose_full<NV, BIG>(&e[j], (int)(bw & 0xffu), (int)(bw >> 8), yp, up, vp, W, H, layouts_g, masks_g, n, n_masks, srgb_and_ablate, tables, tiles_x, s_tab);
        }
    }
}

}  // namespace
