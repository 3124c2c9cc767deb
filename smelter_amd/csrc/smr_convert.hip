// smr_convert.hip — colour-format converters either side of the node texture.
//
// Replaces (one kernel per WGSL pass; each fragment = one thread-iteration):
//   wgpu/format/planar_yuv_to_rgba.{rs,wgsl}, nv12_to_rgba, interleaved_{uyvy,yuyv}_to_rgba,
//   bgra_to_rgba, argb_to_rgba, wgpu/utils/{add,remove}_premultiplied_alpha,
//   wgpu/format/rgba_to_yuv.{rs,wgsl} (3 passes fused into one launch), rgba_to_nv12,
//   wgpu/utils/r8_fill_with_color (black fallback).
//
// HBM-bound byte work: every thread handles 4 horizontally adjacent pixels so loads are
// 4 B (luma) and stores 16 B per lane; rows are pitched to 256 B so every row starts
// on a fresh cache line.
#include "smr_convert_dev.h"
#include "smr_convert_420.h"

namespace {

constexpr int BLOCK = 256;

// kind: 0 planar (3 R8 planes), 1 NV12 (R8 + RG8)
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_yuv_to_rgba(SurfView yp, SurfView up, SurfView vp, SurfView dst, int full) {
    const int groups = (dst.w + 3) >> 2;
    const int gx = blockIdx.x * BLOCK + threadIdx.x;
    const int y = blockIdx.y;
    if (gx >= groups) return;
    const int x0 = gx * 4;
    const float tv = ((float)y + 0.5f) / (float)dst.h;
    u32 px[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int x = x0 + i;
        if (x >= dst.w) { px[i] = 0; continue; }
        float tu = ((float)x + 0.5f) / (float)dst.w;
        float yy = (float)yp.ptr[(size_t)y * yp.pitch + x] / 255.0f;
        float uu, vv;
        if (KIND == 0) {
            uu = sample_plane_bilinear(up, 1, 0, tu, tv);
            vv = sample_plane_bilinear(vp, 1, 0, tu, tv);
        } else {
            uu = sample_plane_bilinear(up, 2, 0, tu, tv);
            vv = sample_plane_bilinear(up, 2, 1, tu, tv);
        }
        px[i] = yuv_to_rgb_px(yy, uu, vv, full != 0);
    }
    u8 *row = dst.ptr + (size_t)y * dst.pitch;
    if (x0 + 3 < dst.w) {
        *(uint4 *)(row + (size_t)x0 * 4) = make_uint4(px[0], px[1], px[2], px[3]);
    } else {
        for (int i = 0; i < 4 && x0 + i < dst.w; i++) *(u32 *)(row + (size_t)(x0 + i) * 4) = px[i];
    }
}

// The same pass for planar 4:2:0 (limited or full range), 4:2:2, 4:4:4 and NV12 frames, several frames per launch: what k_yuv_to_rgba
// computes, value for value, without its per-pixel coordinate arithmetic and divisions.  For a w x h frame with (w / 2) x (h / 2) chroma
// the sample position of luma column x in chroma texels is x / 2 - 0.25: the float evaluation ((x + .5) / w) * (w / 2) - .5 is off by
// less than 1e-3 for w <= 16384, so floor() and the 8-bit sub-texel fraction (subtexel(): a multiple of 1 / 256) come out as
// floor(x / 2 - .25) and exactly .75 (x even) / .25 (x odd); rows likewise.  A thread converts a 4 x 2 pixel block (columns 4 g .. + 3,
// rows 2 p, 2 p + 1) from chroma columns 2 g - 1 .. 2 g + 2 and rows p - 1 .. p + 1 (clamped like the sampler clamps), with
// sample_plane_bilinear's products and sums in its order, byte / 255 as unorm_of_byte (the IEEE quotient for every byte) and the
// matrix + store of yuv_to_rgb_px.  tests/test_gpu_parity.py holds it to the general kernel bit for bit.
// (ConvJob / ConvBatch: smr_convert_420.h)
__global__ __launch_bounds__(BLOCK) void k_yuv_to_rgba_batch(const ConvBatch B) {
    const ConvJob &J = B.j[blockIdx.z];
    const int g = blockIdx.x * 64 + (threadIdx.x & 63), p = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int w = J.dst.w, h = J.dst.h, cw = J.sx ? w >> 1 : w, ch = J.sy ? h >> 1 : h;
    if (4 * g >= w || 2 * p >= h) return;
    if (J.packed >= 3) {  // k_swizzle's byte permutes, 16 B per row and thread (rows are 16 B aligned: smr_frames_to_rgba_batch checks)
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int y = 2 * p + r;
            if (y >= h) break;
            const u8 *sr = J.yp.ptr + ((u32)y * J.yp.pitch + 16u * (u32)g);
            u8 *row = J.dst.ptr + ((u32)y * J.dst.pitch + 16u * (u32)g);
            u32 t[4];
            if (4 * g + 3 < w) {
                const uint4 q = *(const uint4 *)sr;
                t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) t[i] = 4 * g + i < w ? ((const u32 *)sr)[i] : 0u;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) t[i] = J.packed == 3 ? (t[i] & 0xff00ff00u) | ((t[i] & 0xffu) << 16) | ((t[i] >> 16) & 0xffu) : (t[i] >> 24) | (t[i] << 8);
            if (4 * g + 3 < w) {
                *(uint4 *)row = make_uint4(t[0], t[1], t[2], t[3]);
            } else {
                for (int i = 0; i < 4 && 4 * g + i < w; i++) ((u32 *)row)[i] = t[i];
            }
        }
        return;
    }
    if (J.packed) {
        // interleaved_{uyvy,yuyv}_to_rgba.wgsl:24-62 (k_interleaved422_to_rgba): pixel x takes the luma of its own half of group x / 2 and
        // the group's chroma, no interpolation — x_pos = floor(x + .5 - 2 / w + .0002) is x for every width from 8 up
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int y = 2 * p + r;
            if (y >= h) break;
            const u8 *sr = J.yp.ptr + ((u32)y * J.yp.pitch + 8u * (u32)g);
            u32 px[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (4 * g + i >= w) { px[i] = 0u; continue; }
                const u32 d = *(const u32 *)(sr + 4 * (i >> 1));
                const u32 c0 = d & 0xffu, c1 = (d >> 8) & 0xffu, c2 = (d >> 16) & 0xffu, c3 = d >> 24;
                const u32 ub = J.packed == 1 ? c0 : c1, vb = J.packed == 1 ? c2 : c3;
                const u32 yb = J.packed == 1 ? ((i & 1) ? c3 : c1) : ((i & 1) ? c2 : c0);
                px[i] = yuv_to_rgb_px_cr(unorm_of_byte(yb), unorm_of_byte(ub), unorm_of_byte(vb), false);
            }
            u8 *row = J.dst.ptr + ((u32)y * J.dst.pitch + 16u * (u32)g);
            if (4 * g + 3 < w) {
                *(uint4 *)row = make_uint4(px[0], px[1], px[2], px[3]);
            } else {
                for (int i = 0; i < 4 && 4 * g + i < w; i++) ((u32 *)row)[i] = px[i];
            }
        }
        return;
    }
    // chroma neighbourhood as unorm values: t[plane][row j = p - 1 + j][column 2 g - 1 + k], columns and rows clamped like the sampler
    // clamps.  The four bytes of a row come from two aligned dwords (three for NV12's interleaved pairs) starting at the window's first
    // existing column; the clamped columns are then byte moves: at the left edge (g = 0) the window is columns 0, 0, 1, 2, at the right
    // edge the last existing column repeats.  Planes whose rows are not dword-aligned or too tight take the bytes one by one.
    float t[2][3][4];
    // (a plane that is not subsampled along an axis is sampled at its texel centres: fraction 0 or, an ulp short of the centre, 1 with
    //  the window one texel earlier — a * 1 + b * 0 or a * 0 + b * 1: the texel itself either way, bit for bit)
    const int first = J.sx ? 2 * g - 1 : 4 * g;               // leftmost chroma column of the window (-1 for g = 0 of a subsampled row)
    const int first_ld = first < 0 ? 0 : first;               // ... that exists
    const int byte0 = J.nv ? 2 * first_ld : first_ld;         // its byte offset in the row
    const int base = byte0 & ~3;
    const u32 sh = (u32)(byte0 - base);
    const bool dwords = (J.up.pitch & 3u) == 0 && (((uintptr_t)J.up.ptr) & 3) == 0 && (J.vp.pitch & 3u) == 0 && (((uintptr_t)J.vp.ptr) & 3) == 0 &&
                        (u32)base + (J.nv ? 12u : 8u) <= J.up.pitch && (J.nv || (u32)base + 8u <= J.vp.pitch);
    const int nvalid = cw - first;                            // window columns 0 .. nvalid - 1 are at or left of the last column (>= 2)
    const u32 right_fix = nvalid >= 4 || !J.sx ? 0x03020100u : nvalid == 3 ? 0x02020100u : 0x01010100u;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int cy = clampi(J.sy ? p - 1 + j : 2 * p + j, 0, ch - 1);  // (full-height chroma: rows 2 p, 2 p + 1; the third is not used)
        const u8 *ur = J.up.ptr + (u32)cy * J.up.pitch, *vr = J.vp.ptr + (u32)cy * J.vp.pitch;  // (a plane is far below 4 GiB)
        if (dwords) {  // (uniform)
            u32 uw, vw;
            if (J.nv) {
                const u32 d0 = *(const u32 *)(ur + base), d1 = *(const u32 *)(ur + base + 4), d2 = *(const u32 *)(ur + base + 8);
                const u32 w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);  // U V U V of two columns each
                uw = __builtin_amdgcn_perm(w1, w0, 0x06040200u);
                vw = __builtin_amdgcn_perm(w1, w0, 0x07050301u);
            } else {
                uw = __builtin_amdgcn_alignbyte(*(const u32 *)(ur + base + 4), *(const u32 *)(ur + base), sh);
                vw = __builtin_amdgcn_alignbyte(*(const u32 *)(vr + base + 4), *(const u32 *)(vr + base), sh);
            }
            if (first < 0) {  // columns 0 1 2 3 -> 0 0 1 2
                uw = (uw << 8) | (uw & 0xffu);
                vw = (vw << 8) | (vw & 0xffu);
            }
            uw = __builtin_amdgcn_perm(0u, uw, right_fix);
            vw = __builtin_amdgcn_perm(0u, vw, right_fix);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                t[0][j][k] = unorm_of_byte((uw >> (8 * k)) & 0xffu);
                t[1][j][k] = unorm_of_byte((vw >> (8 * k)) & 0xffu);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int cx = clampi(first + k, 0, cw - 1);
                t[0][j][k] = unorm_of_byte(J.nv ? ur[2 * cx] : ur[cx]);
                t[1][j][k] = unorm_of_byte(J.nv ? ur[2 * cx + 1] : vr[cx]);
            }
        }
    }
    // horizontal lerps of the three chroma rows at the four luma columns: a * (1 - fx) + b * fx with fx = .75, .25, .75, .25
    float H[2][3][4];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            H[c][j][0] = J.sx ? t[c][j][0] * 0.25f + t[c][j][1] * 0.75f : t[c][j][0];
            H[c][j][1] = J.sx ? t[c][j][1] * 0.75f + t[c][j][2] * 0.25f : t[c][j][1];
            H[c][j][2] = J.sx ? t[c][j][1] * 0.25f + t[c][j][2] * 0.75f : t[c][j][2];
            H[c][j][3] = J.sx ? t[c][j][2] * 0.75f + t[c][j][3] * 0.25f : t[c][j][3];
        }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int y = 2 * p + r;
        if (y >= h) break;
        const u8 *yr = J.yp.ptr + ((u32)y * J.yp.pitch + 4u * (u32)g);
        const bool whole = 4 * g + 3 < w && (((uintptr_t)yr) & 3) == 0;
        const u32 y4 = whole ? *(const u32 *)yr : 0u;
        u32 px[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (4 * g + i >= w) { px[i] = 0u; continue; }
            const float yy = unorm_of_byte(whole ? (y4 >> (8 * i)) & 0xffu : (u32)yr[i]);
            // row 2 p: chroma rows (p - 1, p), fy = .75; row 2 p + 1: rows (p, p + 1), fy = .25 — top * (1 - fy) + bot * fy
            float uu = r == 0 ? H[0][0][i] * 0.25f + H[0][1][i] * 0.75f : H[0][1][i] * 0.75f + H[0][2][i] * 0.25f;
            float vv = r == 0 ? H[1][0][i] * 0.25f + H[1][1][i] * 0.75f : H[1][1][i] * 0.75f + H[1][2][i] * 0.25f;
            if (!J.sy) { uu = H[0][r][i]; vv = H[1][r][i]; }
            px[i] = yuv_to_rgb_px_cr(yy, uu, vv, J.full != 0);
        }
        u8 *row = J.dst.ptr + ((u32)y * J.dst.pitch + 16u * (u32)g);
        if (4 * g + 3 < w) {
            *(uint4 *)row = make_uint4(px[0], px[1], px[2], px[3]);
        } else {
            for (int i = 0; i < 4 && 4 * g + i < w; i++) ((u32 *)row)[i] = px[i];
        }
    }
}

// interleaved_{uyvy,yuyv}_to_rgba.wgsl:24-62. src is the (w/2) x h RGBA8 packed texture.
__global__ __launch_bounds__(BLOCK) void k_interleaved422_to_rgba(SurfView src, SurfView dst, int order) {
    const int x = blockIdx.x * BLOCK + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dst.w) return;
    const int tw = src.w;
    float dimx = (float)tw;
    float half_pixel_width = 0.5f / dimx;
    float tcx = ((float)x + 0.5f) / (float)dst.w;
    float xf = (tcx * dimx - half_pixel_width + 0.0001f) * 2.0f;
    unsigned x_pos = (unsigned)xf;
    int tx = clampi((int)(x_pos / 2), 0, tw - 1);
    u32 t = *(const u32 *)(src.ptr + (size_t)y * src.pitch + (size_t)tx * 4);
    float c0 = (float)(t & 0xff) / 255.0f, c1 = (float)((t >> 8) & 0xff) / 255.0f;
    float c2 = (float)((t >> 16) & 0xff) / 255.0f, c3 = (float)(t >> 24) / 255.0f;
    float yy, uu, vv;
    if (order == 0) { uu = c0; vv = c2; yy = (x_pos % 2 != 0) ? c3 : c1; }
    else            { uu = c1; vv = c3; yy = (x_pos % 2 != 0) ? c2 : c0; }
    *(u32 *)(dst.ptr + (size_t)y * dst.pitch + (size_t)x * 4) = yuv_to_rgb_px(yy, uu, vv, false);
}

// bgra_to_rgba.wgsl:24-28 (sample.bgra) / argb_to_rgba.wgsl:24-28 (sample.argb): byte permutes.
__global__ __launch_bounds__(BLOCK) void k_swizzle(SurfView src, SurfView dst, int kind) {
    const int groups = (dst.w + 3) >> 2;
    const int gx = blockIdx.x * BLOCK + threadIdx.x;
    const int y = blockIdx.y;
    if (gx >= groups) return;
    const int x0 = gx * 4;
    const u8 *srow = src.ptr + (size_t)y * src.pitch;
    u8 *drow = dst.ptr + (size_t)y * dst.pitch;
    u32 in[4], o[4];
    if (x0 + 3 < dst.w) {
        uint4 v = *(const uint4 *)(srow + (size_t)x0 * 4);
        in[0] = v.x; in[1] = v.y; in[2] = v.z; in[3] = v.w;
    } else {
        for (int i = 0; i < 4; i++) in[i] = (x0 + i < dst.w) ? *(const u32 *)(srow + (size_t)(x0 + i) * 4) : 0;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u32 p = in[i];
        if (kind == 0) o[i] = (p & 0xff00ff00u) | ((p & 0xff) << 16) | ((p >> 16) & 0xff);  // [x2,x1,x0,x3]
        else o[i] = (p >> 24) | (p << 8);                                                    // [x3,x0,x1,x2]
    }
    if (x0 + 3 < dst.w) {
        *(uint4 *)(drow + (size_t)x0 * 4) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
        for (int i = 0; i < 4 && x0 + i < dst.w; i++) *(u32 *)(drow + (size_t)(x0 + i) * 4) = o[i];
    }
}

// add_premultiplied_alpha.wgsl:24-35 / remove_premultiplied_alpha.wgsl:24-35
// mode 0: add (pxi decides sRGB vs unorm), mode 1: remove (unorm only)
__global__ __launch_bounds__(BLOCK) void k_premult(SurfView src, SurfView dst, int pxi, int mode, const float *__restrict__ tables) {
    const int x = blockIdx.x * BLOCK + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dst.w) return;
    const float *dec = tables, *thr = tables + 256;
    u32 p = *(const u32 *)(src.ptr + (size_t)y * src.pitch + (size_t)x * 4);
    u32 c[3] = {p & 0xff, (p >> 8) & 0xff, (p >> 16) & 0xff};
    float a = (float)(p >> 24) / 255.0f;
    float am = a > 0.00001f ? a : 0.00001f;
    u32 o[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (mode == 0) {
            float v = (pxi == PXI_RGBA8_SRGB) ? dec[c[k]] : (float)c[k] / 255.0f;
            v = clampf(v * am, 0.0f, 1.0f);
            o[k] = (pxi == PXI_RGBA8_SRGB) ? srgb_encode8(v, thr) : unorm8(v);
        } else {
            o[k] = unorm8(((float)c[k] / 255.0f) / am);
        }
    }
    *(u32 *)(dst.ptr + (size_t)y * dst.pitch + (size_t)x * 4) = o[0] | (o[1] << 8) | (o[2] << 16) | (unorm8(a) << 24);
}

// Y plane: exact texel fetch (target size == source size), 4 px per thread.
__global__ __launch_bounds__(BLOCK) void k_rgba_to_y(SurfView src, SurfView yp) {
    const int groups = (yp.w + 3) >> 2;
    const int gx = blockIdx.x * BLOCK + threadIdx.x;
    const int y = blockIdx.y;
    if (gx >= groups) return;
    const int x0 = gx * 4;
    const u8 *srow = src.ptr + (size_t)y * src.pitch;
    u32 out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int x = x0 + i;
        if (x >= yp.w) break;
        u32 p = *(const u32 *)(srow + (size_t)x * 4);
        float4 c = make_float4((float)(p & 0xff) / 255.0f, (float)((p >> 8) & 0xff) / 255.0f,
                               (float)((p >> 16) & 0xff) / 255.0f, (float)(p >> 24) / 255.0f);
        out |= unorm8(yuv_component(c, 0)) << (8 * i);
    }
    u8 *drow = yp.ptr + (size_t)y * yp.pitch;
    if (x0 + 3 < yp.w) *(u32 *)(drow + x0) = out;
    else for (int i = 0; i < 4 && x0 + i < yp.w; i++) drow[x0 + i] = (u8)(out >> (8 * i));
}

// chroma: one bilinear tap at the chroma texel centre (=> 2x2 mean for 4:2:0).
// NV = 0: separate U, V planes (R8); NV = 1: interleaved UV plane (RG8)
template <int NV>
__global__ __launch_bounds__(BLOCK) void k_rgba_to_chroma(SurfView src, SurfView up, SurfView vp, int cw, int ch) {
    const int x = blockIdx.x * BLOCK + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cw || y >= ch) return;
    float4 c = sample_rgba_bilinear(src, PXI_RGBA8_UNORM, ((float)x + 0.5f) / (float)cw, ((float)y + 0.5f) / (float)ch, nullptr);
    u32 u = unorm8(yuv_component(c, 1)), v = unorm8(yuv_component(c, 2));
    if (NV == 0) {
        up.ptr[(size_t)y * up.pitch + x] = (u8)u;
        vp.ptr[(size_t)y * vp.pitch + x] = (u8)v;
    } else {
        *(u16 *)(up.ptr + (size_t)y * up.pitch + (size_t)x * 2) = (u16)(u | (v << 8));
    }
}

// 4:2:0 frames (planar or NV12, limited or full range) the 4 x 4 block converter takes (smr_convert_420.h): one launch for up to 16 frames.
//
// The launch is PERSISTENT and its work is cut into EQUAL SHARES: as many workgroups as the device holds at once (the host sizes the grid),
// and wave w computes the w-th share of the launch's unit sequence (ConvBatch: block rows fastest, then column blocks, then jobs): a
// vertical run of blocks in one column block, or the end of one and the start of the next — cv420_run keeps the chroma rows neighbouring
// blocks share and requests block P + 1 before it computes block P; the tables are built once per resident workgroup instead of once per
// 16 rows.  What it bought, measured (profiles/r05_convert_waves.txt): 8 % fewer vector instructions per launch (9.65 M against 10.47 M), the
// same 22 - 23 us in the kernel trace as round 4's grid of one block per thread (22.8 us) — the wave stamps show why: the kernel is bound by
// the issue of its vector instructions (a SIMD's waves finish one after the other, oldest first; no traffic at all: as slow), mostly the
// byte extracts, packs, conversions and table-address arithmetic around the float arithmetic — and ~2 % more frames per second with two
// frames in flight.  (A ticket counter instead of fixed shares was measured first: one word serves ~90 atomics per microsecond — 150 us
// for this launch's 8 640 tickets.)
#ifdef CV_TIMING
__device__ unsigned long long g_cv_stamps[32768][10];  // per wave of the last launch: entry (shader clock), tables built, first task: loads in, window converted, first block done; queue dry, stores done; [7] / [8] = realtime at entry / exit, [9] = XCC_ID << 32 | HW_ID
extern "C" __attribute__((visibility("default"))) int smr_debug_convert_stamps(unsigned long long *out, unsigned n_waves) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cv_stamps), (size_t)n_waves * 10 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

template <bool NV>
#ifndef SMR_CONVERT_MIN_WAVES
#define SMR_CONVERT_MIN_WAVES 1  // waves per SIMD the converter's register allocation must leave room for (A/B knob: 8 = at most 64 registers, so that more of its waves fit beside another lane's resampler)
#endif
__global__ __launch_bounds__(BLOCK, SMR_CONVERT_MIN_WAVES) void k_yuv420_to_rgba(const ConvBatch B) {
    __shared__ float s_ylut[256], s_nlut[256];
#ifdef SMR_PRIO_CONVERT
    __builtin_amdgcn_s_setprio(SMR_PRIO_CONVERT);
#endif
#ifdef CV_TIMING
    unsigned long long *st = g_cv_stamps[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 32767u];
    if ((threadIdx.x & 63) == 0) { st[7] = __builtin_amdgcn_s_memrealtime(); st[1] = st[2] = st[3] = st[4] = st[5] = st[6] = 0; }
    CV_STAMP(st, 0, "s_nop 0");
#else
    unsigned long long *st = nullptr;
#endif
    const u32 lane = threadIdx.x & 63u;
    // both ranges' luma tables: a wave's share may touch several jobs (the full-range luma value of a byte is byte / 255 itself: s_nlut)
    s_ylut[threadIdx.x & 255] = cv420_luma_of_byte(threadIdx.x & 255u, false);
    s_nlut[threadIdx.x & 255] = unorm_of_byte(threadIdx.x & 255u);
    __syncthreads();
    CV_STAMP(st, 1, "s_nop 0");
    cv420_share<NV>(B, blockIdx.x, cv_uniform(threadIdx.x >> 6), gridDim.x, lane, s_ylut, s_nlut, st);
#ifdef CV_TIMING
    st = g_cv_stamps[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 32767u];
#endif
    CV_STAMP(st, 5, "s_nop 0");                // the queue is dry, the last stores are issued
    CV_STAMP(st, 6, "s_waitcnt vmcnt(0)");     // ... and acknowledged
#ifdef CV_TIMING
    if ((threadIdx.x & 63) == 0) { st[8] = __builtin_amdgcn_s_memrealtime(); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); st[9] = ((unsigned long long)xcc << 32) | hw; }
#endif
}

// The same for frames with a chroma plane whose rows fill its pitch and that the library did not allocate (conv_420_ok: a wrapped decoder
// surface with tight rows): cv420_load_chroma's TIGHT build requests nothing behind a window's last column.  A kernel of its own, so that the
// plain one stays the instruction stream it was measured as.
template <bool NV>
__global__ __launch_bounds__(BLOCK) void k_yuv420_to_rgba_tight(const ConvBatch B) {
    __shared__ float s_ylut[256], s_nlut[256];
#ifdef SMR_PRIO_CONVERT
    __builtin_amdgcn_s_setprio(SMR_PRIO_CONVERT);
#endif
#ifdef CV_TIMING
    unsigned long long *st = g_cv_stamps[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 32767u];
    if ((threadIdx.x & 63) == 0) { st[7] = __builtin_amdgcn_s_memrealtime(); st[1] = st[2] = st[3] = st[4] = st[5] = st[6] = 0; }
    CV_STAMP(st, 0, "s_nop 0");
#else
    unsigned long long *st = nullptr;
#endif
    const u32 lane = threadIdx.x & 63u;
    // both ranges' luma tables: a wave's share may touch several jobs (the full-range luma value of a byte is byte / 255 itself: s_nlut)
    s_ylut[threadIdx.x & 255] = cv420_luma_of_byte(threadIdx.x & 255u, false);
    s_nlut[threadIdx.x & 255] = unorm_of_byte(threadIdx.x & 255u);
    __syncthreads();
    CV_STAMP(st, 1, "s_nop 0");
    cv420_share<NV, true>(B, blockIdx.x, cv_uniform(threadIdx.x >> 6), gridDim.x, lane, s_ylut, s_nlut, st);
#ifdef CV_TIMING
    st = g_cv_stamps[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 32767u];
#endif
    CV_STAMP(st, 5, "s_nop 0");                // the queue is dry, the last stores are issued
    CV_STAMP(st, 6, "s_waitcnt vmcnt(0)");     // ... and acknowledged
#ifdef CV_TIMING
    if ((threadIdx.x & 63) == 0) { st[8] = __builtin_amdgcn_s_memrealtime(); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); st[9] = ((unsigned long long)xcc << 32) | hw; }
#endif
}

// rgba_to_yuv.wgsl's three passes (k_rgba_to_y + k_rgba_to_chroma) in one launch for even-sized frames: a thread owns a 4 x 2 pixel block,
// reads its eight texels once and writes the luma dwords and its share of the chroma planes.  Same values bit for bit: channels are
// byte / 255 (unorm_of_byte), a chroma sample of a subsampled axis sits exactly between two texels (sub-texel fraction 128 / 256 for
// sizes up to 8192: a * .5 + b * .5, then rows likewise), one of a full-resolution axis on a texel (x * 1 + y * 0), and the plane
// formulas + store are yuv_byte (smr_convert_dev.h: yuv_component + unorm8 without the clamps that cannot act on values in [0, 1]).
// sx / sy: chroma subsampling (4:2:0 = 1, 1; 4:2:2 = 1, 0; 4:4:4 = 0, 0); nv: interleaved chroma in `up` (NV12).
__global__ __launch_bounds__(BLOCK) void k_rgba_to_planes(SurfView src, SurfView yp, SurfView up, SurfView vp, int sx, int sy, int nv) {
    const int g = blockIdx.x * 64 + (threadIdx.x & 63), p = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int w = src.w, h = src.h;
    if (4 * g >= w || 2 * p >= h) return;
    const int nx = min(4, w - 4 * g), ny = min(2, h - 2 * p);
    float c[2][4][3];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int y = min(2 * p + r, h - 1);
        const u8 *row = src.ptr + ((u32)y * src.pitch + 16u * (u32)g);
        u32 t[4];
        if (nx == 4) {
            const uint4 q = *(const uint4 *)row;
            t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) t[i] = ((const u32 *)row)[min(i, nx - 1)];
        }
        u32 yq = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c[r][i][0] = unorm_of_byte(t[i] & 0xffu); c[r][i][1] = unorm_of_byte((t[i] >> 8) & 0xffu); c[r][i][2] = unorm_of_byte((t[i] >> 16) & 0xffu);
            yq |= yuv_byte(c[r][i][0], c[r][i][1], c[r][i][2], 0) << (8 * i);
        }
        if (r < ny) {
            u8 *d = yp.ptr + ((u32)(2 * p + r) * yp.pitch + 4u * (u32)g);
            if (nx == 4) *(u32 *)d = yq;
            else for (int i = 0; i < nx; i++) d[i] = (u8)(yq >> (8 * i));
        }
    }
    // chroma samples of this block: (sx ? 2 : 4) columns x (sy ? 1 : 2) rows
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (sy && r == 1) break;
        if (r >= ny) break;
        u32 uq = 0, vq = 0;
        const int n = sx ? 2 : 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k >= n) break;
            float m[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                if (sx) {
                    const float top = c[r][2 * k][ch] * 0.5f + c[r][2 * k + 1][ch] * 0.5f;
                    m[ch] = sy ? top * 0.5f + (c[1][2 * k][ch] * 0.5f + c[1][2 * k + 1][ch] * 0.5f) * 0.5f : top;
                } else {
                    m[ch] = c[r][k][ch];
                }
            }
            uq |= yuv_byte(m[0], m[1], m[2], 1) << (8 * k);
            vq |= yuv_byte(m[0], m[1], m[2], 2) << (8 * k);
        }
        const int cy = sy ? p : 2 * p + r, cx0 = sx ? 2 * g : 4 * g, cn = sx ? (nx + 1) / 2 : nx;  // chroma row, first column, columns of this block
        if (nv) {
            u8 *d = up.ptr + ((u32)cy * up.pitch + 2u * (u32)cx0);
            const u32 uv = (uq & 0xffu) | ((vq & 0xffu) << 8) | ((uq & 0xff00u) << 8) | ((vq & 0xff00u) << 16);
            if (cn == 2) *(u32 *)d = uv;
            else *(u16 *)d = (u16)uv;
        } else {
            u8 *du = up.ptr + ((u32)cy * up.pitch + (u32)cx0), *dv = vp.ptr + ((u32)cy * vp.pitch + (u32)cx0);
            if (cn == 4) { *(u32 *)du = uq; *(u32 *)dv = vq; }
            else if (cn == 2 && sx) { *(u16 *)du = (u16)uq; *(u16 *)dv = (u16)vq; }
            else for (int i = 0; i < cn; i++) { du[i] = (u8)(uq >> (8 * i)); dv[i] = (u8)(vq >> (8 * i)); }
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_fill_bytes(SurfView p, int row_bytes, u32 value4) {
    const int x = (blockIdx.x * BLOCK + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (x >= row_bytes) return;
    u8 *row = p.ptr + (size_t)y * p.pitch;
    if (x + 3 < row_bytes) *(u32 *)(row + x) = value4;
    else for (int i = 0; x + i < row_bytes; i++) row[x + i] = (u8)(value4 >> (8 * i));
}

inline dim3 grid_px(int w_items, int h) { return dim3((unsigned)((w_items + BLOCK - 1) / BLOCK), (unsigned)h, 1); }

u32 host_unorm8(float x) {
    if (!(x > 0.0f)) x = 0.0f;
    if (x > 1.0f) x = 1.0f;
    return (u32)(int)(x * 255.0f + 0.5f);
}

}  // namespace


static int smr_frame_to_rgba_general(smr_ctx *ctx, const smr_frame *in, smr_surface *node);  // one launch per frame, any geometry

// k_yuv_to_rgba_batch's frames: planar 4:2:0 / 4:2:2 / 4:4:4 and NV12 whose subsampled axes are even (chroma planes exactly half), within the
// size the coordinate argument holds for
static bool conv_batchable(const smr_ctx *ctx, const smr_frame *in) {
    if (in->width < 2 || in->height < 2 || in->width > 16384 || in->height > 16384 || ctx->convert_impl == SMR_CONVERT_GENERAL) return false;
    if (in->format == SMR_FRAME_UYVY422 || in->format == SMR_FRAME_YUYV422)
        return in->width % 2 == 0 && in->width >= 8 && in->planes[0] && (in->planes[0]->pitch & 3u) == 0 && (((uintptr_t)in->planes[0]->ptr) & 3) == 0;
    if (in->format == SMR_FRAME_BGRA || in->format == SMR_FRAME_ARGB)
        return in->planes[0] && (in->planes[0]->pitch & 15u) == 0 && (((uintptr_t)in->planes[0]->ptr) & 15) == 0;
    const bool nv = in->format == SMR_FRAME_NV12;
    const bool sx = in->format != SMR_FRAME_PLANAR_YUV444, sy = in->format == SMR_FRAME_PLANAR_YUV420 || in->format == SMR_FRAME_PLANAR_YUVJ420 || nv;
    if (in->format > SMR_FRAME_PLANAR_YUVJ420 && !nv) return false;  // (BGRA / ARGB / RGBA: byte permutes of their own)
    if ((sx && in->width % 2) || (sy && in->height % 2)) return false;
    return in->planes[0] && in->planes[1] && (nv || in->planes[2]);
}

// k_yuv420_to_rgba's frames (cv420_block, smr_convert_420.h): 4:2:0 planar / NV12, width a multiple of 4 from 8, even height, every plane
// dword-aligned.  The plain kernel reads up to a dword past the last block's window; the bytes of that dword are never used (the window's
// last column is the plane's edge, repeated), but they are read: a plane the library allocated may be read past a row's end — the next row,
// or the 16 bytes every allocation ends with (SMR_SURFACE_TAIL) — a wrapped one is read inside its rows' bytes only: it takes
// k_yuv420_to_rgba_tight, which requests nothing behind a window's last column (tight rows — a decoder's pitch == bytes per row — or wide ones).  -> 0: not a frame for the block converter | 1: the plain kernel | 2: the tight one
static int conv_420_mode(const smr_ctx *ctx, const smr_frame *in) {
    if (ctx->convert_impl != SMR_CONVERT_AUTO) return 0;
    const bool nv = in->format == SMR_FRAME_NV12;
    if (in->format != SMR_FRAME_PLANAR_YUV420 && in->format != SMR_FRAME_PLANAR_YUVJ420 && !nv) return 0;
    if (in->width % 4 || in->width < 8 || in->height % 2 || in->height < 2 || in->width > 16384 || in->height > 16384) return 0;
    auto dwords = [](const smr_surface *s) { return s && (s->pitch & 3u) == 0 && (((uintptr_t)s->ptr) & 3) == 0; };
    if (!dwords(in->planes[0]) || !dwords(in->planes[1]) || (!nv && !dwords(in->planes[2]))) return 0;
    const u32 cw = in->width / 2;
    // the last block's window starts at chroma column cw - 3: its bytes begin in the dword at ((cw - 3) [* 2]) & ~3 and the loads reach 8 (12) bytes from there
    const u32 need = nv ? ((2u * (cw - 3u)) & ~3u) + 12u : ((cw - 3u) & ~3u) + 8u;
    const u32 row = nv ? 2u * cw : cw;
    if (in->planes[0]->pitch < in->width || in->planes[1]->pitch < row || (!nv && in->planes[2]->pitch < row)) return 0;
    // (a wrapped plane with a wide pitch still ends with its last ROW for many producers — pitch * (h - 1) + row bytes, not pitch * h — and the plain
    //  kernel's reach into that row's padding would leave the allocation: every plane the library does not own takes the tight kernel, + 2 %)
    (void)need;
    auto reach_ok = [&](const smr_surface *s) { return s->owned; };
    return reach_ok(in->planes[1]) && (nv || reach_ok(in->planes[2])) ? 1 : 2;
}
static bool conv_420_ok(const smr_ctx *ctx, const smr_frame *in) { return conv_420_mode(ctx, in) != 0; }

bool smr_conv_rgb12_ok(const smr_ctx *ctx, const smr_frame *in) { return in && conv_420_ok(ctx, in) && 3u * in->width <= 7682u * 2u; }

// smr_frame_to_rgba for several frames: one launch for every 16 frames of a kind (k_yuv420_to_rgba planar / NV12, k_yuv_to_rgba_batch), the
// others one by one.  Everything queued is launched before the call returns, errors included.
int smr_frames_to_rgba_batch(smr_ctx *ctx, const smr_frame *const *in, smr_surface *const *nodes, u32 n, const u8 *rgb12) {
    if (!ctx || (n && (!in || !nodes))) return SMR_ERR_INVALID;
    // validate first, enqueue afterwards: no frame is left half-queued by a bad one behind it
    for (u32 i = 0; i < n; i++) {
        if (!in[i] || !nodes[i]) return SMR_ERR_INVALID;
        if (rgb12 && rgb12[i]) {
            if (nodes[i]->fmt != SMR_PX_R8 || nodes[i]->w != 3 * in[i]->width || nodes[i]->h != in[i]->height || (nodes[i]->pitch & 3u) || (((uintptr_t)nodes[i]->ptr) & 3) ||
                !conv_420_ok(ctx, in[i]))
                return smr_fail(ctx, SMR_ERR_INVALID, "smr_frame_to_rgba: an RGB12 node must be R8 %ux%u of a 4:2:0 frame the block converter takes", 3 * in[i]->width, in[i]->height);
        } else if (nodes[i]->fmt != SMR_PX_RGBA8 || nodes[i]->w != in[i]->width || nodes[i]->h != in[i]->height)
            return smr_fail(ctx, SMR_ERR_INVALID, "smr_frame_to_rgba: node surface must be RGBA8 %ux%u", in[i]->width, in[i]->height);
        if (int rc = smr_validate_frame(ctx, in[i], "smr_frame_to_rgba")) return rc;
    }
    struct Queue {
        ConvBatch B;
        u32 nb = 0;
        int mw = 0, mh = 0;
    };
    Queue q[5];  // 0: k_yuv_to_rgba_batch | 1: k_yuv420_to_rgba planar | 2: ... NV12 | 3, 4: k_yuv420_to_rgba_tight planar, NV12
    auto flush = [&](int k) -> int {
        Queue &Q = q[k];
        if (!Q.nb) return SMR_OK;
        StageScope scope(ctx, SMR_STAGE_INGEST);
        ctx->kernel_launches[SMR_KERNEL_FRAME_TO_RGBA]++;  // (per launch)
        if (k != 0) ctx->kernel_launches[SMR_KERNEL_FRAME_TO_RGBA_420]++;
        if (k == 0) hipLaunchKernelGGL(k_yuv_to_rgba_batch, dim3((unsigned)((Q.mw + 255) / 256), (unsigned)((Q.mh + 7) / 8), Q.nb), dim3(BLOCK), 0, ctx->stream, Q.B);
        else {
            // as many workgroups as stay resident together (6 per CU at 77 registers: convert_wg_per_cu), never more waves than units.  A dynamic LDS
            // request that caps a CU at exactly that number (an even spread: every SIMD the same number of waves) was measured and is no
            // faster in the kernel trace (6 per CU: 23.1 us capped, 22.0 us uncapped, 4 per CU: 23.3 / 23.7 — profiles/r05_convert_waves.txt)
            // while it keeps other launches' workgroups off the CU; the knob remains for laboratory builds (SMR_CONVERT_LDS_PAD)
            u32 total = 0;
            for (u32 j = 0; j < Q.nb; j++) total += (((u32)Q.B.j[j].dst.w + 255u) / 256u) * (((u32)Q.B.j[j].dst.h + 3u) / 4u);
            // convert_wg_per_cu 0 (the default): five per CU for a batch of a few frames, six for a large one.  With two frames in flight the
            // kernel runs beside the other lane's kernels; configs[2] (8 640 units: 1.4 per wave at six) gains 2 % at five, configs[3]
            // (64 800 units) loses 1 % (profiles/r06_sensitivity.txt section 9)
            const int want = ctx->convert_wg_per_cu > 0 ? ctx->convert_wg_per_cu : (total >= 32768u ? 6 : 5);
            const u32 per_cu = (u32)(want > 6 ? 6 : want);
            u32 blocks = (u32)ctx->cu_count * per_cu;
            if (blocks > (total + 3u) / 4u) blocks = (total + 3u) / 4u;
            // the partition (smr_convert_420.h): bands of block rows dealt to the XCDs, equal shares per wave inside an XCD
            const u32 bands = cv420_plan(Q.B, (int)Q.nb, blocks * 4u);
            if (blocks < (bands < 8u ? bands : 8u)) blocks = bands < 8u ? bands : 8u;  // (every XCD that owns a band runs a workgroup)
            const u32 lds_pad = ctx->convert_lds_pad;
            if (k == 1) hipLaunchKernelGGL(k_yuv420_to_rgba<false>, dim3(blocks), dim3(BLOCK), lds_pad, ctx->stream, Q.B);
            else if (k == 2) hipLaunchKernelGGL(k_yuv420_to_rgba<true>, dim3(blocks), dim3(BLOCK), lds_pad, ctx->stream, Q.B);
            else if (k == 3) hipLaunchKernelGGL(k_yuv420_to_rgba_tight<false>, dim3(blocks), dim3(BLOCK), lds_pad, ctx->stream, Q.B);
            else hipLaunchKernelGGL(k_yuv420_to_rgba_tight<true>, dim3(blocks), dim3(BLOCK), lds_pad, ctx->stream, Q.B);
        }
        Q.nb = 0; Q.mw = 0; Q.mh = 0;
        SMR_HIP(ctx, hipGetLastError());
        return SMR_OK;
    };
    auto flush_all = [&]() -> int {
        int rc = SMR_OK;
        for (int k = 0; k < 5; k++)
            if (int r = flush(k)) rc = rc ? rc : r;
        return rc;
    };
    for (u32 i = 0; i < n; i++) {
        const bool c12 = rgb12 && rgb12[i];
        const bool aligned16 = c12 || ((((uintptr_t)nodes[i]->ptr) & 15) == 0 && (nodes[i]->pitch & 15u) == 0);  // (the block kernels store 16 B)
        const bool nv = in[i]->format == SMR_FRAME_NV12;
        const int mode = conv_420_mode(ctx, in[i]);
        const int k = !aligned16 ? -1 : mode ? (nv ? 2 : 1) + (mode == 2 ? 2 : 0) : conv_batchable(ctx, in[i]) ? 0 : -1;
        if (k < 0) {
            if (int rc = smr_frame_to_rgba_general(ctx, in[i], nodes[i])) {
                (void)flush_all();
                return rc;
            }
            continue;
        }
        Queue &Q = q[k];
        ConvJob &J = Q.B.j[Q.nb++];
        const bool packed = in[i]->format == SMR_FRAME_UYVY422 || in[i]->format == SMR_FRAME_YUYV422 || in[i]->format == SMR_FRAME_BGRA || in[i]->format == SMR_FRAME_ARGB;
        J.yp = view_of(in[i]->planes[0]);
        J.up = packed ? J.yp : view_of(in[i]->planes[1]);
        J.vp = packed || nv ? J.up : view_of(in[i]->planes[2]);
        J.packed = in[i]->format == SMR_FRAME_UYVY422 ? 1 : in[i]->format == SMR_FRAME_YUYV422 ? 2 : in[i]->format == SMR_FRAME_BGRA ? 3 : in[i]->format == SMR_FRAME_ARGB ? 4 : 0;
        J.dst = view_of(nodes[i]);
        J.rgb12 = c12 ? 1 : 0;
        if (c12) J.dst.w = (int)in[i]->width;  // (in pixels: the kernel's geometry; the rows hold 3 w bytes)
        J.full = in[i]->format == SMR_FRAME_PLANAR_YUVJ420 ? 1 : 0;
        J.nv = nv ? 1 : 0;
        J.sx = in[i]->format != SMR_FRAME_PLANAR_YUV444 ? 1 : 0;
        J.sy = (in[i]->format == SMR_FRAME_PLANAR_YUV420 || in[i]->format == SMR_FRAME_PLANAR_YUVJ420 || nv) ? 1 : 0;
        Q.mw = (int)in[i]->width > Q.mw ? (int)in[i]->width : Q.mw;
        Q.mh = (int)in[i]->height > Q.mh ? (int)in[i]->height : Q.mh;
        if (Q.nb == MAX_CONV_JOBS)
            if (int rc = flush(k)) {
                (void)flush_all();
                return rc;
            }
    }
    return flush_all();
}

extern "C" {

int smr_frame_to_rgba(smr_ctx *ctx, const smr_frame *in, smr_surface *node) {
    SMR_ENTER(ctx);
    if (!ctx || !in || !node) return SMR_ERR_INVALID;
    return smr_frames_to_rgba_batch(ctx, &in, &node, 1);  // (one frame: the batch kernel where it applies, the general kernels elsewhere)
}

}  // extern "C"

static int smr_frame_to_rgba_general(smr_ctx *ctx, const smr_frame *in, smr_surface *node) {
    StageScope scope(ctx, SMR_STAGE_INGEST);
    ctx->kernel_launches[SMR_KERNEL_FRAME_TO_RGBA]++;
    SurfView dst = view_of(node);
    const int w = (int)in->width, h = (int)in->height;
    switch (in->format) {
    case SMR_FRAME_PLANAR_YUV420:
    case SMR_FRAME_PLANAR_YUV422:
    case SMR_FRAME_PLANAR_YUV444:
    case SMR_FRAME_PLANAR_YUVJ420: {
        if (!in->planes[1] || !in->planes[2]) return smr_fail(ctx, SMR_ERR_INVALID, "smr_frame_to_rgba: missing chroma plane");
        SurfView yp = view_of(in->planes[0]), up = view_of(in->planes[1]), vp = view_of(in->planes[2]);
        // planes keep a 1x1 placeholder when the logical chroma size is 0; sample with the logical size
        hipLaunchKernelGGL(k_yuv_to_rgba<0>, grid_px((w + 3) / 4, h), dim3(BLOCK), 0, ctx->stream, yp, up, vp, dst,
                           in->format == SMR_FRAME_PLANAR_YUVJ420 ? 1 : 0);
        break;
    }
    case SMR_FRAME_NV12: {
        if (!in->planes[1]) return smr_fail(ctx, SMR_ERR_INVALID, "smr_frame_to_rgba: missing UV plane");
        SurfView yp = view_of(in->planes[0]), up = view_of(in->planes[1]);
        hipLaunchKernelGGL(k_yuv_to_rgba<1>, grid_px((w + 3) / 4, h), dim3(BLOCK), 0, ctx->stream, yp, up, up, dst, 0);
        break;
    }
    case SMR_FRAME_UYVY422:
    case SMR_FRAME_YUYV422:
        hipLaunchKernelGGL(k_interleaved422_to_rgba, grid_px(w, h), dim3(BLOCK), 0, ctx->stream, view_of(in->planes[0]), dst,
                           in->format == SMR_FRAME_UYVY422 ? 0 : 1);
        break;
    case SMR_FRAME_BGRA:
    case SMR_FRAME_ARGB:
        hipLaunchKernelGGL(k_swizzle, grid_px((w + 3) / 4, h), dim3(BLOCK), 0, ctx->stream, view_of(in->planes[0]), dst,
                           in->format == SMR_FRAME_BGRA ? 0 : 1);
        break;
    case SMR_FRAME_RGBA:
        // input_texture/rgba_texture.rs:38-63: straight-alpha texture -> premultiplied node texture
        hipLaunchKernelGGL(k_premult, grid_px(w, h), dim3(BLOCK), 0, ctx->stream, view_of(in->planes[0]), dst,
                           ctx->srgb() ? PXI_RGBA8_SRGB : PXI_RGBA8_UNORM, 0, ctx->d_tables);
        break;
    default:
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_frame_to_rgba: unknown frame format %u", in->format);
    }
    SMR_HIP(ctx, hipGetLastError());
    return SMR_OK;
}

extern "C" {

static int premult_common(smr_ctx *ctx, const smr_surface *src, smr_surface *dst, int mode) {
    SMR_ENTER(ctx);
    if (!ctx || !src || !dst) return SMR_ERR_INVALID;
    if (src->fmt != SMR_PX_RGBA8 || dst->fmt != SMR_PX_RGBA8 || src->w != dst->w || src->h != dst->h)
        return smr_fail(ctx, SMR_ERR_INVALID, "premultiply: surfaces must be RGBA8 of equal size");
    StageScope scope(ctx, SMR_STAGE_INGEST);
    hipLaunchKernelGGL(k_premult, grid_px((int)dst->w, (int)dst->h), dim3(BLOCK), 0, ctx->stream, view_of(src), view_of(dst),
                       (ctx->srgb() && mode == 0) ? PXI_RGBA8_SRGB : PXI_RGBA8_UNORM, mode, ctx->d_tables);
    SMR_HIP(ctx, hipGetLastError());
    return SMR_OK;
}

int smr_add_premultiplied_alpha(smr_ctx *ctx, const smr_surface *src, smr_surface *dst) { return premult_common(ctx, src, dst, 0); }
int smr_remove_premultiplied_alpha(smr_ctx *ctx, const smr_surface *src, smr_surface *dst) { return premult_common(ctx, src, dst, 1); }

int smr_rgba_to_frame(smr_ctx *ctx, const smr_surface *node, const smr_frame *out) {
    SMR_ENTER(ctx);
    if (!ctx || !node || !out) return SMR_ERR_INVALID;
    if (node->fmt != SMR_PX_RGBA8 || node->w != out->width || node->h != out->height)
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_rgba_to_frame: node must be RGBA8 %ux%u", out->width, out->height);
    if (int rc = smr_validate_frame(ctx, out, "smr_rgba_to_frame")) return rc;
    StageScope scope(ctx, SMR_STAGE_OUTPUT);
    const int w = (int)out->width, h = (int)out->height;
    SurfView src = view_of(node);
    int cw, ch;
    switch (out->format) {
    case SMR_FRAME_RGBA:
        // OutputTexture::Rgba8UnormWgpuTexture (render_loop.rs:81-103): the frame is a clone of the root node's texture, as it is
        SMR_HIP(ctx, hipMemcpy2DAsync(out->planes[0]->ptr, out->planes[0]->pitch, node->ptr, node->pitch, (size_t)w * 4, (size_t)h,
                                      hipMemcpyDeviceToDevice, ctx->stream));
        return SMR_OK;
    case SMR_FRAME_PLANAR_YUV420: cw = w / 2; ch = h / 2; break;
    case SMR_FRAME_PLANAR_YUV422: cw = w / 2; ch = h; break;
    case SMR_FRAME_PLANAR_YUV444: cw = w; ch = h; break;
    case SMR_FRAME_NV12: cw = w / 2; ch = h / 2; break;
    default:
        // output_texture.rs:26-38 only offers 420/422/444 planar, RGBA and NV12 outputs
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_rgba_to_frame: unsupported output frame format %u", out->format);
    }
    if (!out->planes[0] || !out->planes[1]) return smr_fail(ctx, SMR_ERR_INVALID, "smr_rgba_to_frame: missing planes");
    {
        // one launch for frames whose subsampled axes are even and whose planes take dword / word stores (every surface this library allocates)
        const bool nv = out->format == SMR_FRAME_NV12;
        const int sx = out->format != SMR_FRAME_PLANAR_YUV444, sy = out->format == SMR_FRAME_PLANAR_YUV420 || nv;
        const smr_surface *pu = out->planes[1], *pv = nv ? out->planes[1] : out->planes[2];
        auto aligned = [](const smr_surface *s) { return s && (s->pitch & 3u) == 0 && (((uintptr_t)s->ptr) & 3) == 0; };
        if (pv && (!sx || w % 2 == 0) && (!sy || h % 2 == 0) && w >= 2 && h >= 2 && w <= 8192 && h <= 8192 && aligned(out->planes[0]) && aligned(pu) && aligned(pv) &&
            (((uintptr_t)src.ptr) & 15) == 0 && (src.pitch & 15u) == 0 && ctx->convert_impl != SMR_CONVERT_GENERAL) {
            hipLaunchKernelGGL(k_rgba_to_planes, dim3((unsigned)((w + 255) / 256), (unsigned)((h + 7) / 8), 1), dim3(BLOCK), 0, ctx->stream, src,
                               view_of(out->planes[0]), view_of(pu), view_of(pv), sx, sy, nv ? 1 : 0);
            SMR_HIP(ctx, hipGetLastError());
            return SMR_OK;
        }
    }
    hipLaunchKernelGGL(k_rgba_to_y, grid_px((w + 3) / 4, h), dim3(BLOCK), 0, ctx->stream, src, view_of(out->planes[0]));
    if (cw > 0 && ch > 0) {
        if (out->format == SMR_FRAME_NV12) {
            SurfView uv = view_of(out->planes[1]);
            hipLaunchKernelGGL(k_rgba_to_chroma<1>, grid_px(cw, ch), dim3(BLOCK), 0, ctx->stream, src, uv, uv, cw, ch);
        } else {
            if (!out->planes[2]) return smr_fail(ctx, SMR_ERR_INVALID, "smr_rgba_to_frame: missing V plane");
            hipLaunchKernelGGL(k_rgba_to_chroma<0>, grid_px(cw, ch), dim3(BLOCK), 0, ctx->stream, src, view_of(out->planes[1]),
                               view_of(out->planes[2]), cw, ch);
        }
    }
    SMR_HIP(ctx, hipGetLastError());
    return SMR_OK;
}

int smr_frame_fill_black(smr_ctx *ctx, const smr_frame *out) {
    SMR_ENTER(ctx);
    if (ctx && out)
        if (int rc = smr_validate_frame(ctx, out, "smr_frame_fill_black")) return rc;
    if (!ctx || !out || !out->planes[0]) return SMR_ERR_INVALID;
    // RGBColor::BLACK.to_yuv(), smelter-render/src/scene/types.rs:28-41
    const float y = (0.0f * 0.85882354f) + (16.0f / 255.0f);
    const float c = ((0.0f + 0.5f) * 0.8784314f) + (16.0f / 255.0f);
    const u32 yb = host_unorm8(y), cb = host_unorm8(c);
    StageScope scope(ctx, SMR_STAGE_OUTPUT);
    if (out->format == SMR_FRAME_RGBA) {  // render_loop.rs:140-158: a fresh (all-zero) texture when the root is empty
        SMR_HIP(ctx, hipMemset2DAsync(out->planes[0]->ptr, out->planes[0]->pitch, 0, (size_t)out->width * 4, out->height, ctx->stream));
        return SMR_OK;
    }
    for (int i = 0; i < 3; i++) {
        const smr_surface *s = out->planes[i];
        if (!s) continue;
        u32 b = (i == 0 && (out->format <= SMR_FRAME_PLANAR_YUVJ420 || out->format == SMR_FRAME_NV12)) ? yb : cb;
        if (out->format > SMR_FRAME_PLANAR_YUVJ420 && out->format != SMR_FRAME_NV12)
            return smr_fail(ctx, SMR_ERR_INVALID, "smr_frame_fill_black: only planar YUV / NV12 frames");
        int row_bytes = (int)(s->w * bytes_per_px(s->fmt));
        hipLaunchKernelGGL(k_fill_bytes, grid_px((row_bytes + 3) / 4, (int)s->h), dim3(BLOCK), 0, ctx->stream, view_of(s),
                           row_bytes, b * 0x01010101u);
    }
    SMR_HIP(ctx, hipGetLastError());
    return SMR_OK;
}

}  // extern "C"
