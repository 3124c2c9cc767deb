// smr_fused.hip — the per-frame hot path: LayoutNode::render + read_outputs in two kernel waves.
//
// Reference sequence per frame (smelter-render/src/state/render_loop.rs:19-230,
// transformations/layout.rs:169-278): per input a YUV->RGBA pass, per scaled child two Lanczos
// passes through an Rgba16Float intermediate, N blended layout draws, three RGBA->YUV passes —
// every one a full round trip through memory (~470 MB of traffic for 8x1080p -> 4K, against
// 37 MB of algorithmic bytes).
//
// Here:
//   wave A  k_ingest_resample — for ALL scaled inputs in one launch: reads the raw Y/U/V planes,
//           converts + quantises + sRGB-decodes into LDS, horizontal Lanczos into an f16 LDS
//           intermediate, vertical Lanczos, sRGB-encodes and writes the dst-sized RGBA8 tile.
//           The node texture and the f16 intermediate never exist in HBM.
//   wave B  k_compose_output — classifies the layout list per 128x16 tile (untouched / solid /
//           opaque / general), composites 4x2 pixel blocks in registers starting at the last
//           opaque layer and writes Y, U, V (or NV12) directly; the RGBA8 output frame never
//           exists in HBM either.
// Every quantisation point of the reference pipeline (u8 node texture, f16 intermediate, u8 sRGB
// tile, u8 render target after each draw) is reproduced in registers, and the arithmetic is the
// same f32 sequence as the general kernels (smr_convert_dev.h, smr_resample_dev.h,
// smr_layout_dev.h); the only substitutions are exact ones (LUTs for u8 -> f32, correctly rounded
// division through a reciprocal + two FMAs, skipping layers that an opaque layer overwrites).
// Anything the fused kernels do not cover (single-pass plans, box pre-reduction, vertical-first
// plans, packed/NV12 inputs, odd output sizes, 4:2:2 / 4:4:4 outputs) falls back to the general
// kernels of smr_convert/smr_resample/smr_layout per layout, never to the CPU.
#include "smr_convert_dev.h"
#include "smr_layout_dev.h"
#include "smr_resample_dev.h"

#include <cmath>
#include <cstdlib>

namespace {

// a / b, correctly rounded, from rb = RN(1/b): q0 = RN(a*rb); r = a - q0*b (exact, FMA); q = RN(q0 + r*rb)
// (Markstein; holds for normal operands unless b's significand is all ones).
__device__ __forceinline__ float div_cr(float a, float b, float rb) {
    float q0 = a * rb;
    float r = __builtin_fmaf(-q0, b, a);
    return __builtin_fmaf(r, rb, q0);
}

// ------------------------------------------------------------------ weight tables (device cache)
__global__ __launch_bounds__(64) void k_build_weights(float scale, float offset, int taps, int n, int *__restrict__ first,
                                                      float *__restrict__ wsum, float *__restrict__ w) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float tmp[MAX_TAPS];
    float s;
    first[i] = lanczos_weights(i, scale, offset, taps, tmp, &s);
    wsum[i] = s;
    for (int t = 0; t < taps; t++) w[(size_t)i * taps + t] = tmp[t];
}

struct WeightPtrs {
    const int *first;
    const float *wsum;
    const float *w;
    int taps;
};

int host_taps(float scale) {
    float kernel_scale = scale > 1.0f ? scale : 1.0f;
    int taps = (int)ceilf(2.0f * (3.0f * kernel_scale)) + 1;
    return taps > MAX_TAPS ? MAX_TAPS : taps;
}

int get_weights(smr_ctx *ctx, float scale, float offset, int n, WeightPtrs *out) {
    const int taps = host_taps(scale);
    ctx->weight_clock++;
    smr_ctx::WeightTable *hit = nullptr, *victim = nullptr;
    for (auto &t : ctx->weight_tables) {
        if (t.dev && t.n == n && t.scale == scale && t.offset == offset) { hit = &t; break; }
    }
    if (!hit) {
        if (ctx->weight_tables.size() < 64) {
            ctx->weight_tables.emplace_back();
            victim = &ctx->weight_tables.back();
        } else {
            for (auto &t : ctx->weight_tables)
                if (!victim || t.last_use < victim->last_use) victim = &t;
        }
        const size_t need = (size_t)n * (2 + taps) * 4;
        if (victim->bytes < need) {
            if (victim->dev) {
                SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a queued kernel may still read it
                (void)hipFree(victim->dev);
                victim->dev = nullptr;
                victim->bytes = 0;
            }
            size_t want = (need + 4095) & ~(size_t)4095;
            SMR_HIP(ctx, hipMalloc(&victim->dev, want));
            victim->bytes = want;
        }
        victim->scale = scale; victim->offset = offset; victim->n = n; victim->taps = taps;
        int *first = (int *)victim->dev;
        float *wsum = (float *)victim->dev + n;
        float *w = (float *)victim->dev + 2 * (size_t)n;
        hipLaunchKernelGGL(k_build_weights, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, scale, offset, taps, n, first, wsum, w);
        SMR_HIP(ctx, hipGetLastError());
        hit = victim;
    }
    hit->last_use = ctx->weight_clock;
    out->first = (const int *)hit->dev;
    out->wsum = (const float *)hit->dev + n;
    out->w = (const float *)hit->dev + 2 * (size_t)n;
    out->taps = taps;
    return SMR_OK;
}

// ------------------------------------------------------------------ wave A: ingest + resample
constexpr int TW = 64;   // output tile width  (one lane per column)
constexpr int TH = 32;   // output tile height
constexpr int CH = 8;    // source rows converted + H-filtered per chunk
constexpr int A_THREADS = 256;

struct IngestJob {
    SurfView yp, up, vp;  // planar source planes (chroma views carry the logical chroma size)
    SurfView dst;         // RGBA8 tile, dst-sized
    int src_w, src_h;
    int full_range;
    int fast420;          // 4:2:0 with even luma size: 2x2-quad conversion path
    int taps_h, taps_v;
    const int *first_h; const float *wsum_h; const float *w_h;
    const int *first_v; const float *wsum_v; const float *w_v;
    int tiles_x, tiles_y;
    int nc_max, nr_max;
};

__device__ __forceinline__ float4 half4_to_float4(uint2 raw) {
    __half2 lo = *(const __half2 *)&raw.x, hi = *(const __half2 *)&raw.y;
    float2 a = __half22float2(lo), b = __half22float2(hi);
    return make_float4(a.x, a.y, b.x, b.y);
}

// rest of planar_yuv_to_rgba.wgsl:53-57 once y,u,v are range-expanded: matrix, clamp, unorm8 store, then the
// node texture's sRGB view decode (LUT) -> linear RGB
__device__ __forceinline__ float4 yuv_expanded_to_linear(float y, float u, float v, const float *__restrict__ s_dec) {
    float r = y + 1.5748f * (v - 0.5f);
    float g = y - 0.1873f * (u - 0.5f) - 0.4681f * (v - 0.5f);
    float b = y + 1.8556f * (u - 0.5f);
    return make_float4(s_dec[unorm8(r)], s_dec[unorm8(g)], s_dec[unorm8(b)], 1.0f);
}

__device__ __forceinline__ float expand_chroma(float u) {
    // clamp((u - 16/255) / 0.87843137254, 0, 1) — planar_yuv_to_rgba.wgsl:49-50
    const float C = 0.87843137254f;
    return clampf(div_cr(u - (16.0f / 255.0f), C, 1.0f / C), 0.0f, 1.0f);
}

__global__ __launch_bounds__(A_THREADS) void k_ingest_resample(const IngestJob *__restrict__ jobs, const float *__restrict__ tables) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const IngestJob &J = jobs[blockIdx.z];
    if ((int)blockIdx.x >= J.tiles_x || (int)blockIdx.y >= J.tiles_y) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int tw = min(TW, J.dst.w - tx0), th = min(TH, J.dst.h - ty0);
    const int taps_h = J.taps_h, taps_v = J.taps_v;
    const int ncm = J.nc_max;

    // ---- LDS carve (all region sizes are multiples of 16 B)
    float *s_tab = (float *)smem;                       // SMR_TABLE_FLOATS: dec | thr | enc
    float *s_n255 = s_tab + SMR_TABLE_FLOATS;           // 256: u8 / 255
    float *s_ylut = s_n255 + 256;                       // 256: luma u8 -> range-expanded y
    float *s_wh = s_ylut + 256;                         // [taps_h][TW]
    float *s_wv = s_wh + MAX_TAPS * TW;                 // [TH][taps_v]
    int *s_fh = (int *)(s_wv + TH * MAX_TAPS);          // [TW]
    int *s_fv = s_fh + TW;                              // [TH]
    float *s_wsh = (float *)(s_fv + TH);                // [TW]   wsum
    float *s_rsh = s_wsh + TW;                          // [TW]   1 / wsum
    float *s_wsv = s_rsh + TW;                          // [TH]
    float *s_rsv = s_wsv + TH;                          // [TH]
    float4 *S = (float4 *)(s_rsv + TH);                 // [CH][nc_max] linear RGB (w = 1)
    uint2 *M = (uint2 *)(S + (size_t)CH * ncm);         // [nr_max][TW] half4

    for (int i = tid; i < SMR_TABLE_FLOATS; i += A_THREADS) s_tab[i] = tables[i];
    {
        // u8 -> f32 conversions done once per table entry with the same IEEE operations the per-pixel path uses
        const float v = (float)tid / 255.0f;
        s_n255[tid] = v;
        s_ylut[tid] = J.full_range ? v : clampf((v - (16.0f / 255.0f)) / 0.85882352941f, 0.0f, 1.0f);
    }
    for (int i = tid; i < taps_h * TW; i += A_THREADS) {
        int t = i / TW, x = i - t * TW;
        s_wh[i] = x < tw ? J.w_h[(size_t)(tx0 + x) * taps_h + t] : 0.0f;
    }
    for (int i = tid; i < th * taps_v; i += A_THREADS) s_wv[i] = J.w_v[(size_t)ty0 * taps_v + i];
    if (tid < tw) {
        s_fh[tid] = J.first_h[tx0 + tid];
        const float ws = J.wsum_h[tx0 + tid];
        s_wsh[tid] = ws;
        s_rsh[tid] = 1.0f / ws;
    }
    if (tid >= 64 && tid - 64 < th) {
        s_fv[tid - 64] = J.first_v[ty0 + tid - 64];
        const float ws = J.wsum_v[ty0 + tid - 64];
        s_wsv[tid - 64] = ws;
        s_rsv[tid - 64] = 1.0f / ws;
    }
    __syncthreads();
    const float *s_dec = s_tab, *s_thr = s_tab + 256;

    const int sw = J.src_w, sh = J.src_h;
    // first[] is non-decreasing in the output coordinate, so the tile's source footprint is:
    int c_lo = clampi(s_fh[0], 0, sw - 1);
    const int c_hi = clampi(s_fh[tw - 1] + taps_h - 1, 0, sw - 1);
    int r_lo = clampi(s_fv[0], 0, sh - 1);
    const int r_hi = clampi(s_fv[th - 1] + taps_v - 1, 0, sh - 1);
    if (J.fast420) {
        // 2x2 conversion quads start on odd luma coordinates (they share one 2x2 chroma neighbourhood)
        c_lo -= (c_lo & 1) ^ 1;
        r_lo -= (r_lo & 1) ^ 1;
    }
    const int NC = c_hi - c_lo + 1, NR = r_hi - r_lo + 1;

    for (int rc = 0; rc < NR; rc += CH) {
        // ---- convert CH source rows: YUV -> RGBA8 (node texture bytes) -> sRGB-decoded linear f32
        if (J.fast420) {
            const int qcols = (NC + 1) >> 1;
            for (int q = tid; q < (CH / 2) * qcols; q += A_THREADS) {
                const int qr = q / qcols, qc = q - qr * qcols;
                const int x0 = c_lo + 2 * qc, x1 = x0 + 1;         // x0 odd (or -1), x1 even
                const int y0 = r_lo + rc + 2 * qr, y1 = y0 + 1;     // y0 odd (or -1), y1 even
                if (y0 > r_hi) continue;
                const int qx = x1 >> 1, qy = y1 >> 1;
                const int cxa = clampi(qx - 1, 0, J.up.w - 1), cxb = clampi(qx, 0, J.up.w - 1);
                const int cya = clampi(qy - 1, 0, J.up.h - 1), cyb = clampi(qy, 0, J.up.h - 1);
                const u8 *ua = J.up.ptr + (size_t)cya * J.up.pitch, *ub = J.up.ptr + (size_t)cyb * J.up.pitch;
                const u8 *va = J.vp.ptr + (size_t)cya * J.vp.pitch, *vb = J.vp.ptr + (size_t)cyb * J.vp.pitch;
                const float u00 = s_n255[ua[cxa]], u01 = s_n255[ua[cxb]], u10 = s_n255[ub[cxa]], u11 = s_n255[ub[cxb]];
                const float v00 = s_n255[va[cxa]], v01 = s_n255[va[cxb]], v10 = s_n255[vb[cxa]], v11 = s_n255[vb[cxb]];
                // bilinear weights of the chroma tap: x0 (odd) -> fx = 1/4, x1 (even) -> fx = 3/4; same vertically
#pragma unroll
                for (int iy = 0; iy < 2; iy++) {
                    const int sy = iy ? y1 : y0;
                    if (sy < 0 || sy > r_hi) continue;
                    const float fy = iy ? 0.75f : 0.25f, gy = 1.0f - fy;
                    const u8 *yrow = J.yp.ptr + (size_t)sy * J.yp.pitch;
#pragma unroll
                    for (int ix = 0; ix < 2; ix++) {
                        const int sx = ix ? x1 : x0;
                        if (sx < 0 || sx > c_hi) continue;
                        const float fx = ix ? 0.75f : 0.25f, gx = 1.0f - fx;
                        const float uu = (u00 * gx + u01 * fx) * gy + (u10 * gx + u11 * fx) * fy;
                        const float vv = (v00 * gx + v01 * fx) * gy + (v10 * gx + v11 * fx) * fy;
                        const float yy = s_ylut[yrow[sx]];
                        const float ue = J.full_range ? uu : expand_chroma(uu), ve = J.full_range ? vv : expand_chroma(vv);
                        S[(2 * qr + iy) * ncm + (sx - c_lo)] = yuv_expanded_to_linear(yy, ue, ve, s_dec);
                    }
                }
            }
        } else {
            for (int row = wave; row < CH; row += 4) {
                const int sy = r_lo + rc + row;
                if (sy > r_hi) continue;
                const float tv = ((float)sy + 0.5f) / (float)sh;
                const u8 *yrow = J.yp.ptr + (size_t)sy * J.yp.pitch;
                for (int col = lane; col < NC; col += 64) {
                    const int sx = c_lo + col;
                    const float tu = ((float)sx + 0.5f) / (float)sw;
                    const float yy = (float)yrow[sx] / 255.0f;
                    const float uu = sample_plane_bilinear(J.up, 1, 0, tu, tv);
                    const float vv = sample_plane_bilinear(J.vp, 1, 0, tu, tv);
                    const u32 p = yuv_to_rgb_px(yy, uu, vv, J.full_range != 0);
                    S[row * ncm + col] = make_float4(s_dec[p & 0xff], s_dec[(p >> 8) & 0xff], s_dec[(p >> 16) & 0xff], 1.0f);
                }
            }
        }
        __syncthreads();
        // ---- horizontal Lanczos of those rows into the f16 intermediate (pass 1 of the separable plan).
        //      Sources are opaque (alpha == 1): (sum of w * 1) / wsum == 1 exactly, so alpha is not computed.
        if (lane < tw) {
            const int fh = s_fh[lane];
            const float wsh = s_wsh[lane], rsh = s_rsh[lane];
            const int row_a = wave * 2, row_b = row_a + 1;
            const bool va_ = r_lo + rc + row_a <= r_hi && r_lo + rc + row_a >= 0, vb_ = r_lo + rc + row_b <= r_hi && r_lo + rc + row_b >= 0;
            float ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
            const float4 *Sa = S + row_a * ncm - c_lo, *Sb = S + row_b * ncm - c_lo;
            for (int t = 0; t < taps_h; t++) {
                const float wgt = s_wh[t * TW + lane];
                const int s = clampi(fh + t, 0, sw - 1);
                const float4 ta = Sa[s], tb = Sb[s];
                ax = __builtin_fmaf(ta.x, wgt, ax); ay = __builtin_fmaf(ta.y, wgt, ay); az = __builtin_fmaf(ta.z, wgt, az);
                bx = __builtin_fmaf(tb.x, wgt, bx); by = __builtin_fmaf(tb.y, wgt, by); bz = __builtin_fmaf(tb.z, wgt, bz);
            }
            if (va_) {
                __half2 lo = __floats2half2_rn(div_cr(ax, wsh, rsh), div_cr(ay, wsh, rsh)), hi = __floats2half2_rn(div_cr(az, wsh, rsh), 1.0f);
                uint2 raw; raw.x = *(const u32 *)&lo; raw.y = *(const u32 *)&hi;
                M[(size_t)(rc + row_a) * TW + lane] = raw;
            }
            if (vb_) {
                __half2 lo = __floats2half2_rn(div_cr(bx, wsh, rsh), div_cr(by, wsh, rsh)), hi = __floats2half2_rn(div_cr(bz, wsh, rsh), 1.0f);
                uint2 raw; raw.x = *(const u32 *)&lo; raw.y = *(const u32 *)&hi;
                M[(size_t)(rc + row_b) * TW + lane] = raw;
            }
        }
        __syncthreads();
    }

    // ---- vertical Lanczos (pass 2) + sRGB encode + store
    if (lane < tw) {
        for (int y = wave; y < th; y += 4) {
            const int fv = s_fv[y];
            float sx_ = 0.f, sy_ = 0.f, sz_ = 0.f;
            const float *wv = s_wv + y * taps_v;
            for (int t = 0; t < taps_v; t++) {
                const float wgt = wv[t];
                const int r = clampi(fv + t, 0, sh - 1) - r_lo;
                const float4 m = half4_to_float4(M[(size_t)r * TW + lane]);
                sx_ = __builtin_fmaf(m.x, wgt, sx_); sy_ = __builtin_fmaf(m.y, wgt, sy_); sz_ = __builtin_fmaf(m.z, wgt, sz_);
            }
            const float ws = s_wsv[y], rs = s_rsv[y];
            const u32 r8 = srgb_encode8(div_cr(sx_, ws, rs), s_thr), g8 = srgb_encode8(div_cr(sy_, ws, rs), s_thr),
                      b8 = srgb_encode8(div_cr(sz_, ws, rs), s_thr);
            *(u32 *)(J.dst.ptr + (size_t)(ty0 + y) * J.dst.pitch + (size_t)(tx0 + lane) * 4) = r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
        }
    }
}

size_t ingest_lds_bytes(int nc_max, int nr_max) {
    size_t floats = SMR_TABLE_FLOATS + 256 + 256 + (size_t)MAX_TAPS * TW + (size_t)TH * MAX_TAPS + TW + TH + 2 * TW + 2 * TH;
    return floats * 4 + (size_t)CH * nc_max * 16 + (size_t)nr_max * TW * 8;
}

// ------------------------------------------------------------------ wave B: compose + output convert
constexpr int B_TILE_W = 128, B_TILE_H = 16;  // pixels; 32 x 8 threads, one 4x2 pixel block each

// Per-tile classification of the layout list (one thread per layout):
//   touch  — bounding box intersects the tile
//   solid  — for every pixel of the tile the fragment equals the layout's base value (the colour, or the
//            texture sample): tile inside the unrotated rect inset past radius / border / AA and inside every mask
//   start  — the last solid layout whose base value is opaque: everything before it is overwritten
__device__ __forceinline__ void classify_layouts(u32 *s_touch, u32 *s_solid, int *s_start, const DevLayout *__restrict__ layouts,
                                                 const smr_mask *__restrict__ masks, int n, int x0, int y0, int x1, int y1, int tid,
                                                 int nthreads) {
    if (tid < MAX_LAYOUT_WORDS) { s_touch[tid] = 0; s_solid[tid] = 0; }
    if (tid == 0) *s_start = -1;
    __syncthreads();
    const float cx0 = (float)x0 + 0.5f, cx1 = (float)x1 - 0.5f, cy0 = (float)y0 + 0.5f, cy1 = (float)y1 - 0.5f;
    for (int i = tid; i < n; i += nthreads) {
        const DevLayout &L = layouts[i];
        if (!(L.bx0 < x1 && L.bx1 > x0 && L.by0 < y1 && L.by1 > y0)) continue;
        atomicOr(&s_touch[i >> 5], 1u << (i & 31));
        if (!(L.flags & DL_UNROTATED)) continue;
        bool solid = L.left + L.inset <= cx0 && cx1 <= L.left + L.width - L.inset && L.top + L.inset <= cy0 &&
                     cy1 <= L.top + L.height - L.inset;
        for (u32 m = 0; solid && m < L.masks_len; m++) {
            const smr_mask &K = masks[L.masks_off + m];
            const float mi = fmaxf(fmaxf(K.radius[0], K.radius[1]), fmaxf(K.radius[2], K.radius[3])) + 1.0f;
            solid = K.left + mi <= cx0 && cx1 <= K.left + K.width - mi && K.top + mi <= cy0 && cy1 <= K.top + K.height - mi;
        }
        if (!solid) continue;
        atomicOr(&s_solid[i >> 5], 1u << (i & 31));
        const bool opaque = (L.type == 0) ? (L.src_kind == 2) : ((L.flags & DL_COLOR_OPAQUE) != 0);
        if (opaque) atomicMax(s_start, i);
    }
    __syncthreads();
}

// NV = 0: planar Y,U,V (4:2:0); NV = 1: NV12 (Y + interleaved UV)
template <int NV>
__global__ __launch_bounds__(256) void k_compose_output(SurfView yp, SurfView up, SurfView vp, int W, int H,
                                                        const DevLayout *__restrict__ layouts, const smr_mask *__restrict__ masks,
                                                        int n, int srgb, const float *__restrict__ tables) {
    __shared__ u32 s_touch[MAX_LAYOUT_WORDS], s_solid[MAX_LAYOUT_WORDS];
    __shared__ int s_start;
    __shared__ float s_tab[SMR_TABLE_FLOATS];
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * B_TILE_W, ty0 = blockIdx.y * B_TILE_H;
    for (int i = tid; i < SMR_TABLE_FLOATS; i += 256) s_tab[i] = tables[i];
    classify_layouts(s_touch, s_solid, &s_start, layouts, masks, n, tx0, ty0, min(tx0 + B_TILE_W, W), min(ty0 + B_TILE_H, H), tid, 256);
    const float *dec = s_tab, *thr = s_tab + 256;

    const int px0 = tx0 + 4 * (tid & 31), py0 = ty0 + 2 * (tid >> 5);
    if (px0 >= W || py0 >= H) return;  // W % 4 == 0, H % 2 == 0: a block is entirely inside or outside

    u32 acc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // [row][col]: acc[r * 4 + c]
    const int start = s_start;
    const int words = (n + 31) >> 5;
    for (int wi = start < 0 ? 0 : (start >> 5); wi < words; wi++) {
        u32 bits = s_touch[wi];
        if (start >= 0 && wi == (start >> 5)) bits &= ~((1u << (start & 31)) - 1u);
        const u32 solid_bits = s_solid[wi];
        while (bits) {
            const int b = __builtin_ctz(bits);
            const int li = (wi << 5) + b;
            bits &= bits - 1;
            const DevLayout &L = layouts[li];
            const bool solid = (solid_bits >> b) & 1u;
            if (li == start) {
                // opaque base layer: dst is irrelevant (dst * (1 - 1) == 0 exactly)
                if (L.type != 0) {
#pragma unroll
                    for (int k = 0; k < 8; k++) acc[k] = L.solid_px;
                } else if (L.flags & DL_ALIGNED) {
                    // 1:1 blit of an opaque texture: bilinear weights are exactly (1,0), decode -> encode is the identity
                    const u8 *r0 = L.src.ptr + (size_t)(py0 - L.iy) * L.src.pitch + (size_t)(px0 - L.ix) * 4;
                    const u8 *r1 = r0 + L.src.pitch;
                    u32 a0[4], a1[4];
                    if ((((uintptr_t)r0) & 15) == 0 && (L.src.pitch & 15) == 0) {
                        const uint4 t0 = *(const uint4 *)r0, t1 = *(const uint4 *)r1;
                        a0[0] = t0.x; a0[1] = t0.y; a0[2] = t0.z; a0[3] = t0.w;
                        a1[0] = t1.x; a1[1] = t1.y; a1[2] = t1.z; a1[3] = t1.w;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; c++) { a0[c] = ((const u32 *)r0)[c]; a1[c] = ((const u32 *)r1)[c]; }
                    }
#pragma unroll
                    for (int c = 0; c < 4; c++) { acc[c] = a0[c]; acc[4 + c] = a1[c]; }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) acc[k] = composite_layout(0u, L, masks, px0 + (k & 3), py0 + (k >> 2), srgb, dec, thr);
                }
                continue;
            }
            if (solid && L.type != 0) {
                // fragment == colour everywhere in this tile: skip coverage / SDF / masks
                const float4 frag = make_float4(L.color[0], L.color[1], L.color[2], L.color[3]);
#pragma unroll
                for (int k = 0; k < 8; k++) acc[k] = blend_store(acc[k], frag, srgb, dec, thr);
                continue;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] = composite_layout(acc[k], L, masks, px0 + (k & 3), py0 + (k >> 2), srgb, dec, thr);
        }
    }

    // RGBA -> Y'CbCr on the raw (gamma-encoded) bytes: rgba_to_yuv.wgsl:26-54 / rgba_to_nv12.wgsl:24-52
    float4 c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = unpack_unorm(acc[k]);
    u32 yrow0 = 0, yrow1 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        yrow0 |= unorm8(yuv_component(c[k], 0)) << (8 * k);
        yrow1 |= unorm8(yuv_component(c[4 + k], 0)) << (8 * k);
    }
    *(u32 *)(yp.ptr + (size_t)py0 * yp.pitch + px0) = yrow0;
    *(u32 *)(yp.ptr + (size_t)(py0 + 1) * yp.pitch + px0) = yrow1;
    // chroma: the bilinear tap at the chroma texel centre = weights (1/2, 1/2) x (1/2, 1/2)
    const float fx = 0.5f, gx = 1.0f - fx, fy = 0.5f, gy = 1.0f - fy;
    u32 uv[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const float4 &p00 = c[2 * j], &p01 = c[2 * j + 1], &p10 = c[4 + 2 * j], &p11 = c[4 + 2 * j + 1];
        float4 m;
        m.x = (p00.x * gx + p01.x * fx) * gy + (p10.x * gx + p11.x * fx) * fy;
        m.y = (p00.y * gx + p01.y * fx) * gy + (p10.y * gx + p11.y * fx) * fy;
        m.z = (p00.z * gx + p01.z * fx) * gy + (p10.z * gx + p11.z * fx) * fy;
        m.w = 0.0f;
        uv[j][0] = unorm8(yuv_component(m, 1));
        uv[j][1] = unorm8(yuv_component(m, 2));
    }
    const int cx = px0 >> 1, cy = py0 >> 1;
    if (NV == 0) {
        *(u16 *)(up.ptr + (size_t)cy * up.pitch + cx) = (u16)(uv[0][0] | (uv[1][0] << 8));
        *(u16 *)(vp.ptr + (size_t)cy * vp.pitch + cx) = (u16)(uv[0][1] | (uv[1][1] << 8));
    } else {
        *(u32 *)(up.ptr + (size_t)cy * up.pitch + (size_t)cx * 2) = uv[0][0] | (uv[0][1] << 8) | (uv[1][0] << 16) | (uv[1][1] << 24);
    }
}

bool fused_disabled(smr_ctx *ctx) {
    if (ctx->fused_disabled < 0) {
        const char *e = getenv("SMR_DISABLE_FUSED");
        ctx->fused_disabled = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return ctx->fused_disabled == 1;
}

bool is_planar_yuv(u32 fmt) { return fmt <= SMR_FRAME_PLANAR_YUVJ420; }
// every YUV-family FrameData variant converts to alpha == 1 (wgpu/format/*_to_rgba.wgsl return vec4(.., 1.0))
bool frame_is_opaque(u32 fmt) { return fmt <= SMR_FRAME_NV12; }

// What wave A covers: planar YUV frames, separable plan, no box pre-reduction, horizontal pass first.
bool can_fuse_ingest(const smr_frame *f, const smr_resample_plan &plan) {
    return f && is_planar_yuv(f->format) && f->planes[0] && f->planes[1] && f->planes[2] && plan.kind == 2 && plan.levels[0] == 0 &&
           plan.levels[1] == 0 && plan.axis[0] == 0;
}

int make_ingest_job(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile, IngestJob *out) {
    WeightPtrs wh, wv;
    int rc = get_weights(ctx, plan.scale[0], plan.offset[0], (int)tile->w, &wh);
    if (rc != SMR_OK) return rc;
    rc = get_weights(ctx, plan.scale[1], plan.offset[1], (int)tile->h, &wv);
    if (rc != SMR_OK) return rc;
    IngestJob &J = *out;
    J.yp = view_of(f->planes[0]); J.up = view_of(f->planes[1]); J.vp = view_of(f->planes[2]);
    J.dst = view_of(tile);
    J.src_w = (int)f->width; J.src_h = (int)f->height;
    J.full_range = f->format == SMR_FRAME_PLANAR_YUVJ420 ? 1 : 0;
    J.fast420 = ((f->format == SMR_FRAME_PLANAR_YUV420 || f->format == SMR_FRAME_PLANAR_YUVJ420) && f->width % 2 == 0 &&
                 f->height % 2 == 0 && f->width >= 2 && f->height >= 2) ? 1 : 0;
    J.taps_h = wh.taps; J.taps_v = wv.taps;
    J.first_h = wh.first; J.wsum_h = wh.wsum; J.w_h = wh.w;
    J.first_v = wv.first; J.wsum_v = wv.wsum; J.w_v = wv.w;
    J.tiles_x = ((int)tile->w + TW - 1) / TW; J.tiles_y = ((int)tile->h + TH - 1) / TH;
    // +1: the quad path aligns the footprint start down to an odd coordinate
    J.nc_max = (int)ceilf((float)TW * fmaxf(plan.scale[0], 0.0f)) + wh.taps + 3;
    J.nr_max = (int)ceilf((float)TH * fmaxf(plan.scale[1], 0.0f)) + wv.taps + 3;
    if (J.nc_max > J.src_w + 1) J.nc_max = J.src_w + 1;
    if (J.nr_max > J.src_h + 1) J.nr_max = J.src_h + 1;
    J.nr_max = (J.nr_max + CH - 1) / CH * CH;  // chunk loop writes whole chunks of rows
    return SMR_OK;
}

int launch_ingest(smr_ctx *ctx, const std::vector<IngestJob> &jobs, const void *jobs_dev) {
    int gx = 0, gy = 0;
    size_t lds = 0;
    for (auto &J : jobs) {
        gx = J.tiles_x > gx ? J.tiles_x : gx;
        gy = J.tiles_y > gy ? J.tiles_y : gy;
        size_t b = ingest_lds_bytes(J.nc_max, J.nr_max);
        lds = b > lds ? b : lds;
    }
    if (lds > 160 * 1024) return smr_fail(ctx, SMR_ERR_INTERNAL, "k_ingest_resample: %zu B of LDS needed", lds);
    static bool attr_set = false;
    if (!attr_set) {
        SMR_HIP(ctx, hipFuncSetAttribute((const void *)k_ingest_resample, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    StageScope scope(ctx, SMR_STAGE_FUSED_INGEST);
    hipLaunchKernelGGL(k_ingest_resample, dim3((unsigned)gx, (unsigned)gy, (unsigned)jobs.size()), dim3(A_THREADS), lds, ctx->stream,
                       (const IngestJob *)jobs_dev, ctx->d_tables);
    SMR_HIP(ctx, hipGetLastError());
    return SMR_OK;
}

// surface-cache slot ranges (ctx->surf_cache)
constexpr size_t SLOT_TARGET = 0;
constexpr size_t SLOT_INGEST_NODE = 1;
constexpr size_t SLOT_NODE0 = 16;     // + source index
constexpr size_t SLOT_TILE0 = 2048;   // + layout index

}  // namespace

extern "C" int smr_render_layouts(smr_ctx *ctx, const smr_layout *layouts, uint32_t n, const smr_source *sources,
                                  uint32_t n_sources, uint32_t out_w, uint32_t out_h, const smr_frame *out,
                                  smr_surface *out_rgba) {
    if (!ctx || (n && !layouts) || (n_sources && !sources)) return SMR_ERR_INVALID;
    if (!out && !out_rgba) return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: no output given");
    if (out_w == 0 || out_h == 0) return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: empty output");
    if (out && (out->width != out_w || out->height != out_h))
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: output frame is %ux%u, expected %ux%u", out->width, out->height, out_w, out_h);
    if (out_rgba && (out_rgba->fmt != SMR_PX_RGBA8 || out_rgba->w != out_w || out_rgba->h != out_h))
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: out_rgba must be RGBA8 %ux%u", out_w, out_h);
    if (n > ctx->max_layouts) n = ctx->max_layouts;
    if (n_sources > 1024) return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: too many sources");
    const bool fused = !fused_disabled(ctx);

    // ---- sources: node views; frames get a node surface only if some layout needs one
    std::vector<SurfView> views(n_sources + n + 1);
    std::vector<int> kinds(n_sources + n + 1, 0);  // 0 none, 1 RGBA8, 2 RGBA8 known opaque
    std::vector<int> src_w(n_sources + 1, 0), src_h(n_sources + 1, 0);
    std::vector<u8> node_ready(n_sources + 1, 0);
    for (u32 i = 0; i < n_sources; i++) {
        const smr_source &s = sources[i];
        if (s.kind == SMR_SOURCE_SURFACE && s.surface) {
            if (s.surface->fmt != SMR_PX_RGBA8) return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: source %u is not RGBA8", i);
            views[i] = view_of(s.surface);
            kinds[i] = 1;
            src_w[i] = (int)s.surface->w;
            src_h[i] = (int)s.surface->h;
            node_ready[i] = 1;
        } else if (s.kind == SMR_SOURCE_FRAME && s.frame && s.frame->planes[0]) {
            kinds[i] = frame_is_opaque(s.frame->format) ? 2 : 1;  // view filled lazily by ensure_node
            src_w[i] = (int)s.frame->width;
            src_h[i] = (int)s.frame->height;
        } else if (s.kind != SMR_SOURCE_NONE && s.kind > SMR_SOURCE_FRAME) {
            return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: bad source kind %u", s.kind);
        }
    }
    auto ensure_node = [&](u32 i) -> int {
        if (node_ready[i]) return SMR_OK;
        const smr_frame *f = sources[i].frame;
        smr_surface *node = smr_cached_surface(ctx, SLOT_NODE0 + i, f->width, f->height, SMR_PX_RGBA8);
        if (!node) return SMR_ERR_OOM;
        int rc = smr_frame_to_rgba(ctx, f, node);
        if (rc != SMR_OK) return rc;
        views[i] = view_of(node);
        node_ready[i] = 1;
        return SMR_OK;
    };

    // ---- resample_scaled_children (layout.rs:238-278): per texture layout decide direct / general / fused
    std::vector<smr_layout> eff(layouts, layouts + n);
    std::vector<IngestJob> jobs;
    u32 next_view = n_sources;
    for (u32 li = 0; li < n; li++) {
        smr_layout &L = eff[li];
        if (L.type != 0) continue;
        const u32 si = L.source_index;
        if (si >= n_sources || kinds[si] == 0) { L.source_index = SMR_NO_SOURCE; continue; }
        const bool is_frame = sources[si].kind == SMR_SOURCE_FRAME;
        bool resampled = false;
        if (ctx->srgb()) {  // CpuOptimized has no resampler (layout/layout_renderer.rs:22-27)
            const float rw = roundf(L.width), rh = roundf(L.height);
            const u32 dw = rw >= 1.0f ? (u32)rw : 1u, dh = rh >= 1.0f ? (u32)rh : 1u;
            smr_resample_plan plan;
            int kind = smr_resample_plan_make((u32)src_w[si], (u32)src_h[si], L.crop, dw, dh, &plan);
            if (kind < 0) return smr_fail(ctx, kind, "smr_render_layouts: degenerate resample plan for layout %u", li);
            if (kind > 0) {
                if (dw > 16384 || dh > 16384) return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: layout %u is too large", li);
                smr_surface *tile = smr_cached_surface(ctx, SLOT_TILE0 + li, dw, dh, SMR_PX_RGBA8);
                if (!tile) return SMR_ERR_OOM;
                if (fused && is_frame && can_fuse_ingest(sources[si].frame, plan)) {
                    IngestJob J;
                    int rc = make_ingest_job(ctx, sources[si].frame, plan, tile, &J);
                    if (rc != SMR_OK) return rc;
                    jobs.push_back(J);
                } else {
                    if (is_frame) {
                        int rc = ensure_node(si);
                        if (rc != SMR_OK) return rc;
                    }
                    smr_surface node;  // non-owning alias of the node view for the general resampler
                    node.ptr = views[si].ptr; node.pitch = views[si].pitch; node.w = (u32)views[si].w; node.h = (u32)views[si].h;
                    node.fmt = SMR_PX_RGBA8;
                    int rc = smr_resample(ctx, &node, L.crop, tile);
                    if (rc < 0) return rc;
                }
                // ResampledChild::output_crop (resampler.rs:292-299)
                L.crop[0] = 0.0f; L.crop[1] = 0.0f; L.crop[2] = (float)dw; L.crop[3] = (float)dh;
                views[next_view] = view_of(tile);
                // a resampled opaque source stays opaque: alpha = (sum of w) / (sum of w) == 1 exactly in both passes
                kinds[next_view] = kinds[si];
                L.source_index = next_view++;
                resampled = true;
            }
        }
        if (!resampled && is_frame) {
            int rc = ensure_node(si);
            if (rc != SMR_OK) return rc;
        }
    }

    // ---- parameters -> device (one pinned staging slot, one copy)
    PackedLayouts packed;
    int rc = smr_pack_layouts(ctx, eff.data(), n, views.data(), kinds.data(), next_view, (int)out_w, (int)out_h,
                              jobs.size() * sizeof(IngestJob), &packed);
    if (rc != SMR_OK) return rc;
    if (!jobs.empty()) memcpy(packed.extra_host, jobs.data(), jobs.size() * sizeof(IngestJob));
    rc = smr_pack_commit(ctx, &packed);
    if (rc != SMR_OK) return rc;

    // ---- wave A
    if (!jobs.empty()) {
        rc = launch_ingest(ctx, jobs, packed.extra_dev);
        if (rc != SMR_OK) return rc;
    }

    // ---- wave B (or the general compositor + output converters)
    const bool fuse_out = fused && out && !out_rgba && (out->format == SMR_FRAME_PLANAR_YUV420 || out->format == SMR_FRAME_NV12) &&
                          (out_w % 4 == 0) && (out_h % 2 == 0) && out->planes[0] && out->planes[1] &&
                          (out->format == SMR_FRAME_NV12 || out->planes[2]);
    if (fuse_out) {
        StageScope scope(ctx, SMR_STAGE_FUSED_COMPOSE);
        dim3 grid((out_w + B_TILE_W - 1) / B_TILE_W, (out_h + B_TILE_H - 1) / B_TILE_H, 1);
        SurfView yp = view_of(out->planes[0]), up = view_of(out->planes[1]);
        if (out->format == SMR_FRAME_NV12) {
            hipLaunchKernelGGL(k_compose_output<1>, grid, dim3(256), 0, ctx->stream, yp, up, up, (int)out_w, (int)out_h, packed.layouts,
                               packed.masks, packed.n, ctx->srgb() ? 1 : 0, ctx->d_tables);
        } else {
            hipLaunchKernelGGL(k_compose_output<0>, grid, dim3(256), 0, ctx->stream, yp, up, view_of(out->planes[2]), (int)out_w, (int)out_h,
                               packed.layouts, packed.masks, packed.n, ctx->srgb() ? 1 : 0, ctx->d_tables);
        }
        SMR_HIP(ctx, hipGetLastError());
        return smr_pack_done(ctx, &packed);
    }

    smr_surface *target = out_rgba ? out_rgba : smr_cached_surface(ctx, SLOT_TARGET, out_w, out_h, SMR_PX_RGBA8);
    if (!target) return SMR_ERR_OOM;
    rc = smr_launch_apply_layouts(ctx, target, &packed);
    if (rc != SMR_OK) return rc;
    rc = smr_pack_done(ctx, &packed);
    if (rc != SMR_OK) return rc;
    if (out) return smr_rgba_to_frame(ctx, target, out);
    return SMR_OK;
}

// InputTexture::convert_to_node_texture + ResampledChild::render for one input, fused when the
// plan allows it (wave A with a single job), otherwise convert + general resample.  This is the
// per-shard step of the multi-GPU path: each GPU turns its inputs into dst-sized tiles.
extern "C" int smr_ingest_resample(smr_ctx *ctx, const smr_frame *in, const float crop[4], smr_surface *dst) {
    if (!ctx || !in || !crop || !dst) return SMR_ERR_INVALID;
    if (dst->fmt != SMR_PX_RGBA8) return smr_fail(ctx, SMR_ERR_INVALID, "smr_ingest_resample: dst must be RGBA8");
    if (!ctx->srgb()) return smr_fail(ctx, SMR_ERR_INVALID, "smr_ingest_resample: CpuOptimized mode has no resampler");
    smr_resample_plan plan;
    int kind = smr_resample_plan_make(in->width, in->height, crop, dst->w, dst->h, &plan);
    if (kind < 0) return smr_fail(ctx, kind, "smr_ingest_resample: degenerate plan");
    if (kind == 0) return 0;
    if (!fused_disabled(ctx) && can_fuse_ingest(in, plan)) {
        std::vector<IngestJob> jobs(1);
        int rc = make_ingest_job(ctx, in, plan, dst, &jobs[0]);
        if (rc != SMR_OK) return rc;
        PackedLayouts packed;
        rc = smr_pack_layouts(ctx, nullptr, 0, nullptr, nullptr, 0, 1, 1, sizeof(IngestJob), &packed);
        if (rc != SMR_OK) return rc;
        memcpy(packed.extra_host, jobs.data(), sizeof(IngestJob));
        rc = smr_pack_commit(ctx, &packed);
        if (rc != SMR_OK) return rc;
        rc = launch_ingest(ctx, jobs, packed.extra_dev);
        if (rc != SMR_OK) return rc;
        rc = smr_pack_done(ctx, &packed);
        return rc == SMR_OK ? kind : rc;
    }
    smr_surface *node = smr_cached_surface(ctx, SLOT_INGEST_NODE, in->width, in->height, SMR_PX_RGBA8);
    if (!node) return SMR_ERR_OOM;
    int rc = smr_frame_to_rgba(ctx, in, node);
    if (rc != SMR_OK) return rc;
    return smr_resample(ctx, node, crop, dst);
}
