// smr_fused_ingest.h — wave A of the hot path: k_ingest_resample (included by smr_fused.hip only).
//
// One launch covers every scaled planar-YUV input of the frame (blockIdx.z = job).  A 512-thread
// workgroup produces one 64x32 tile of the dst-sized RGBA8 tile surface:
//   * prologue: the tile's raw Y/U/V source footprint (~10 KB), the weight tables of its 64 columns /
//     32 rows and the sRGB tables are pulled into LDS with one round of loads — the only global-memory
//     latency the workgroup is exposed to;
//   * each of the 8 waves then runs its own pipeline over the footprint, two source rows at a time:
//     convert (YUV -> RGBA8 bytes -> sRGB-decoded linear f32; 2x2 quads sharing one chroma neighbourhood
//     for 4:2:0) into a wave-private LDS strip, then the horizontal Lanczos of exactly those two rows
//     into the shared f16 intermediate M (LDS).  No workgroup barrier separates the two — only the
//     wave's own in-order LDS stream;
//   * one barrier, then the vertical Lanczos over M, sRGB encode, one coalesced 256 B row store per wave.
// The node texture (RGBA8, input-sized) and the Rgba16Float intermediate of the reference never exist
// in HBM; their quantisation (u8, f16) is applied in registers at the same points.
#pragma once

#include "smr_convert_dev.h"
#include "smr_resample_dev.h"

#include <cmath>
#include <vector>

namespace {

// clamp for values that are never NaN (one v_med3_f32)
__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ u32 unorm8_fast(float x) { return (u32)(int)(clamp01(x) * 255.0f + 0.5f); }

// ------------------------------------------------------------------ weight tables (device cache)
__global__ __launch_bounds__(64) void k_build_weights(float scale, float offset, int taps, int n, int *__restrict__ first,
                                                      float *__restrict__ wsum, float *__restrict__ w) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float tmp[MAX_TAPS];
    float s;
    first[i] = lanczos_weights(i, scale, offset, taps, tmp, &s);
    wsum[i] = s;
    for (int t = 0; t < taps; t++) w[(size_t)t * n + i] = tmp[t];  // tap-major: a tile reads contiguous runs
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

// ------------------------------------------------------------------ kernel
constexpr int TW = 64;          // output tile width  (one lane per column)
constexpr int TH = 24;          // output tile height (keeps LDS at ~74 KB for k = 1.5: two workgroups per CU)
constexpr int A_WAVES = 8;
constexpr int A_THREADS = A_WAVES * 64;

struct IngestJob {
    SurfView yp, up, vp;  // planar source planes (chroma views carry the logical chroma size)
    SurfView dst;         // RGBA8 tile, dst-sized
    int src_w, src_h;
    int full_range;
    int fast420;          // 4:2:0 with even luma size: LDS-staged 2x2-quad conversion path
    int ablate;           // profiling only (SMR_ABLATE): 1 skip convert, 2 skip H taps, 4 skip V taps, 8 skip staging
    int taps_h, taps_v;
    float scale_h, off_h, scale_v, off_v;  // axis mappings (resampler.rs:36-48): source texels per output texel, crop offset
    const float *wsum_h; const float *w_h;  // device weight tables: wsum[n], w[taps][n] (tap-major)
    const float *wsum_v; const float *w_v;
    int tiles_x, tiles_y;
    int nc_max, nr_max;   // LDS capacity: columns of a source strip, rows of M (even)
};

// raw-footprint staging geometry (bytes)
__host__ __device__ inline int raw_y_stride(int nc_max) { return (nc_max + 8 + 3) & ~3; }
__host__ __device__ inline int raw_c_stride(int nc_max) { return ((nc_max >> 1) + 4 + 3) & ~3; }
__host__ __device__ inline int raw_c_rows(int nr_max) { return (nr_max >> 1) + 2; }

__device__ __forceinline__ float4 half4_to_float4(uint2 raw) {
    __half2 lo = *(const __half2 *)&raw.x, hi = *(const __half2 *)&raw.y;
    float2 a = __half22float2(lo), b = __half22float2(hi);
    return make_float4(a.x, a.y, b.x, b.y);
}

__device__ __forceinline__ uint2 float4_to_half4(float x, float y, float z, float w) {
    __half2 lo = __floats2half2_rn(x, y), hi = __floats2half2_rn(z, w);
    uint2 raw;
    raw.x = *(const u32 *)&lo;
    raw.y = *(const u32 *)&hi;
    return raw;
}

// rest of planar_yuv_to_rgba.wgsl:53-57 once y,u,v are range-expanded: matrix, clamp, unorm8 store, then the
// node texture's sRGB view decode (LUT) -> linear RGB
__device__ __forceinline__ float4 yuv_expanded_to_linear(float y, float u, float v, const float *__restrict__ s_dec) {
    float r = y + 1.5748f * (v - 0.5f);
    float g = y - 0.1873f * (u - 0.5f) - 0.4681f * (v - 0.5f);
    float b = y + 1.8556f * (u - 0.5f);
    return make_float4(s_dec[unorm8_fast(r)], s_dec[unorm8_fast(g)], s_dec[unorm8_fast(b)], 1.0f);
}

__device__ __forceinline__ float expand_chroma(float u) {
    // clamp((u - 16/255) / 0.87843137254, 0, 1) — planar_yuv_to_rgba.wgsl:49-50
    const float C = 0.87843137254f;
    return clamp01(div_cr(u - (16.0f / 255.0f), C, 1.0f / C));
}

// Job descriptors travel in the kernel-argument segment (scalar loads, no dependent global round trip).
constexpr int MAX_JOBS_PER_LAUNCH = 16;
struct IngestArgs {
    IngestJob jobs[MAX_JOBS_PER_LAUNCH];
};

__global__ __launch_bounds__(A_THREADS) void k_ingest_resample(const IngestArgs args, const float *__restrict__ tables) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const IngestJob &J = args.jobs[blockIdx.z];
    if ((int)blockIdx.x >= J.tiles_x || (int)blockIdx.y >= J.tiles_y) return;
    if (J.ablate & 16) return;  // profiling: pure dispatch cost of this grid

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int tw = min(TW, J.dst.w - tx0), th = min(TH, J.dst.h - ty0);
    const int taps_h = J.taps_h, taps_v = J.taps_v;
    const int ncm = J.nc_max;
    const int sw = J.src_w, sh = J.src_h;

    // ---- LDS carve (every region is a multiple of 16 B)
    float *s_tab = (float *)smem;                          // SMR_TABLE_FLOATS: dec | thr | enc
    float *s_n255 = s_tab + SMR_TABLE_FLOATS;              // 256: u8 / 255
    float *s_ylut = s_n255 + 256;                          // 256: luma u8 -> range-expanded y
    float *s_wh = s_ylut + 256;                            // [taps_h][TW]
    float *s_wv = s_wh + ((taps_h * TW + 3) & ~3);         // [TH][taps_v]
    int *s_fh = (int *)(s_wv + ((TH * taps_v + 3) & ~3));  // [TW]
    int *s_fv = s_fh + TW;                                 // [TH]
    float *s_wsh = (float *)(s_fv + TH);                   // [TW]   wsum
    float *s_rsh = s_wsh + TW;                             // [TW]   1 / wsum
    float *s_wsv = s_rsh + TW;                             // [TH]
    float *s_rsv = s_wsv + TH;                             // [TH]
    float4 *S_all = (float4 *)(s_rsv + TH);                // [A_WAVES][2][nc_max] linear RGBA, wave-private strips
    uint2 *M = (uint2 *)(S_all + (size_t)A_WAVES * 2 * ncm);  // [nr_max][TW] half4
    u8 *rawY = (u8 *)(M + (size_t)J.nr_max * TW);          // [nr_max][ys]        (fast420 only)
    const int ys = raw_y_stride(ncm), cs = raw_c_stride(ncm), crows = raw_c_rows(J.nr_max);
    u8 *rawU = rawY + (size_t)J.nr_max * ys;               // [crows][cs]
    u8 *rawV = rawU + (size_t)crows * cs;

    // ---- the tile's source footprint (first[] is non-decreasing in the output coordinate)
    //      computed from the mapping itself (same f32 sequence as the weight tables) — no dependent load
    int c_lo = clampi(lanczos_first(tx0, J.scale_h, J.off_h), 0, sw - 1);
    const int c_hi = clampi(lanczos_first(tx0 + tw - 1, J.scale_h, J.off_h) + taps_h - 1, 0, sw - 1);
    int r_lo = clampi(lanczos_first(ty0, J.scale_v, J.off_v), 0, sh - 1);
    const int r_hi = clampi(lanczos_first(ty0 + th - 1, J.scale_v, J.off_v) + taps_v - 1, 0, sh - 1);
    if (J.fast420) {
        // 2x2 conversion quads start on odd luma coordinates (they share one 2x2 chroma neighbourhood)
        c_lo -= (c_lo & 1) ^ 1;
        r_lo -= (r_lo & 1) ^ 1;
    }
    const int NC = c_hi - c_lo + 1, NR = r_hi - r_lo + 1;
    const int n_pairs = (NR + 1) >> 1;
    const int cbase = max(c_lo, 0) & ~3;          // luma staging keeps global dword alignment
    const int qx0 = (c_lo + 1) >> 1, qy0 = (r_lo + 1) >> 1;  // chroma index of the first quad column / row

    // ---- prologue: one round of global loads
    if (!(J.ablate & 32))
    for (int i = tid; i < SMR_TABLE_FLOATS; i += A_THREADS) s_tab[i] = tables[i];
    if (tid < 256 && !(J.ablate & 64)) {
        // u8 -> f32 conversions done once per table entry with the same IEEE operations the per-pixel path uses
        const float v = (float)tid / 255.0f;
        s_n255[tid] = v;
        s_ylut[tid] = J.full_range ? v : clampf((v - (16.0f / 255.0f)) / 0.85882352941f, 0.0f, 1.0f);
    }
    // weight tables are stored tap-major in global memory (w[t * n + i]): a tile's slice of every tap is contiguous
    for (int i = tid; i < taps_h * TW; i += A_THREADS) {
        int t = i / TW, x = i - t * TW;
        s_wh[i] = x < tw ? J.w_h[(size_t)t * J.dst.w + tx0 + x] : 0.0f;
    }
    for (int i = tid; i < taps_v * TH; i += A_THREADS) {
        int t = i / TH, y = i - t * TH;
        if (y < th) s_wv[y * taps_v + t] = J.w_v[(size_t)t * J.dst.h + ty0 + y];
    }
    if (tid < tw) {
        s_fh[tid] = lanczos_first(tx0 + tid, J.scale_h, J.off_h);
        const float ws = J.wsum_h[tx0 + tid];
        s_wsh[tid] = ws;
        s_rsh[tid] = 1.0f / ws;
    }
    if (tid >= 64 && tid - 64 < th) {
        s_fv[tid - 64] = lanczos_first(ty0 + tid - 64, J.scale_v, J.off_v);
        const float ws = J.wsum_v[ty0 + tid - 64];
        s_wsv[tid - 64] = ws;
        s_rsv[tid - 64] = 1.0f / ws;
    }
    if (J.fast420 && !(J.ablate & 8)) {
        // luma: aligned dwords of rows [max(r_lo,0), r_hi], columns [cbase, c_hi].  Half a wave per row (32 dwords =
        // 128 B), 16 rows per sweep; every sweep's loads are issued before the first LDS store waits on them.
        const int ndw = ((c_hi - cbase) >> 2) + 1;
        const int row0 = max(r_lo, 0);
        const int nrows = r_hi - row0 + 1;
        for (int d0 = 0; d0 < ndw; d0 += 32) {
            const int d = d0 + (lane & 31);
            constexpr int SWEEPS = 4;  // 64 rows per outer iteration
            for (int rbase = 0; rbase < nrows; rbase += 16 * SWEEPS) {
                u32 v[SWEEPS];
#pragma unroll
                for (int s = 0; s < SWEEPS; s++) {
                    const int rr = rbase + s * 16 + wave * 2 + (lane >> 5);
                    // rows are pitched to >= 4 B multiples, so reading the last dword of a row never leaves the allocation
                    v[s] = (rr < nrows && d < ndw) ? *(const u32 *)(J.yp.ptr + (size_t)(row0 + rr) * J.yp.pitch + cbase + 4 * d) : 0u;
                }
#pragma unroll
                for (int s = 0; s < SWEEPS; s++) {
                    const int rr = rbase + s * 16 + wave * 2 + (lane >> 5);
                    if (rr < nrows && d < ndw) *(u32 *)(rawY + (size_t)(row0 + rr - r_lo) * ys + 4 * d) = v[s];
                }
            }
        }
        // chroma: rows qy0-1 .. qy0+n_pairs-1 (+1), columns qx0-1 .. qx0+nq, edge clamp baked in
        const int nq = (NC + 1) >> 1;
        const int ccols = nq + 1, crw = n_pairs + 1;
        for (int k = lane; k < ccols; k += 64) {
            const int cx = clampi(qx0 - 1 + k, 0, J.up.w - 1);
            constexpr int CS = 4;  // 32 chroma rows per outer iteration, all loads issued before the stores
            for (int jbase = 0; jbase < crw; jbase += A_WAVES * CS) {
                u8 bu[CS], bv[CS];
#pragma unroll
                for (int s = 0; s < CS; s++) {
                    const int j = jbase + s * A_WAVES + wave;
                    const int cy = clampi(qy0 - 1 + j, 0, J.up.h - 1);
                    bu[s] = j < crw ? J.up.ptr[(size_t)cy * J.up.pitch + cx] : (u8)0;
                    bv[s] = j < crw ? J.vp.ptr[(size_t)cy * J.vp.pitch + cx] : (u8)0;
                }
#pragma unroll
                for (int s = 0; s < CS; s++) {
                    const int j = jbase + s * A_WAVES + wave;
                    if (j < crw) { rawU[j * cs + k] = bu[s]; rawV[j * cs + k] = bv[s]; }
                }
            }
        }
    }
    __syncthreads();
    const float *s_dec = s_tab, *s_thr = s_tab + 256;
    float4 *S = S_all + (size_t)wave * 2 * ncm;  // this wave's two-row strip

    for (int pr = wave; pr < n_pairs; pr += A_WAVES) {
        const int y0 = r_lo + 2 * pr, y1 = y0 + 1;
        // ---- convert the two source rows: YUV -> RGBA8 (node texture bytes) -> sRGB-decoded linear f32
        if (J.ablate & 1) {
        } else if (J.fast420) {
            const u8 *ua = rawU + pr * cs, *ub = ua + cs, *va = rawV + pr * cs, *vb = va + cs;
            const u8 *yr0 = rawY + (size_t)(2 * pr) * ys - cbase, *yr1 = yr0 + ys;  // index by absolute x
            const bool ok0 = y0 >= 0, ok1 = y1 <= r_hi;
            for (int qc = lane; qc < ((NC + 1) >> 1); qc += 64) {
                const int x0 = c_lo + 2 * qc, x1 = x0 + 1;  // x0 odd (or -1), x1 even
                const float u00 = s_n255[ua[qc]], u01 = s_n255[ua[qc + 1]], u10 = s_n255[ub[qc]], u11 = s_n255[ub[qc + 1]];
                const float v00 = s_n255[va[qc]], v01 = s_n255[va[qc + 1]], v10 = s_n255[vb[qc]], v11 = s_n255[vb[qc + 1]];
                // bilinear weights of the chroma tap: odd coordinate -> 1/4, even -> 3/4 (planar_yuv_to_rgba.wgsl:37-39)
#pragma unroll
                for (int ix = 0; ix < 2; ix++) {
                    const int sx = ix ? x1 : x0;
                    if (sx < 0 || sx > c_hi) continue;
                    const float fx = ix ? 0.75f : 0.25f, gx = 1.0f - fx;
                    const float ut = u00 * gx + u01 * fx, ubt = u10 * gx + u11 * fx;
                    const float vt = v00 * gx + v01 * fx, vbt = v10 * gx + v11 * fx;
                    if (ok0) {
                        const float uu = ut * 0.75f + ubt * 0.25f, vv = vt * 0.75f + vbt * 0.25f;
                        const float ue = J.full_range ? uu : expand_chroma(uu), ve = J.full_range ? vv : expand_chroma(vv);
                        S[sx - c_lo] = yuv_expanded_to_linear(s_ylut[yr0[sx]], ue, ve, s_dec);
                    }
                    if (ok1) {
                        const float uu = ut * 0.25f + ubt * 0.75f, vv = vt * 0.25f + vbt * 0.75f;
                        const float ue = J.full_range ? uu : expand_chroma(uu), ve = J.full_range ? vv : expand_chroma(vv);
                        S[ncm + sx - c_lo] = yuv_expanded_to_linear(s_ylut[yr1[sx]], ue, ve, s_dec);
                    }
                }
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int sy = rr ? y1 : y0;
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
                    S[rr * ncm + col] = make_float4(s_dec[p & 0xff], s_dec[(p >> 8) & 0xff], s_dec[(p >> 16) & 0xff], 1.0f);
                }
            }
        }
        // the strip is private to this wave and a wave's LDS operations complete in order: a fence is all that is needed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- horizontal Lanczos of the two rows into the f16 intermediate (pass 1 of the separable plan).
        //      All four channels ride in packed FMAs; alpha comes out as (sum w)/(sum w) == 1 exactly.
        if (lane < tw) {
            const int fh = s_fh[lane];
            const float wsh = s_wsh[lane], rsh = s_rsh[lane];
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *wcol = s_wh + lane;
            if (J.ablate & 2) {
            } else if (fh >= 0 && fh + taps_h - 1 <= sw - 1) {
                // interior: no edge clamp, consecutive texels
                const float4 *pa = S + (fh - c_lo), *pb = pa + ncm;
                for (int t = 0; t < taps_h; t++) {
                    const float wgt = wcol[t * TW];
                    const float4 ta = pa[t], tb = pb[t];
                    a.x = __builtin_fmaf(ta.x, wgt, a.x); a.y = __builtin_fmaf(ta.y, wgt, a.y);
                    a.z = __builtin_fmaf(ta.z, wgt, a.z); a.w = __builtin_fmaf(ta.w, wgt, a.w);
                    b.x = __builtin_fmaf(tb.x, wgt, b.x); b.y = __builtin_fmaf(tb.y, wgt, b.y);
                    b.z = __builtin_fmaf(tb.z, wgt, b.z); b.w = __builtin_fmaf(tb.w, wgt, b.w);
                }
            } else {
                const float4 *Sa = S - c_lo, *Sb = S + ncm - c_lo;
                for (int t = 0; t < taps_h; t++) {
                    const float wgt = wcol[t * TW];
                    const int s = clampi(fh + t, 0, sw - 1);
                    const float4 ta = Sa[s], tb = Sb[s];
                    a.x = __builtin_fmaf(ta.x, wgt, a.x); a.y = __builtin_fmaf(ta.y, wgt, a.y);
                    a.z = __builtin_fmaf(ta.z, wgt, a.z); a.w = __builtin_fmaf(ta.w, wgt, a.w);
                    b.x = __builtin_fmaf(tb.x, wgt, b.x); b.y = __builtin_fmaf(tb.y, wgt, b.y);
                    b.z = __builtin_fmaf(tb.z, wgt, b.z); b.w = __builtin_fmaf(tb.w, wgt, b.w);
                }
            }
            if (y0 >= 0) M[(size_t)(2 * pr) * TW + lane] = float4_to_half4(div_cr(a.x, wsh, rsh), div_cr(a.y, wsh, rsh), div_cr(a.z, wsh, rsh), div_cr(a.w, wsh, rsh));
            if (y1 <= r_hi) M[(size_t)(2 * pr + 1) * TW + lane] = float4_to_half4(div_cr(b.x, wsh, rsh), div_cr(b.y, wsh, rsh), div_cr(b.z, wsh, rsh), div_cr(b.w, wsh, rsh));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();

    // ---- vertical Lanczos (pass 2) + sRGB encode + store
    if (lane < tw && !(J.ablate & 128)) {
        for (int y = wave; y < th; y += A_WAVES) {
            const int fv = s_fv[y];
            float sx_ = 0.f, sy_ = 0.f, sz_ = 0.f;
            const float *wv = s_wv + y * taps_v;
            if (J.ablate & 4) {
            } else if (fv >= 0 && fv + taps_v - 1 <= sh - 1) {
                const uint2 *pm = M + (size_t)(fv - r_lo) * TW + lane;
                for (int t = 0; t < taps_v; t++) {
                    const float wgt = wv[t];
                    const float4 m = half4_to_float4(pm[(size_t)t * TW]);
                    sx_ = __builtin_fmaf(m.x, wgt, sx_); sy_ = __builtin_fmaf(m.y, wgt, sy_); sz_ = __builtin_fmaf(m.z, wgt, sz_);
                }
            } else {
                for (int t = 0; t < taps_v; t++) {
                    const float wgt = wv[t];
                    const int r = clampi(fv + t, 0, sh - 1) - r_lo;
                    const float4 m = half4_to_float4(M[(size_t)r * TW + lane]);
                    sx_ = __builtin_fmaf(m.x, wgt, sx_); sy_ = __builtin_fmaf(m.y, wgt, sy_); sz_ = __builtin_fmaf(m.z, wgt, sz_);
                }
            }
            const float ws = s_wsv[y], rs = s_rsv[y];
            const u32 r8 = srgb_encode8(div_cr(sx_, ws, rs), s_thr), g8 = srgb_encode8(div_cr(sy_, ws, rs), s_thr),
                      b8 = srgb_encode8(div_cr(sz_, ws, rs), s_thr);
            *(u32 *)(J.dst.ptr + (size_t)(ty0 + y) * J.dst.pitch + (size_t)(tx0 + lane) * 4) = r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
        }
    }
}

size_t ingest_lds_bytes(const IngestJob &J) {
    size_t floats = SMR_TABLE_FLOATS + 256 + 256 + (size_t)((J.taps_h * TW + 3) & ~3) + (size_t)((TH * J.taps_v + 3) & ~3) + TW + TH +
                    2 * TW + 2 * TH;
    size_t bytes = floats * 4 + (size_t)A_WAVES * 2 * J.nc_max * 16 + (size_t)J.nr_max * TW * 8;
    if (J.fast420) bytes += (size_t)J.nr_max * raw_y_stride(J.nc_max) + 2 * (size_t)raw_c_rows(J.nr_max) * raw_c_stride(J.nc_max);
    return (bytes + 15) & ~(size_t)15;
}

bool is_planar_yuv(u32 fmt) { return fmt <= SMR_FRAME_PLANAR_YUVJ420; }

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
    // the staged path reads whole dwords of luma rows: needs the 256 B row pitch of smr_frame_create / a 4 B-aligned pitch
    J.fast420 = ((f->format == SMR_FRAME_PLANAR_YUV420 || f->format == SMR_FRAME_PLANAR_YUVJ420) && f->width % 2 == 0 &&
                 f->height % 2 == 0 && f->width >= 2 && f->height >= 2 && (J.yp.pitch % 4) == 0 && (((uintptr_t)J.yp.ptr) % 4) == 0 &&
                 J.yp.pitch >= ((f->width + 3u) & ~3u)) ? 1 : 0;
    J.ablate = ctx->ablate;
    J.taps_h = wh.taps; J.taps_v = wv.taps;
    J.scale_h = plan.scale[0]; J.off_h = plan.offset[0];
    J.scale_v = plan.scale[1]; J.off_v = plan.offset[1];
    J.wsum_h = wh.wsum; J.w_h = wh.w;
    J.wsum_v = wv.wsum; J.w_v = wv.w;
    J.tiles_x = ((int)tile->w + TW - 1) / TW; J.tiles_y = ((int)tile->h + TH - 1) / TH;
    // +1: the quad path aligns the footprint start down to an odd coordinate
    J.nc_max = (int)ceilf((float)TW * fmaxf(plan.scale[0], 0.0f)) + wh.taps + 3;
    J.nr_max = (int)ceilf((float)TH * fmaxf(plan.scale[1], 0.0f)) + wv.taps + 3;
    if (J.nc_max > J.src_w + 1) J.nc_max = J.src_w + 1;
    if (J.nr_max > J.src_h + 1) J.nr_max = J.src_h + 1;
    J.nr_max = (J.nr_max + 1) & ~1;  // rows are produced in pairs
    return SMR_OK;
}

int launch_ingest(smr_ctx *ctx, const std::vector<IngestJob> &jobs) {
    static bool attr_set = false;
    if (!attr_set) {
        SMR_HIP(ctx, hipFuncSetAttribute((const void *)k_ingest_resample, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    StageScope scope(ctx, SMR_STAGE_FUSED_INGEST);
    for (size_t j0 = 0; j0 < jobs.size(); j0 += MAX_JOBS_PER_LAUNCH) {
        const size_t nj = jobs.size() - j0 < (size_t)MAX_JOBS_PER_LAUNCH ? jobs.size() - j0 : (size_t)MAX_JOBS_PER_LAUNCH;
        IngestArgs args;
        memset(&args, 0, sizeof(args));
        int gx = 0, gy = 0;
        size_t lds = 0;
        for (size_t j = 0; j < nj; j++) {
            const IngestJob &J = jobs[j0 + j];
            args.jobs[j] = J;
            gx = J.tiles_x > gx ? J.tiles_x : gx;
            gy = J.tiles_y > gy ? J.tiles_y : gy;
            size_t b = ingest_lds_bytes(J);
            lds = b > lds ? b : lds;
        }
        if (lds > 160 * 1024) return smr_fail(ctx, SMR_ERR_INTERNAL, "k_ingest_resample: %zu B of LDS needed", lds);
        hipLaunchKernelGGL(k_ingest_resample, dim3((unsigned)gx, (unsigned)gy, (unsigned)nj), dim3(A_THREADS), lds, ctx->stream, args,
                           ctx->d_tables);
        SMR_HIP(ctx, hipGetLastError());
    }
    return SMR_OK;
}

}  // namespace
