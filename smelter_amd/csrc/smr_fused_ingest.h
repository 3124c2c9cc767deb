// smr_fused_ingest.h — wave A of the hot path: k_ingest_resample (included by smr_fused.hip only).
//
// One launch covers every scaled planar-YUV / NV12 input of the frame: the launch's output rows (all jobs, strip-major) are
// split evenly over the blocks.  A 512-thread workgroup owns a 64-column strip of a dst-sized RGBA8 tile surface over a
// range of its rows and streams the matching source rows through LDS in chunks of 16 (one row pair per wave):
//   * stage   — the chunk's raw Y/U/V footprint (aligned dwords) into LDS;
//   * convert — per wave: YUV -> RGBA8 bytes -> sRGB-decoded linear f32 (2x2 quads sharing one chroma neighbourhood for
//               4:2:0) into a wave-private LDS strip, then the horizontal Lanczos of exactly those two rows into a ring of
//               f16 rows M (LDS) — no barrier between the two, only the wave's own in-order LDS stream;
//   * resolve — every output row whose vertical window is now complete: vertical Lanczos over M, sRGB encode, one coalesced
//               256 B row store per wave.
// A source row is converted and filtered horizontally exactly once per strip (no vertical halo), the per-block setup
// (tables, horizontal weights) is paid once per ~240 output rows.  The node texture (RGBA8, input-sized) and the
// Rgba16Float intermediate of the reference never exist in HBM; their quantisation (u8, f16) is applied in registers at
// the same points.
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
        // Jobs are collected first and launched afterwards: a table this call (ctx->weight_call) already handed to a pending job
        // is never a victim — rebuilding it in place or freeing it would give that job wrong weights or a dangling pointer.
        // The cache grows instead (bounded by 2 x max_layouts distinct tables per call).
        if (ctx->weight_tables.size() >= 64)
            for (auto &t : ctx->weight_tables)
                if (t.last_call != ctx->weight_call && (!victim || t.last_use < victim->last_use)) victim = &t;
        if (!victim) {
            ctx->weight_tables.emplace_back();
            victim = &ctx->weight_tables.back();
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
    hit->last_call = ctx->weight_call;
    out->first = (const int *)hit->dev;
    out->wsum = (const float *)hit->dev + n;
    out->w = (const float *)hit->dev + 2 * (size_t)n;
    out->taps = taps;
    return SMR_OK;
}

// ------------------------------------------------------------------ kernel
constexpr int TW = 64;          // strip width (one lane per output column)
constexpr int CH = 16;          // source rows per chunk: one row pair per wave
constexpr int MR = 64;          // rows of the f16 intermediate ring (power of two)
constexpr int A_WAVES = 8;
constexpr int A_THREADS = A_WAVES * 64;
constexpr int VR_MAX = 64;      // output rows whose vertical weights are resident per chunk (one ballot wide)
static_assert(CH == 2 * A_WAVES, "one row pair per wave and chunk");

struct IngestJob {
    SurfView yp, up, vp;  // planar source planes (chroma views carry the logical chroma size)
    SurfView dst;         // RGBA8 tile, dst-sized
    int src_w, src_h;
    int full_range;
    int fast420;          // 4:2:0 with even luma size and dword-aligned planes: LDS-staged 2x2-quad conversion path
    int nv12;             // `up` is the interleaved UV plane of an NV12 frame (2 bytes per chroma texel), `vp` is unused
    int ablate;           // profiling only (SMR_ABLATE): 1 skip convert, 2 skip H taps, 4 skip V taps, 8 skip staging, 16 dispatch only
    int taps_h, taps_v;
    float scale_h, off_h, scale_v, off_v;  // axis mappings (resampler.rs:36-48): source texels per output texel, crop offset
    const float *wsum_h; const float *w_h;  // device weight tables: wsum[n], w[taps][n] (tap-major)
    const float *wsum_v; const float *w_v;
    int tw;               // strip width in columns: 64, or 32 when that is what keeps two workgroups per CU (LDS)
    int strips_x;         // strips of the tile (launch_ingest splits their rows over the blocks)
    int nc_max;           // LDS capacity: columns of a source strip
    int vr;               // output rows with resident vertical weights per chunk
    int defer8;           // resolve output rows in multiples of 8 (one per wave) while the ring has room for the stragglers
};

// raw-footprint staging geometry (bytes)
__host__ __device__ inline int raw_y_stride(int nc_max) { return (nc_max + 8 + 3) & ~3; }
__host__ __device__ inline int raw_c_stride(int nc_max) { return ((nc_max >> 1) + 12 + 3) & ~3; }
constexpr int RAW_C_ROWS = CH / 2 + 2;

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
    // Work is measured in "row units": job j owns units [unit_prefix[j], unit_prefix[j+1]) = strips_x * dst.h rows, strip-major.
    // Block b takes the contiguous range [b * units_per_block, ...): every block of the launch gets the same number of output
    // rows (two blocks per CU, all resident at once) whatever the number and size of the jobs; a range that crosses a strip
    // boundary is processed as two pieces.
    int unit_prefix[MAX_JOBS_PER_LAUNCH + 1];
    int n_jobs;
    int units_per_block;
};

// Output rows [oy0, oy1) of strip `strip` of job J.  TWT = strip width in columns: 64 (one lane per column, two source rows per
// lane in the horizontal pass, one output row per wave in the vertical pass) or 32 (the two half-waves take one source row /
// one output row each) — the narrow strip halves the LDS footprint of the source strips and of the ring at large scale factors.
template <int TWT>
__device__ __forceinline__ void ingest_strip(const IngestJob &J, int strip, int oy0, int oy1, const float *__restrict__ tables, u8 *smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx0 = strip * TWT;
    const int tw = min(TWT, J.dst.w - tx0);
    const int taps_h = J.taps_h, taps_v = J.taps_v;
    const int ncm = J.nc_max;
    const int sw = J.src_w, sh = J.src_h;
    const int VR = J.vr, VRa = (VR + 3) & ~3;

    // ---- LDS carve (every region is a multiple of 16 B)
    float *s_tab = (float *)smem;                          // SMR_TABLE_FLOATS: dec | thr | enc
    float *s_n255 = s_tab + SMR_TABLE_FLOATS;              // 256: u8 / 255
    float *s_ylut = s_n255 + 256;                          // 256: luma u8 -> range-expanded y
    float *s_wh = s_ylut + 256;                            // [taps_h][TWT]
    float *s_wsh = s_wh + ((taps_h * TWT + 3) & ~3);        // [TWT]  wsum
    float *s_rsh = s_wsh + TWT;                             // [TWT]  1 / wsum
    int *s_fh = (int *)(s_rsh + TWT);                       // [TW]
    float *s_wv = (float *)(s_fh + TWT);                    // [3][taps_v][VRp]  (tap-major like the global table; three chunks deep)
    const int VRp = VR <= 32 ? 32 : 64;
    const int wv_sz = taps_v * VRp;
    int *s_fv = (int *)(s_wv + 3 * wv_sz);                 // [3][VRa]
    float *s_wsv = (float *)(s_fv + 3 * VRa);              // [3][VRa]
    float *s_rsv = s_wsv + 3 * VRa;                        // [3][VRa]
    float4 *S_all = (float4 *)(s_rsv + 3 * VRa);           // [A_WAVES][2][nc_max] linear RGBA, wave-private strips
    uint2 *M = (uint2 *)(S_all + (size_t)A_WAVES * 2 * ncm);  // [MR][TWT] half4 ring, row r lives in slot (r - R_lo) & (MR - 1)
    u8 *raw0 = (u8 *)(M + (size_t)MR * TWT);                // [2]{ Y [CH][ys], U [RAW_C_ROWS][cs], V [RAW_C_ROWS][cs] }  (fast420 only)
    const int ys = raw_y_stride(ncm), cs = raw_c_stride(ncm);
    const int raw_sz = CH * ys + 2 * RAW_C_ROWS * cs;

    // ---- the strip's source footprint (first[] is non-decreasing in the output coordinate), computed from the mapping
    //      itself (same f32 sequence as the weight tables) — no dependent load
    int c_lo = clampi(lanczos_first(tx0, J.scale_h, J.off_h), 0, sw - 1);
    const int c_hi = clampi(lanczos_first(tx0 + tw - 1, J.scale_h, J.off_h) + taps_h - 1, 0, sw - 1);
    int R_lo = clampi(lanczos_first(oy0, J.scale_v, J.off_v), 0, sh - 1);
    const int R_hi = clampi(lanczos_first(oy1 - 1, J.scale_v, J.off_v) + taps_v - 1, 0, sh - 1);
    if (J.fast420) {
        // 2x2 conversion quads start on odd luma coordinates (they share one 2x2 chroma neighbourhood)
        c_lo -= (c_lo & 1) ^ 1;
        R_lo -= (R_lo & 1) ^ 1;
    }
    const int NC = c_hi - c_lo + 1;
    const int nq = (NC + 1) >> 1;                          // quad columns
    const int cbase = max(c_lo, 0) & ~3;                   // luma staging keeps global dword alignment
    const int qx0 = (c_lo + 1) >> 1;                       // chroma index of the first quad column
    const int cb = max(qx0 - 1, 0) & ~3;                   // first staged chroma column (dword aligned)
    const int chi = min(qx0 + nq - 1, J.up.w - 1);         // last chroma column any quad reads

    // ---- prologue: tables and the horizontal weights of the strip (once per block)
    for (int i = tid; i < SMR_TABLE_FLOATS; i += A_THREADS) s_tab[i] = tables[i];
    if (tid < 256) {
        // u8 -> f32 conversions done once per table entry with the same IEEE operations the per-pixel path uses
        const float v = (float)tid / 255.0f;
        s_n255[tid] = v;
        s_ylut[tid] = J.full_range ? v : clampf((v - (16.0f / 255.0f)) / 0.85882352941f, 0.0f, 1.0f);
    }
    // weight tables are stored tap-major in global memory (w[t * n + i]): a strip's slice of every tap is contiguous
    for (int i = tid; i < taps_h * TWT; i += A_THREADS) {
        const int t = i / TWT, x = i % TWT;
        s_wh[i] = x < tw ? J.w_h[(size_t)t * J.dst.w + tx0 + x] : 0.0f;
    }
    if (tid < tw) {
        s_fh[tid] = lanczos_first(tx0 + tid, J.scale_h, J.off_h);
        const float ws = J.wsum_h[tx0 + tid];
        s_wsh[tid] = ws;
        s_rsh[tid] = 1.0f / ws;
    }
    const float *s_dec = s_tab, *s_thr = s_tab + 256;
    float4 *S = S_all + (size_t)wave * 2 * ncm;  // this wave's two-row strip

    // ---- staging geometry (block-uniform) and the software pipeline
    //      The global loads of chunk k+1 are issued before chunk k's arithmetic and land in LDS after it.  One dword of luma,
    //      one of each chroma plane and one vertical weight per thread cover the usual footprints; larger ones
    //      (ndw > 32, ndc > lanes per chroma row, taps_v * VRp > 512) take the unpipelined path.
    //      NV12: the interleaved UV rows (2 B per chroma texel) are staged as they are into the U|V region, rows of 2 * cs bytes.
    const int cm = J.nv12 ? 2 : 1;                        // staged bytes per chroma texel
    const int csr = cs * cm;                              // staged chroma row stride (bytes)
    const int csh = J.nv12 ? 5 : 4;                       // chroma staging: 32 (NV12) / 16 lanes per row, 16 / 32 rows >= np + 1
    const int ndw = ((c_hi - cbase) >> 2) + 1;            // luma dwords per row: columns [cbase, c_hi]
    const int ndc = (((chi - cb + 1) * cm - 1) >> 2) + 1; // chroma dwords per row: columns [cb, chi]
    const bool stage = J.fast420 && !(J.ablate & 8);
    const bool piped = ndw <= 32 && ndc <= (1 << csh) && wv_sz <= A_THREADS;
    const int vsh = VRp == 32 ? 5 : 6;
    u32 py = 0u, pu = 0u, pv = 0u;
    float pw = 0.0f, pws = 1.0f;
    u8 *rawY = raw0, *rawU = rawY + CH * ys, *rawV = rawU + RAW_C_ROWS * cs;  // the current chunk's raw buffer (re-pointed per chunk)
    // rows are pitched to >= 4 B multiples (fast420), so reading the last dword of a row never leaves the allocation
    auto issue = [&](int base, int y_next) {
        const int e = min(base + CH - 1, R_hi);
        const int np = (e - base + 2) >> 1;
        if (stage) {
            const int rr = wave * 2 + (lane >> 5), row = base + rr, d = lane & 31;  // luma: half a wave per row (128 B)
            if (row >= 0 && row <= e && d < ndw) py = *(const u32 *)(J.yp.ptr + (size_t)row * J.yp.pitch + cbase + 4 * d);
            const int j = tid >> csh, d2 = tid & ((1 << csh) - 1);
            if (j <= np && d2 < ndc) {
                const int cy = clampi(((base + 1) >> 1) - 1 + j, 0, J.up.h - 1);
                pu = *(const u32 *)(J.up.ptr + (size_t)cy * J.up.pitch + cb * cm + 4 * d2);
                if (!J.nv12) pv = *(const u32 *)(J.vp.ptr + (size_t)cy * J.vp.pitch + cb + 4 * d2);
            }
        }
        const int nv = min(VR, oy1 - y_next);
        const int t = tid >> vsh, j = tid & (VRp - 1);
        if (tid < wv_sz && j < nv) pw = J.w_v[(size_t)t * J.dst.h + y_next + j];
        if (wave == A_WAVES - 1 && lane < nv) pws = J.wsum_v[y_next + lane];
    };
    auto land = [&](int base, int nv, float *wvb, float *wsvb, float *rsvb) {
        const int e = min(base + CH - 1, R_hi);
        const int np = (e - base + 2) >> 1;
        if (stage) {
            const int rr = wave * 2 + (lane >> 5), row = base + rr, d = lane & 31;
            if (row >= 0 && row <= e && d < ndw) *(u32 *)(rawY + (size_t)rr * ys + 4 * d) = py;
            const int j = tid >> csh, d2 = tid & ((1 << csh) - 1);
            if (j <= np && d2 < ndc) {
                *(u32 *)(rawU + j * csr + 4 * d2) = pu;
                if (!J.nv12) *(u32 *)(rawV + j * cs + 4 * d2) = pv;
            }
        }
        if (tid < wv_sz && (tid & (VRp - 1)) < nv) wvb[tid] = pw;
        if (wave == A_WAVES - 1 && lane < nv) { wsvb[lane] = pws; rsvb[lane] = 1.0f / pws; }
    };
    auto stage_unpiped = [&](int base, int y_next, int nv, float *wvb, float *wsvb, float *rsvb) {
        const int e = min(base + CH - 1, R_hi);
        const int np = (e - base + 2) >> 1;
        if (stage) {
            const int rr = wave * 2 + (lane >> 5), row = base + rr;
            for (int d = lane & 31; d < ndw; d += 32)
                if (row >= 0 && row <= e) *(u32 *)(rawY + (size_t)rr * ys + 4 * d) = *(const u32 *)(J.yp.ptr + (size_t)row * J.yp.pitch + cbase + 4 * d);
            const int j = tid >> 4;
            if (j <= np) {
                const int cy = clampi(((base + 1) >> 1) - 1 + j, 0, J.up.h - 1);
                for (int d = tid & 15; d < ndc; d += 16) {
                    *(u32 *)(rawU + j * csr + 4 * d) = *(const u32 *)(J.up.ptr + (size_t)cy * J.up.pitch + cb * cm + 4 * d);
                    if (!J.nv12) *(u32 *)(rawV + j * cs + 4 * d) = *(const u32 *)(J.vp.ptr + (size_t)cy * J.vp.pitch + cb + 4 * d);
                }
            }
        }
        for (int i = tid; i < wv_sz; i += A_THREADS) {
            const int t = i >> vsh, j = i & (VRp - 1);
            if (j < nv) wvb[i] = J.w_v[(size_t)t * J.dst.h + y_next + j];
        }
        if (wave == A_WAVES - 1 && lane < nv) {
            const float ws = J.wsum_v[y_next + lane];
            wsvb[lane] = ws;
            rsvb[lane] = 1.0f / ws;
        }
    };

    // resolve: vertical Lanczos (pass 2) over the ring + sRGB encode + store of output rows [yr, yr + n); weights in buffer b.
    // A wave takes 64 / TWT output rows per step (lane = column, upper half-wave = the next row when TWT == 32).
    constexpr int RPW = 64 / TWT;
    const int col = lane & (TWT - 1), sub = lane / TWT;
    auto resolve = [&](int yr, int n, int b) {
        if (J.ablate & 128) return;
        __builtin_amdgcn_s_setprio(2);
        const float *wvb = s_wv + b * wv_sz;
        const int *fvb = s_fv + b * VRa;
        const float *wsvb = s_wsv + b * VRa, *rsvb = s_rsv + b * VRa;
        for (int j0 = wave * RPW; j0 < n; j0 += A_WAVES * RPW) {
            const bool live = j0 + sub < n && col < tw;
            const int j = j0 + sub < n ? j0 + sub : j0;  // dead lanes shadow a valid row
            const int y = yr + j;
            const int fv = fvb[j];
            float sx_ = 0.f, sy_ = 0.f, sz_ = 0.f;
            const float *wv = wvb + j;  // tap t at wv[t * VRp]
            const int s0 = (fv - R_lo) & (MR - 1);
            if (!live || (J.ablate & 4)) {
            } else if (fv >= 0 && fv + taps_v - 1 <= sh - 1 && s0 + taps_v <= MR) {
                const uint2 *pm = M + (size_t)s0 * TWT + col;  // the window is contiguous in the ring
#pragma unroll 2
                for (int t = 0; t < taps_v; t++) {
                    const float wgt = wv[t * VRp];
                    const float4 m = half4_to_float4(pm[(size_t)t * TWT]);
                    sx_ = __builtin_fmaf(m.x, wgt, sx_); sy_ = __builtin_fmaf(m.y, wgt, sy_); sz_ = __builtin_fmaf(m.z, wgt, sz_);
                }
            } else {
                for (int t = 0; t < taps_v; t++) {
                    const float wgt = wv[t * VRp];
                    const int r = (clampi(fv + t, 0, sh - 1) - R_lo) & (MR - 1);
                    const float4 m = half4_to_float4(M[(size_t)r * TWT + col]);
                    sx_ = __builtin_fmaf(m.x, wgt, sx_); sy_ = __builtin_fmaf(m.y, wgt, sy_); sz_ = __builtin_fmaf(m.z, wgt, sz_);
                }
            }
            if (live) {
                const float ws = wsvb[j], rs = rsvb[j];
                const u32 r8 = srgb_encode8(div_cr(sx_, ws, rs), s_thr), g8 = srgb_encode8(div_cr(sy_, ws, rs), s_thr),
                          b8 = srgb_encode8(div_cr(sz_, ws, rs), s_thr);
                *(u32 *)(J.dst.ptr + (size_t)y * J.dst.pitch + (size_t)(tx0 + col) * 4) = r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // One barrier per chunk.  Phase k (between barriers k and k+1): every wave converts + filters its row pair of chunk k into
    // the ring AND resolves its share of the output rows that chunk k-1 completed (their window lies in rows <= e_{k-1}, all
    // written before barrier k).  Slower waves may still be in phase k-1 while a fast wave prepares chunk k+1, hence the raw
    // footprint is double-buffered and the vertical weights are kept three chunks deep.
    int y_done = oy0;  // next output row without a resolve slot (uniform across the block: every wave derives the same counts)
    int chunk = 0, vb = 0;
    int pend_y = oy0, pend_n = 0, pend_vb = 0;  // rows completed by the previous chunk, resolved in this phase
    if (piped) issue(R_lo, oy0);
    for (int base = R_lo; base <= R_hi; base += CH, chunk++, vb = vb == 2 ? 0 : vb + 1) {
        const int e = min(base + CH - 1, R_hi);  // last source row this chunk produces
        const int np = (e - base + 2) >> 1;      // row pairs
        rawY = raw0 + (chunk & 1) * raw_sz; rawU = rawY + CH * ys; rawV = rawU + RAW_C_ROWS * cs;
        float *wvb = s_wv + vb * wv_sz;
        int *fvb = s_fv + vb * VRa;
        float *wsvb = s_wsv + vb * VRa, *rsvb = s_rsv + vb * VRa;
        const int nv = min(VR, oy1 - y_done);    // candidate output rows of this chunk

        // ---- (a) which candidates does this chunk complete?  first[] is non-decreasing, so the ready rows are a prefix:
        //      every lane evaluates the mapping of one candidate, one ballot counts them (no LDS, known before the arithmetic)
        const int fv_l = lanczos_first(y_done + (lane < nv ? lane : 0), J.scale_v, J.off_v);
        const bool ready = lane < nv && min(fv_l + taps_v - 1, sh - 1) <= e;
        const unsigned long long rdy = __ballot(ready);
        int n_emit = rdy == ~0ull ? 64 : __builtin_ctzll(~rdy);
        if (e < R_hi && J.defer8) n_emit &= ~7;  // stragglers wait for the next chunk (one row per wave keeps the waves level)
        // ---- (b) this chunk's raw footprint and vertical weights reach LDS
        if (piped) land(base, nv, wvb, wsvb, rsvb);
        else stage_unpiped(base, y_done, nv, wvb, wsvb, rsvb);
        if (wave == A_WAVES - 1 && lane < nv) fvb[lane] = fv_l;
        __syncthreads();
        if (piped && base + CH <= R_hi) issue(base + CH, y_done + n_emit);


        // ---- (c) convert + horizontal Lanczos: one row pair per wave
        // odd waves resolve first, even waves last: the LDS-bound filter loops of one half overlap the VALU-bound conversion
        // of the other half instead of all eight waves hitting the same unit in the same phase
        if (wave & 1) resolve(pend_y, pend_n, pend_vb);
        const int pr = wave;
        if (pr < np) {
            const int y0 = base + 2 * pr, y1 = y0 + 1;
            __builtin_amdgcn_s_setprio(1);
            if (J.ablate & 1) {
            } else if (J.fast420) {
                // chroma rows indexed by chroma x * cm (NV12: U at even, V at odd bytes of the interleaved row)
                const u8 *ua = rawU + pr * csr - cb * cm, *ub = ua + csr;
                const u8 *va = J.nv12 ? ua + 1 : rawV + pr * cs - cb, *vbp = va + csr;
                const u8 *yr0 = rawY + (size_t)(2 * pr) * ys - cbase, *yr1 = yr0 + ys;                        // index by luma x
                const bool ok0 = y0 >= 0, ok1 = y1 <= e;
                const int cw1 = J.up.w - 1;
                // (the range of the frame is decided once per row pair, not per pixel: two copies of the loop)
                auto quads = [&](auto full_c) {
                constexpr bool FULL = decltype(full_c)::value;
                for (int qc = lane; qc < nq; qc += 64) {
                    const int x0 = c_lo + 2 * qc, x1 = x0 + 1;  // x0 odd (or -1), x1 even
                    const int i0 = max(qx0 - 1 + qc, 0) << (cm - 1), i1 = min(qx0 + qc, cw1) << (cm - 1);  // clamp-to-edge chroma taps
                    const float u00 = s_n255[ua[i0]], u01 = s_n255[ua[i1]], u10 = s_n255[ub[i0]], u11 = s_n255[ub[i1]];
                    const float v00 = s_n255[va[i0]], v01 = s_n255[va[i1]], v10 = s_n255[vbp[i0]], v11 = s_n255[vbp[i1]];
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
                            const float ue = FULL ? uu : expand_chroma(uu), ve = FULL ? vv : expand_chroma(vv);
                            S[sx - c_lo] = yuv_expanded_to_linear(s_ylut[yr0[sx]], ue, ve, s_dec);
                        }
                        if (ok1) {
                            const float uu = ut * 0.25f + ubt * 0.75f, vv = vt * 0.25f + vbt * 0.75f;
                            const float ue = FULL ? uu : expand_chroma(uu), ve = FULL ? vv : expand_chroma(vv);
                            S[ncm + sx - c_lo] = yuv_expanded_to_linear(s_ylut[yr1[sx]], ue, ve, s_dec);
                        }
                    }
                }
                };
                if (J.full_range) quads(std::true_type{});
                else quads(std::false_type{});
            } else {
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int sy = rr ? y1 : y0;
                    if (sy < 0 || sy > e) continue;
                    const float tv = ((float)sy + 0.5f) / (float)sh;
                    const u8 *yrow = J.yp.ptr + (size_t)sy * J.yp.pitch;
                    for (int col = lane; col < NC; col += 64) {
                        const int sx = c_lo + col;
                        const float tu = ((float)sx + 0.5f) / (float)sw;
                        const float yy = (float)yrow[sx] / 255.0f;
                        const float uu = sample_plane_bilinear(J.up, cm, 0, tu, tv);
                        const float vv = J.nv12 ? sample_plane_bilinear(J.up, 2, 1, tu, tv) : sample_plane_bilinear(J.vp, 1, 0, tu, tv);
                        const u32 p = yuv_to_rgb_px(yy, uu, vv, J.full_range != 0);
                        S[rr * ncm + col] = make_float4(s_dec[p & 0xff], s_dec[(p >> 8) & 0xff], s_dec[(p >> 16) & 0xff], 1.0f);
                    }
                }
            }
            // the strip is private to this wave and a wave's LDS operations complete in order: a fence is all that is needed
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- horizontal Lanczos of the two rows into the f16 ring (pass 1 of the separable plan).
            //      All four channels ride in packed FMAs; alpha comes out as (sum w)/(sum w) == 1 exactly.
            __builtin_amdgcn_s_setprio(3);  // the LDS-bound loop issues ahead of the other waves' vector work
            if constexpr (TWT == 64) {
                if (lane < tw) {  // one lane per column, both rows
                    const int fh = s_fh[lane];
                    const float wsh = s_wsh[lane], rsh = s_rsh[lane];
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float *wcol = s_wh + lane;
                    if (J.ablate & 2) {
                    } else if (fh >= 0 && fh + taps_h - 1 <= sw - 1) {
                        // interior: no edge clamp, consecutive texels
                        const float4 *pa = S + (fh - c_lo), *pb = pa + ncm;
#pragma unroll 2
                        for (int t = 0; t < taps_h; t++) {
                            const float wgt = wcol[t * TWT];
                            const float4 ta = pa[t], tb = pb[t];
                            a.x = __builtin_fmaf(ta.x, wgt, a.x); a.y = __builtin_fmaf(ta.y, wgt, a.y);
                            a.z = __builtin_fmaf(ta.z, wgt, a.z); a.w = __builtin_fmaf(ta.w, wgt, a.w);
                            b.x = __builtin_fmaf(tb.x, wgt, b.x); b.y = __builtin_fmaf(tb.y, wgt, b.y);
                            b.z = __builtin_fmaf(tb.z, wgt, b.z); b.w = __builtin_fmaf(tb.w, wgt, b.w);
                        }
                    } else {
                        const float4 *Sa = S - c_lo, *Sb = S + ncm - c_lo;
                        for (int t = 0; t < taps_h; t++) {
                            const float wgt = wcol[t * TWT];
                            const int s = clampi(fh + t, 0, sw - 1);
                            const float4 ta = Sa[s], tb = Sb[s];
                            a.x = __builtin_fmaf(ta.x, wgt, a.x); a.y = __builtin_fmaf(ta.y, wgt, a.y);
                            a.z = __builtin_fmaf(ta.z, wgt, a.z); a.w = __builtin_fmaf(ta.w, wgt, a.w);
                            b.x = __builtin_fmaf(tb.x, wgt, b.x); b.y = __builtin_fmaf(tb.y, wgt, b.y);
                            b.z = __builtin_fmaf(tb.z, wgt, b.z); b.w = __builtin_fmaf(tb.w, wgt, b.w);
                        }
                    }
                    if (y0 >= 0) M[(size_t)((y0 - R_lo) & (MR - 1)) * TWT + lane] = float4_to_half4(div_cr(a.x, wsh, rsh), div_cr(a.y, wsh, rsh), div_cr(a.z, wsh, rsh), div_cr(a.w, wsh, rsh));
                    if (y1 <= e) M[(size_t)((y1 - R_lo) & (MR - 1)) * TWT + lane] = float4_to_half4(div_cr(b.x, wsh, rsh), div_cr(b.y, wsh, rsh), div_cr(b.z, wsh, rsh), div_cr(b.w, wsh, rsh));
                }
            } else {
                // 32-column strip: the lower half-wave filters row y0, the upper half-wave row y1
                const int yrow = sub ? y1 : y0;
                if (col < tw && (sub ? y1 <= e : y0 >= 0)) {
                    const int fh = s_fh[col];
                    const float wsh = s_wsh[col], rsh = s_rsh[col];
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float *wcol = s_wh + col;
                    const float4 *Srow = S + sub * ncm;
                    if (J.ablate & 2) {
                    } else if (fh >= 0 && fh + taps_h - 1 <= sw - 1) {
                        const float4 *pa = Srow + (fh - c_lo);
#pragma unroll 2
                        for (int t = 0; t < taps_h; t++) {
                            const float wgt = wcol[t * TWT];
                            const float4 ta = pa[t];
                            a.x = __builtin_fmaf(ta.x, wgt, a.x); a.y = __builtin_fmaf(ta.y, wgt, a.y);
                            a.z = __builtin_fmaf(ta.z, wgt, a.z); a.w = __builtin_fmaf(ta.w, wgt, a.w);
                        }
                    } else {
                        const float4 *Sa = Srow - c_lo;
                        for (int t = 0; t < taps_h; t++) {
                            const float wgt = wcol[t * TWT];
                            const float4 ta = Sa[clampi(fh + t, 0, sw - 1)];
                            a.x = __builtin_fmaf(ta.x, wgt, a.x); a.y = __builtin_fmaf(ta.y, wgt, a.y);
                            a.z = __builtin_fmaf(ta.z, wgt, a.z); a.w = __builtin_fmaf(ta.w, wgt, a.w);
                        }
                    }
                    M[(size_t)((yrow - R_lo) & (MR - 1)) * TWT + col] = float4_to_half4(div_cr(a.x, wsh, rsh), div_cr(a.y, wsh, rsh), div_cr(a.z, wsh, rsh), div_cr(a.w, wsh, rsh));
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        // ---- (d) this wave's share of the rows the previous chunk completed
        if (!(wave & 1)) resolve(pend_y, pend_n, pend_vb);
        pend_y = y_done; pend_n = n_emit; pend_vb = vb;
        y_done += n_emit;
    }
    __syncthreads();
    resolve(pend_y, pend_n, pend_vb);  // rows completed by the last chunk
}

// (second bound = waves per SIMD: two 8-wave workgroups per CU need 4, i.e. at most 128 VGPRs — without it a small change in
//  the tap loops' unrolling silently halves the occupancy)
__global__ __launch_bounds__(A_THREADS, 4) void k_ingest_resample(const IngestArgs args, const float *__restrict__ tables) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    if (args.jobs[0].ablate & 16) return;  // profiling: pure dispatch cost of this grid
    const int total = args.unit_prefix[args.n_jobs];
    // XCD-aware order: workgroup ids go round-robin over the 8 XCDs, each with its own L2.  Virtual id v puts ids that share an
    // XCD next to each other in the unit space, so neighbouring strips (which share source cache lines) — with 8 equal jobs a
    // whole input — go through one L2 and every line is fetched from memory once instead of once per XCD.
    const int per_xcd = (int)gridDim.x >> 3;  // the grid is a multiple of 8
    const int v = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    int u = v * args.units_per_block;
    const int u_end = min(u + args.units_per_block, total);
    bool first = true;
    while (u < u_end) {
        int j = 0;
        while (j + 1 < args.n_jobs && args.unit_prefix[j + 1] <= u) j++;
        const IngestJob &J = args.jobs[j];
        const int local = u - args.unit_prefix[j];
        const int strip = local / J.dst.h, oy0 = local - strip * J.dst.h;
        const int oy1 = min(J.dst.h, oy0 + (u_end - u));
        if (!first) __syncthreads();  // the previous piece's LDS is dead only once every wave has left it
        if (J.tw == 32) ingest_strip<32>(J, strip, oy0, oy1, tables, smem);
        else ingest_strip<64>(J, strip, oy0, oy1, tables, smem);
        u += oy1 - oy0;
        first = false;
    }
}

int ingest_vr(float scale_v) {
    // rows that become ready per chunk (<= CH / scale + 1) plus the stragglers deferred from the previous one (<= 7)
    const float s = scale_v > 0.0f ? scale_v : 1.0f;
    return (int)ceilf((float)CH / s) + 9;
}

size_t ingest_lds_bytes(const IngestJob &J) {
    const size_t vra = ((size_t)J.vr + 3) & ~(size_t)3;
    const size_t vrp = J.vr <= 32 ? 32 : 64;
    const size_t tw = (size_t)J.tw;
    size_t floats = SMR_TABLE_FLOATS + 256 + 256 + (((size_t)J.taps_h * tw + 3) & ~(size_t)3) + 3 * tw + 3 * (size_t)J.taps_v * vrp + 9 * vra;
    size_t bytes = floats * 4 + (size_t)A_WAVES * 2 * J.nc_max * 16 + (size_t)MR * tw * 8;
    if (J.fast420) bytes += 2 * ((size_t)CH * raw_y_stride(J.nc_max) + 2 * (size_t)RAW_C_ROWS * raw_c_stride(J.nc_max));
    return (bytes + 15) & ~(size_t)15;
}

bool is_planar_yuv(u32 fmt) { return fmt <= SMR_FRAME_PLANAR_YUVJ420; }

// What wave A covers: planar YUV frames, separable plan, no box pre-reduction, horizontal pass first, and a vertical
// window that fits the ring (CH new rows + the window + one output row's advance).
bool can_fuse_ingest(const smr_frame *f, const smr_resample_plan &plan) {
    if (!f || !f->planes[0] || !f->planes[1]) return false;
    const bool nv12 = f->format == SMR_FRAME_NV12;
    if (!nv12 && !(is_planar_yuv(f->format) && f->planes[2])) return false;
    if (!(plan.kind == 2 && plan.levels[0] == 0 && plan.levels[1] == 0 && plan.axis[0] == 0)) return false;
    const int taps_v = host_taps(plan.scale[1]);
    // ring: the chunk being written + the previous chunk + the window of the oldest unresolved row
    return ingest_vr(plan.scale[1]) <= VR_MAX && 2 * CH + taps_v + (int)ceilf(plan.scale[1]) + 2 <= MR;
}

int make_ingest_job(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile, IngestJob *out) {
    WeightPtrs wh, wv;
    int rc = get_weights(ctx, plan.scale[0], plan.offset[0], (int)tile->w, &wh);
    if (rc != SMR_OK) return rc;
    rc = get_weights(ctx, plan.scale[1], plan.offset[1], (int)tile->h, &wv);
    if (rc != SMR_OK) return rc;
    IngestJob &J = *out;
    J.nv12 = f->format == SMR_FRAME_NV12 ? 1 : 0;
    J.yp = view_of(f->planes[0]); J.up = view_of(f->planes[1]); J.vp = J.nv12 ? J.up : view_of(f->planes[2]);
    J.dst = view_of(tile);
    J.src_w = (int)f->width; J.src_h = (int)f->height;
    J.full_range = f->format == SMR_FRAME_PLANAR_YUVJ420 ? 1 : 0;
    // the staged path reads whole dwords of luma and chroma rows: needs 4 B-aligned planes whose pitch covers the last dword
    auto dword_ok = [](const SurfView &p, u32 bytes) { return (p.pitch % 4) == 0 && (((uintptr_t)p.ptr) % 4) == 0 && p.pitch >= ((bytes + 3u) & ~3u); };
    const bool is420 = f->format == SMR_FRAME_PLANAR_YUV420 || f->format == SMR_FRAME_PLANAR_YUVJ420 || J.nv12;
    const bool chroma_ok = J.nv12 ? dword_ok(J.up, (f->width / 2) * 2) : (dword_ok(J.up, f->width / 2) && dword_ok(J.vp, f->width / 2));
    J.fast420 = (is420 && f->width % 2 == 0 && f->height % 2 == 0 && f->width >= 2 && f->height >= 2 && dword_ok(J.yp, f->width) &&
                 chroma_ok) ? 1 : 0;
    J.ablate = ctx->ablate;
    J.taps_h = wh.taps; J.taps_v = wv.taps;
    J.scale_h = plan.scale[0]; J.off_h = plan.offset[0];
    J.scale_v = plan.scale[1]; J.off_v = plan.offset[1];
    J.wsum_h = wh.wsum; J.w_h = wh.w;
    J.wsum_v = wv.wsum; J.w_v = wv.w;
    // strip width: 64 columns unless only the 32-column variant fits two workgroups per CU (LDS <= 80 KB each) — the source
    // strips grow with the scale factor
    J.vr = ingest_vr(plan.scale[1]);
    if (J.vr > (int)tile->h) J.vr = (int)tile->h;
    if (J.vr > VR_MAX) J.vr = VR_MAX;
    for (int tw : {64, 32}) {
        J.tw = tw;
        // +1: the quad path aligns the footprint start down to an odd coordinate
        J.nc_max = (int)ceilf((float)tw * fmaxf(plan.scale[0], 0.0f)) + wh.taps + 3;
        if (J.nc_max > J.src_w + 1) J.nc_max = J.src_w + 1;
        if (ingest_lds_bytes(J) <= 80 * 1024) break;
    }
    if (ingest_lds_bytes(J) > 80 * 1024) {  // neither fits twice: the wide strip has the smaller halo
        J.tw = 64;
        J.nc_max = (int)ceilf(64.0f * fmaxf(plan.scale[0], 0.0f)) + wh.taps + 3;
        if (J.nc_max > J.src_w + 1) J.nc_max = J.src_w + 1;
    }
    {  // tests / profiling: SMR_OPT_INGEST_STRIP_WIDTH pins the strip width
        const int tw = ctx->force_tw;
        if (tw == 32 || tw == 64) {
            J.tw = tw;
            J.nc_max = (int)ceilf((float)tw * fmaxf(plan.scale[0], 0.0f)) + wh.taps + 3;
            if (J.nc_max > J.src_w + 1) J.nc_max = J.src_w + 1;
        }
    }
    J.strips_x = ((int)tile->w + J.tw - 1) / J.tw;
    J.defer8 = (2 * CH + wv.taps + (int)ceilf(8.0f * fmaxf(plan.scale[1], 0.0f)) + 2 <= MR) ? 1 : 0;
    return SMR_OK;
}

int launch_ingest(smr_ctx *ctx, std::vector<IngestJob> &jobs) {
    if (!ctx->valu_attr_set) {  // per device, hence per ctx
        SMR_HIP(ctx, hipFuncSetAttribute((const void *)k_ingest_resample, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->valu_attr_set = true;
    }
    StageScope scope(ctx, SMR_STAGE_FUSED_INGEST);
    ctx->kernel_launches[SMR_KERNEL_INGEST_VALU]++;
    for (size_t j0 = 0; j0 < jobs.size(); j0 += MAX_JOBS_PER_LAUNCH) {
        const size_t nj = jobs.size() - j0 < (size_t)MAX_JOBS_PER_LAUNCH ? jobs.size() - j0 : (size_t)MAX_JOBS_PER_LAUNCH;
        IngestArgs args;
        memset(&args, 0, sizeof(args));
        size_t lds = 0;
        int total = 0;
        for (size_t j = 0; j < nj; j++) {
            const IngestJob &J = jobs[j0 + j];
            args.jobs[j] = J;
            args.unit_prefix[j] = total;
            total += J.strips_x * J.dst.h;
            size_t b = ingest_lds_bytes(J);
            lds = b > lds ? b : lds;
        }
        args.unit_prefix[nj] = total;
        args.n_jobs = (int)nj;
        if (lds > 160 * 1024) return smr_fail(ctx, SMR_ERR_INTERNAL, "k_ingest_resample: %zu B of LDS needed", lds);
        // every block is resident at once (two per CU at <= 80 KB of LDS): split the launch's output rows evenly over
        // 2 x CUs blocks, but keep at least 32 rows per block (a piece re-converts the `taps` rows of its vertical halo)
        // 1/16 of the CUs stay free for the compose kernel of the frame in flight on another stream: measured on MI355X the
        // ingest kernel alone loses 1.5 % on 240 instead of 256 CUs, the pipelined frame rate gains 3 %
        // (SMR_INGEST_RESERVE_CUS overrides, profiling only)
        const int reserve = ctx->ingest_reserve_cus >= 0 && ctx->ingest_reserve_cus < ctx->cu_count ? ctx->ingest_reserve_cus : ctx->cu_count / 16;
        int blocks = 2 * (ctx->cu_count - reserve);
        int upb = (total + blocks - 1) / blocks;
        if (upb < 32) upb = 32;
        blocks = ((total + upb - 1) / upb + 7) & ~7;  // whole rounds over the 8 XCDs (surplus blocks find no units and leave)
        args.units_per_block = upb;
        if (blocks > 0)
            hipLaunchKernelGGL(k_ingest_resample, dim3((unsigned)blocks), dim3(A_THREADS), lds, ctx->stream, args, ctx->d_tables);
        SMR_HIP(ctx, hipGetLastError());
    }
    return SMR_OK;
}

}  // namespace
