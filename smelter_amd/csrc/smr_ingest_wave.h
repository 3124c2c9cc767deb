// smr_ingest_wave.h — wave A of the hot path, second generation: k_ingest_wave (included by smr_fused.hip only).
//
// The matrix-core resampler: an RGBA8 node texture (what the exact input converter wrote: smr_convert_420.h — the default route) or,
// with SMR_INGEST_LAB_FUSED, a planar 4:2:0 / NV12 frame converted on the fly -> dst-sized sRGB RGBA8 tile, i.e.
// (planar_yuv_to_rgba.wgsl:35-58 followed by) the two Lanczos3 passes of transformations/layout/resample.wgsl:31-87 with their
// Rgba16Float intermediate (layout/resampler.rs:25-28), both passes as banded GEMMs on v_mfma_f32_16x16x32_f16 with f16-pair
// operands — but organised so that a wave never waits for another wave:
//
//   * A WAVE owns two neighbouring 16-column output tiles (a "column pair", 32 output columns) over a range of 16-row output tiles
//     and streams the source rows in chunks of 16.  There is no barrier in the main loop and nothing but tables in LDS.
//   * Conversion feeds the matrix cores from registers.  Lane (m, q) = (lane & 15, lane >> 4) converts the 4x1 pixel block of chunk
//     row m, texels 4 (4 j + q) .. + 3 of the pair's source window: its (hi, lo) f16 pairs ARE the lane's eight K values of the
//     A operand of k-step j (m_convert_px).  The node texture never exists, not even in LDS.
//   * Pass 1: H[r][x] = sum_k T[r][k] Wh[k][x] accumulates over the k-steps of the pair's window into one accumulator set per
//     tile (weights (w_hi, w_hi) and (w_lo, 0) per texel pair, read from LDS where the workgroup keeps its pair's band).
//   * Pass 2 straight from registers: the f32 accumulator of pass 1 holds, per lane, rows 4 q .. 4 q + 3 of one output column
//     of the chunk.  Rounded to f16 (the reference's Rgba16Float store) that is half of an A operand of the second GEMM
//     O[x][y] = sum_r H[x][r] Wv[r][y] — provided K runs over the rows in the order the lanes happen to hold them.  K is a
//     summation index, so the weight bands are simply built in that order (k_build_wave_weights, axis 3): K slot (q, e) of the k-step
//     made of chunks (c0, c1) is row 4 q + e of c0 for e < 4 and row 4 q + e - 4 of c1 for e >= 4.  A tile's window is the 2 KV
//     chunks that end with the chunk of its last row; chunk c lives in ring slot c mod 2 KV (an absolute grid, so one band per
//     tile serves every piece), the ring is 2 KV register pairs per channel and tile.  No f16 intermediate in LDS either.
//   * Pass-2 weights are f16 pairs too (w_hi + w_lo, two MFMAs on the same A operand): against the reference's two passes applied to
//     the node texture this kernel's conversion produces, every byte of every content class is within 1 LSB (single-f16 pass-2
//     weights are not, on white noise: tools/mfma_precision_sim.py).
//   * The reference's own quantisation points are kept: u8 node texture, f16 (RTNE) between the passes, u8 sRGB tile.
//
// Builds (template <NKS_T, KV_T, FL>): the k-step class (generic, or fixed counts with the loops unrolled / software-pipelined) and,
// in FL: 1 two always-zero fragments skipped | 2048 direct output | 4096 NV12-capable staging | 8192 the source is an RGBA8 node
// texture (16-byte loads straight into the conversion layout, decode table only: 4:2:2 / 4:4:4 / packed YUV after the exact
// converter, opaque surfaces) | + 16384 that texture is RGBA16F, linear light (box-pre-reduced plans) | + 65536 it has an alpha
// channel (four channels) | 32768 single-axis plan: pass 1's f32 sums are encoded and stored directly (no f16 rounding, no pass 2) |
// + 131072 (with 8192) the node texture is RGB12: 12 bytes per four pixels, R x 4, G x 4, B x 4 (what k_yuv420_to_rgba writes for nodes only
// this kernel reads: alpha is 1 everywhere and not stored) — one 12-byte load per lane and k-step instead of 16 |
// 262144 (round 6, the default route of 4:2:0 frames): the source is the FRAME — planar 4:2:0 (+ 4096: NV12) — and the node texture never
// reaches memory: at the top of a chunk the wave converts the chunk's 16 rows x window with the exact converter's own block arithmetic
// (smr_convert_420.h: lane (bc, br) = (lane & 15, lane >> 4) owns the 4 x 4 block of block row br, block column bc — 64 blocks = the 16 x 64 texels
// of a 4-k-step window; two blocks per lane for 8 k-steps), bit for bit k_yuv420_to_rgba's bytes, into an RGB12 chunk in LDS (3 KB per wave),
// from which the k-steps read their blocks exactly like the RGB12 node builds read theirs from memory.  Chunks start on rows 16 c (origin 0:
// whole 4 x 4 blocks; vertical bands of axis 5).  Half the route's memory traffic (no 50 MB node written and read back), one launch fewer.
//
// Work split: a workgroup = W_WAVES waves on the same column pair (they share its pass-1 band in LDS), each with its own vertical
// piece; workgroups are ordered pair-fastest within a band of rows, so neighbouring pairs read the same source lines at the same
// time on the same XCD.  Per-wave state is registers (~48 for the ring, 24 accumulators); LDS per workgroup = decode / encode
// tables (9.5 KB) + the pair's band (4 KB per k-step) + 2 KB of raw Y/U/V staging per wave.
#pragma once

#include "smr_ingest_common.h"
#include "smr_convert_420.h"  // the exact 4:2:0 converter's block arithmetic: the plane-source builds (FL & 262144) convert in the wave

#include <cmath>
#include <type_traits>
#include <vector>

namespace {

#ifndef SMR_WAVE_WAVES
#define SMR_WAVE_WAVES 2
#endif
#ifndef SMR_WAVE_MIN_WAVES
#define SMR_WAVE_MIN_WAVES 2   // waves per SIMD the register allocation must leave room for
#endif
#ifndef SMR_WAVE_MIN_WAVES_RG
#define SMR_WAVE_MIN_WAVES_RG SMR_WAVE_MIN_WAVES   // ... for the narrow class on an opaque RGBA8 node texture (the default route's kernel: 170 VGPRs; A/B knob)
#endif
#ifndef SMR_WAVE_RG_EARLY_WAIT
#define SMR_WAVE_RG_EARLY_WAIT 1   // node-texture builds: wait for the outstanding loads at the top of a chunk (0: A/B — configs[3] 128.6 -> 109.5 us, configs[2] unchanged)
#endif
#ifndef SMR_WAVE_STAGGER
#define SMR_WAVE_STAGGER 0   // A/B: waves in odd hardware slots of a SIMD start s_sleep(N) later (N x 64 cycles), so that the two waves of a SIMD are not in the
                             // same phase (LDS gathers / matrix cores / encode) at the same time
#endif
#ifndef SMR_WAVE_DEFER_STORES
#define SMR_WAVE_DEFER_STORES 1  // node-texture builds: a tile row's stores go out at the top of the NEXT chunk, behind its wait (A/B knob; see wave_piece)
#endif
#ifndef SMR_DIRECT_ABL
#define SMR_DIRECT_ABL 0  // profiling builds only: direct output without 1 its conversion arithmetic, 2 its stores
#endif
#ifndef SMR_WAVE_ABL
#define SMR_WAVE_ABL 0  // profiling builds only (tools/variant.sh): 1 no LUT gathers, 2 no pass-1 MFMAs, 4 no conversion, 8 no pass 2 / encode, 16 no stores, 32 no staging
#endif
#ifndef SMR_WAVE_EXPLICIT_WAIT
#define SMR_WAVE_EXPLICIT_WAIT 0  // 0: leave the wait for the staged loads to the compiler (it waits for the loads only, not for the stores behind them)
#endif
#ifndef SMR_WAVE_PIPE
#define SMR_WAVE_PIPE 1  // class build: the k-steps of a chunk as a software pipeline (LDS reads of the next step and the weights of the last
                         // one go out before a block is converted, the last step's MFMAs follow the conversion): a wave does not wait for LDS
#endif
#ifndef SMR_WAVE_PIPE_FENCE
#define SMR_WAVE_PIPE_FENCE 1  // keep the compiler from re-ordering the pipeline's phases
#endif
#ifndef SMR_WAVE_TIMING
#define SMR_WAVE_TIMING 0  // profiling build (tools/variant.sh): shader cycles per phase of the first wave of every workgroup -> WArgs::dbg
#endif
#ifndef SMR_WAVE_SETPRIO
#define SMR_WAVE_SETPRIO 0   // A/B: s_setprio level around the pass-1 MFMAs of the pipelined builds (measured: see profiles/r03_setprio.txt)
#endif
#ifndef SMR_WAVE_ONE_TILE
#define SMR_WAVE_ONE_TILE 0  // profiling experiment: 1 = every wave works on the first tile of its pair only (half of the output is not written)
#endif
#define W_NTI (SMR_WAVE_ONE_TILE ? 1 : 2)
#ifndef SMR_WAVE_B_REGS
#define SMR_WAVE_B_REGS 0   // 1: narrow class builds keep the pair's pass-1 weights in registers for the whole piece (no LDS band, no re-reads):
                            // + 31 VGPRs, 12 LDS reads fewer per chunk, measured neutral (57.5 vs 57.6 us): the kernel is bound by vector-ALU issue
#endif
#ifndef SMR_WAVE_PREFETCH_B
#define SMR_WAVE_PREFETCH_B 0  // 1: a k-step's weight fragments are read from LDS before its block is converted (A/B knob)
#endif
constexpr int W_WAVES = SMR_WAVE_WAVES;   // waves per workgroup (same column pair, consecutive vertical pieces)
constexpr int W_THREADS = 64 * W_WAVES;
constexpr int W_NKS_MAX = 8;              // k-steps (16 texels as hi/lo pairs) of a pair's source window
constexpr int W_KV_MAX = 4;               // k-steps (two chunks of 16 rows) of a pass-2 window
constexpr int W_WSPAN = 136;              // >= 16 * W_NKS_MAX, >= 32 * W_KV_MAX

// ------------------------------------------------------------------ geometry (host + device: one f32 sequence)
// chunk c = source rows 16 c + org .. 16 c + org + 15.  org = -1 (every build but the plane-source ones): row 0 of a chunk is an odd luma row — rows
// (2 p + 1, 2 p + 2) share chroma rows p, p + 1 (the laboratory builds' on-the-fly conversion); org = 0 (the plane-source builds, FL & 262144):
// a chunk is four rows of the exact converter's 4 x 4 blocks (smr_convert_420.h).  The vertical bands are built per origin (axis 3 / axis 5).
__host__ __device__ inline int w_chunk_of_row(int r, int org = -1) { return (r - org) >> 4; }
__host__ __device__ inline bool w_axis_vertical(int axis) { return axis == 3 || axis == 5; }
__host__ __device__ inline int w_axis_org(int axis) { return axis == 5 ? 0 : -1; }
__host__ __device__ inline int w_posmod(int a, int n) { const int r = a % n; return r < 0 ? r + n : r; }
__host__ __device__ inline int w_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
__host__ __device__ inline int w_taps(float scale) {
    const float kernel_scale = scale > 1.0f ? scale : 1.0f;
    const int taps = (int)ceilf(2.0f * (3.0f * kernel_scale)) + 1;
    return taps > MAX_TAPS ? MAX_TAPS : taps;
}
// first / last source texel that carries a weight for outputs [o0, o1]
__host__ __device__ inline void w_span(int o0, int o1, float scale, float offset, int taps, int n_src, int *lo, int *hi) {
    *lo = w_clampi(lanczos_first(o0, scale, offset), 0, n_src - 1);
    *hi = w_clampi(lanczos_first(o1, scale, offset) + taps - 1, 0, n_src - 1);
}

// A column pair: tiles 2 pair, 2 pair + 1 (the second may not exist).  base = first texel of the window (multiple of 4), tile i's
// band starts at k-step klo[i] (-1: no such tile) and needs kt[i] k-steps; nks = k-steps of the whole window.
struct WPairGeom {
    int base, klo[2], kt[2], last, nks;
};
// (single: axis 4 — one tile per unit, for windows too wide for a pair: the unit's second tile does not exist)
__host__ __device__ inline WPairGeom w_pair_geometry(int pair, float scale, float offset, int taps, int n_dst, int n_src, bool single = false) {
    WPairGeom g;
    const int n_tiles = (n_dst + 15) >> 4;
    g.base = 0; g.last = 0;
    for (int i = 0; i < 2; i++) {
        const int t = single ? pair : 2 * pair + i;
        g.klo[i] = -1; g.kt[i] = 0;
        if (t >= n_tiles || (single && i == 1)) continue;
        const int o0 = 16 * t, o1 = o0 + 15 < n_dst - 1 ? o0 + 15 : n_dst - 1;
        int lo, hi;
        w_span(o0, o1, scale, offset, taps, n_src, &lo, &hi);
        if (i == 0) g.base = lo & ~3;
        if (lo < g.base) lo = g.base;  // (cannot happen: the first texel grows with the output column)
        g.klo[i] = (lo - g.base) >> 4;
        g.kt[i] = ((hi - g.base) >> 4) - g.klo[i] + 1;
        g.last = hi > g.last ? hi : g.last;
    }
    g.nks = ((g.last - g.base) >> 4) + 1;
    return g;
}
// A 16-row output tile of pass 2: first / last chunk that carries a weight
__host__ __device__ inline void w_vtile_chunks(int t, float scale, float offset, int taps, int n_dst, int n_src, int *cs, int *ce, int org = -1) {
    const int o0 = 16 * t, o1 = o0 + 15 < n_dst - 1 ? o0 + 15 : n_dst - 1;
    int lo, hi;
    w_span(o0, o1, scale, offset, taps, n_src, &lo, &hi);
    *cs = w_chunk_of_row(lo, org);
    *ce = w_chunk_of_row(hi, org);
}

// host twin of the geometry the builder uses (same f32 sequence: lanczos_first is __host__ __device__)
// *k01 (axis 2, may be null): every pair's first tile stays clear of the window's last k-step and its second tile of the first one
// (true for the benchmark's scale: the class build then skips those two fragments)
inline void wave_band_geometry(float scale, float offset, int n_dst, int n_src, int axis, int *K, int *nks, bool *k01 = nullptr) {
    const int taps = w_taps(scale);
    const int n_tiles = (n_dst + 15) / 16;
    int k = 1, n = 1;
    bool pat = true;
    if (axis == 2 || axis == 4) {
        const bool single = axis == 4;
        const int units = single ? n_tiles : (n_tiles + 1) / 2;
        for (int p = 0; p < units; p++) {
            const WPairGeom g = w_pair_geometry(p, scale, offset, taps, n_dst, n_src, single);
            for (int i = 0; i < 2; i++) k = g.kt[i] > k ? g.kt[i] : k;
            n = g.nks > n ? g.nks : n;
        }
        for (int p = 0; p < units; p++) {  // (against the job's k-step count n: the class build runs min(n, 4) .. 4 steps)
            const WPairGeom g = w_pair_geometry(p, scale, offset, taps, n_dst, n_src, single);
            if (g.klo[0] + g.kt[0] - 1 > 2) pat = false;          // tile 0 reaches k-step 3
            if (g.klo[1] >= 0 && g.klo[1] < 1) pat = false;       // tile 1 starts in k-step 0
        }
        if (k01) *k01 = pat;
    } else {
        for (int t = 0; t < n_tiles; t++) {
            int cs, ce;
            w_vtile_chunks(t, scale, offset, taps, n_dst, n_src, &cs, &ce, w_axis_org(axis));
            const int need = (ce - cs + 2) / 2;
            k = need > k ? need : k;
        }
    }
    *K = !w_axis_vertical(axis) ? n : k;  // (axis 2 / 4: the band is dense over the unit's window)
    *nks = n;
}

// ------------------------------------------------------------------ weight bands (device cache, one launch per call for all misses)
// axis 2 (pass 1, per column pair):  meta int4 (base, last, klo0 | khi0 << 8, klo1 | khi1 << 8) — tile i's band touches k-steps
//   klo_i .. khi_i of the pair's window (0xffff: no such tile);  frag[pair][tile i][k-step j < NKS][hi | lo][64 lanes], dense over the
//   window (zero fragments where a tile's band does not reach: the benchmark's class runs them all and has no branch in its loop):
//   lane l holds W[k = 8 (l >> 4) + e][n = l & 15], texel = base + 16 j + (k >> 1); the (w_hi, w_hi) fragment multiplies the
//   (t_hi, t_lo) pair, the (w_lo, 0) fragment adds t_hi w_lo (t_lo w_lo is below 2^-22).
// axis 3 (pass 2, per 16-row output tile):  meta int2 (first chunk, last chunk);  frag[tile][k-step p < KV][hi | lo][64 lanes]:
//   lane l = (output row n = l & 15, q = l >> 4), element e: ring slot s = 2 p + (e >> 2), chunk cc = the chunk of the tile's window
//   [ce - 2 KV + 1, ce] with cc mod 2 KV == s, source row 16 cc + org + 4 q + (e & 3) (org = -1; axis 5: the same with org = 0).
struct WWBuild {
    float scale, offset;
    int taps, n_dst, n_src, axis, K, unit0;  // unit0: first workgroup of this band (units = pairs or tiles)
    void *meta;
    uint4 *frag;
};
constexpr int MAX_WWBUILDS = 32;
struct WWBatch {
    WWBuild b[MAX_WWBUILDS];
    int n;
};

#ifdef __HIPCC__
__global__ __launch_bounds__(64) void k_build_wave_weights(const WWBatch args) {
    __shared__ float s_w[16][W_WSPAN];
    __shared__ _Float16 s_q[16][W_WSPAN], s_r[16][W_WSPAN];
    int bi = 0;
    while (bi + 1 < args.n && args.b[bi + 1].unit0 <= (int)blockIdx.x) bi++;
    const WWBuild &B = args.b[bi];
    const float scale = B.scale, offset = B.offset;
    const int taps = B.taps, n_dst = B.n_dst, n_src = B.n_src, K = B.K;
    const int u = (int)blockIdx.x - B.unit0, lane = threadIdx.x;
    const int n16 = lane & 15, q = lane >> 4;
    const bool horiz = !w_axis_vertical(B.axis), single = B.axis == 4;
    const int org = w_axis_org(B.axis);
    const int n_sub = horiz ? 2 : 1;
    WPairGeom G;
    int v_cs = 0, v_ce = 0;
    if (horiz) G = w_pair_geometry(u, scale, offset, taps, n_dst, n_src, single);
    else w_vtile_chunks(u, scale, offset, taps, n_dst, n_src, &v_cs, &v_ce, org);
    for (int sub = 0; sub < n_sub; sub++) {
        const int tile = horiz && !single ? 2 * u + sub : u;
        // origin of the window in source texels / rows, and its length
        const int wlo = v_ce - 2 * K + 1;  // (axis 3) first chunk of the window; may be negative: those chunks do not exist
        const int origin = horiz ? G.base : 16 * wlo + org;
        const int span = horiz ? 16 * K : 32 * K;
        for (int i = lane; i < 16 * W_WSPAN; i += 64) {
            (&s_w[0][0])[i] = 0.0f;
            (&s_q[0][0])[i] = (_Float16)0.0f;
            (&s_r[0][0])[i] = (_Float16)0.0f;
        }
        __syncthreads();
        const bool present = !horiz || G.klo[sub] >= 0;
        if (present && lane < 16 && 16 * tile + lane < n_dst) {
            float w[MAX_TAPS];
            float ws;
            const int first = lanczos_weights(16 * tile + lane, scale, offset, taps, w, &ws);
            for (int i = 0; i < taps; i++) {  // clamp-to-edge folded into the band: taps that clamp onto one texel are summed
                const int idx = w_clampi(first + i, 0, n_src - 1) - origin;
                if (idx >= 0 && idx < span) s_w[lane][idx] += w[i] / ws;
            }
            // two f16 terms per weight: hi = f16(w), lo = f16(w - hi) — the pair carries 22 bits
            for (int i = 0; i < span; i++) {
                const _Float16 qh = (_Float16)s_w[lane][i];
                s_q[lane][i] = qh;
                s_r[lane][i] = (_Float16)(s_w[lane][i] - (float)qh);
            }
        }
        __syncthreads();
        for (int j = 0; j < K; j++) {
            f16x8 v, r;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (horiz) {
                    const int kk = 32 * j + 8 * q + e;
                    v[e] = s_q[n16][kk >> 1];
                    r[e] = (kk & 1) ? (_Float16)0.0f : s_r[n16][kk >> 1];
                } else {
                    const int s = 2 * j + (e >> 2);
                    const int cc = wlo + w_posmod(s - wlo, 2 * K);
                    const int idx = 16 * (cc - wlo) + 4 * q + (e & 3);
                    v[e] = s_q[n16][idx];
                    r[e] = s_r[n16][idx];
                }
            }
            const size_t f = horiz ? (((size_t)u * 2 + sub) * K + j) * 2 : ((size_t)u * K + j) * 2;
            B.frag[f * 64 + lane] = __builtin_bit_cast(uint4, v);
            B.frag[(f + 1) * 64 + lane] = __builtin_bit_cast(uint4, r);
        }
        __syncthreads();
    }
    if (lane == 0) {
        if (horiz) {
            int r[2];
            for (int i = 0; i < 2; i++) r[i] = G.klo[i] < 0 ? 0xffff : (G.klo[i] | ((G.klo[i] + G.kt[i] - 1) << 8));
            ((int4 *)B.meta)[u] = make_int4(G.base, G.last, r[0], r[1]);
        } else {
            ((int2 *)B.meta)[u] = make_int2(v_cs, v_ce);
        }
    }
}
#endif  // __HIPCC__

// ------------------------------------------------------------------ kernel
struct WJob {
    SurfView yp, up, vp;  // source planes (chroma views carry the chroma size; NV12: `up` = interleaved UV, `vp` aliases it)
    SurfView dst;         // RGBA8 tile, dst-sized
    int src_w, src_h;
    MConv conv;
    const int4 *h_meta; const uint4 *h_frag;
    const int2 *v_meta; const uint4 *v_frag;
    int NKS, KV;          // k-steps: of the widest pair window, per pass-2 window
    int n_pairs, n_htiles, n_vtiles;
    int k01;              // host only: the job's bands fit the class build's FL & 1 pattern
    int pieces;           // vertical pieces per column pair (a multiple of W_WAVES)
    int nv12;
    int layer, ox, oy;    // direct output: the layer this tile is blitted by (-1 = none) and its (even) position in the output frame
    int perp;             // single-axis builds (FL & 32768): output row y shows source row y + perp
    int single;           // the pass-1 band has one tile per unit (axis 4: windows too wide for a pair): unit u = output columns 16 u ..
    int full;             // plane-source builds (FL & 262144): the frame is full range (J420)
};

constexpr int MAX_WJOBS_PER_LAUNCH = 16;
struct WArgs {
    WJob jobs[MAX_WJOBS_PER_LAUNCH];
    int wg_prefix[MAX_WJOBS_PER_LAUNCH + 1];  // workgroups per job = n_pairs * pieces / W_WAVES, piece-group major, pair fastest
    int n_jobs;
    int b_bytes;      // LDS bytes reserved for a pair's pass-1 band
    int raw_bytes;    // ... for one wave's raw footprint
    const MDirect *direct;  // device record (rides behind the layout list), nullptr = off
    unsigned long long *dbg;  // profiling builds only (SMR_WAVE_TIMING)
};

constexpr int W_OFF_THR = M_LUT_ENTRIES * 4;
#ifndef SMR_WAVE_ENC1
#define SMR_WAVE_ENC1 1  // 1: one table gather per encoded value (6.5 KB of buckets), 0: estimate byte + threshold (2.7 KB) — A/B
#endif
constexpr int W_OFF_B = W_OFF_THR + (SMR_WAVE_ENC1 ? SMR_ENC_ENTRIES * 4 : (SMR_TABLE_FLOATS - 256) * 4);  // the encode buckets (smr_internal.h: SMR_LUT16_WORDS) | thr[257] + pad + estimate bytes
static_assert(W_OFF_B % 16 == 0, "the band must start on a 16-byte boundary");
__host__ __device__ inline int w_ys(int nks) { return 4 * nks + 1; }  // staged luma dwords per chunk row (+ 1: rows fall on different banks)
__host__ __device__ inline int w_cs(int nks) { return 2 * nks + 2; }  // staged chroma dwords per row: columns base/2 - 1 .. base/2 + 8 nks, from a 4-aligned start
__host__ __device__ inline int w_raw_bytes(int nks) { return 4 * (16 * w_ys(nks) + 2 * 9 * w_cs(nks)); }
__host__ __device__ inline int w_band_bytes(int nks) { return 2 * nks * 2 * 64 * 16; }
constexpr int W_FX_TABLE_BYTES = 2048;                                                     // plane-source builds: ylut + nlut behind the band
__host__ __device__ inline int w_fx_node_bytes(int nks) { return ((16 * (12 * nks + 1) * 4) + 15) & ~15; }  // ... and a wave's RGB12 chunk: 16 rows of 12 nks + 1 dwords

#ifdef __HIPCC__

// Output tiles [vt0, vt1] of column pair `pair` of job J, by one wave.  NKS_T / KV_T: the k-step counts of the launch's class (every
// pair window within NKS_T k-steps, pass-2 windows of exactly KV_T: no branch in the chunk loop); 0 = read them from the job
// (loops unrolled to the maximum and predicated).  FL: 1 class build only: no pair's first tile reaches k-step 3 and no second tile
// starts in k-step 0 (those two zero fragments are skipped), 2048 direct output, 4096 NV12-capable staging.
// `finish_prologue`: stores the workgroup's tables into LDS and meets the other waves — called once, after this wave's first global
// loads are in flight and before its first LDS access.
// srgb_encode8 (smr_internal.h) with the clamp as one median: the operand is a finite matrix-core sum, never a NaN to be quieted first.
// One table gather per value instead of two (estimate byte, then the threshold above it): a bucket — 7 mantissa bits — holds at most one
// threshold, so its entry carries the code of its lowest x and where in the bucket the code steps up.  The clamped operand decides both
// (below 2^-13: bucket 0, code 0, no threshold; at and above 1: the last bucket's top, past its threshold): the values of srgb_encode8.
// Round 6: the bucket is addressed by bits [26:16] of the clamped operand (the clamp keeps bits [31:27] fixed: no subtraction), and its entry
// is (code << 16) | (0xffff - threshold offset): adding the operand's low 16 bits carries into the code exactly when the operand lies past the
// threshold.  Five vector instructions per value (median, shift, mask, mask, add) against ten; the sum's byte 2 is the code.
__device__ __forceinline__ u32 w_encode_sum(float x, const u32 *__restrict__ enc) {
    const u32 b = __float_as_uint(dev_fmed3(x, 1.220703125e-4f, 0.99999994f));  // [0x39000000, 0x3f7fffff]
#if SMR_WAVE_ENC1
    const u32 e = (enc - 0x100)[(b >> 16) & 0x7ffu];  // (0x3900 & 0x7ff = 0x100: the table's first bucket)
    return e + (b & 0xffffu);
#else
    const float *thr = (const float *)enc;
    const u8 *est = (const u8 *)(thr + SMR_ENC_OFFSET_FROM_THR);
    u32 c = est[(b - 0x39000000u) >> 16];
    c += thr[c + 1] <= x ? 1u : 0u;
    return c << 16;
#endif
}
__device__ __forceinline__ u32 w_encode8(float x, const u32 *__restrict__ enc) { return w_encode_sum(x, enc) >> 16; }
// three encoded channels -> an opaque RGBA8 pixel: two byte permutes (byte 2 of each sum; selector 0x0d = 0xff)
__device__ __forceinline__ u32 w_encode_px(float r, float g, float b, const u32 *__restrict__ enc) {
    const u32 rg = dev_perm(w_encode_sum(g, enc), w_encode_sum(r, enc), 0x0c0c0602u);
    return dev_perm(w_encode_sum(b, enc), rg, 0x0d060100u);
}

template <int NKS_T, int KV_T, int FL, typename Pro>
__device__ __forceinline__ void wave_piece(const WJob &J, const MDirect *__restrict__ Dg, int pair, int vt0, int vt1, u8 *smem, u32 b_off, u32 raw_off,
                                           unsigned long long *dbg, Pro finish_prologue, int J_b_bytes) {
    const int lane = threadIdx.x & 63, l16 = lane & 15, lq = lane >> 4;
#if SMR_WAVE_TIMING
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    const bool timing = dbg != nullptr && threadIdx.x < 64;
#define W_MARK(ph)                                                   \
    do {                                                             \
        if (timing) {                                                \
            __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) */     \
            const unsigned long long now = __builtin_readcyclecounter(); \
            tph[ph] += now - tlast;                                  \
            tlast = now;                                             \
        }                                                            \
    } while (0)
#else
#define W_MARK(ph) do { } while (0)
#endif
    // (the narrow class runs its 4 k-steps unconditionally; a wide class (NKS_T = 8) is unrolled to NKS_T steps but lays its LDS out for the
    //  job's own count, so that a 7-step job does not pay for an eighth in LDS and conversions)
    constexpr bool FIXED_NKS = NKS_T > 0 && NKS_T <= 4;
    const int NKS = FIXED_NKS ? NKS_T : J.NKS, KV = KV_T ? KV_T : J.KV;
    constexpr int NKS_N = NKS_T ? NKS_T : W_NKS_MAX, KV_N = KV_T ? KV_T : W_KV_MAX;
    constexpr bool NV = (FL & 4096) != 0, DIRECT = (FL & 2048) != 0, K01 = (FL & 1) != 0;
    // 8192: the source is an RGBA8 node texture with alpha == 1 (a frame of any other format after the exact converter, or an opaque
    // surface): the block of a k-step is one 16-byte load per lane, already in the conversion's layout (lane (m, q): row m, texels
    // 4 q .. 4 q + 3), held in registers — no LDS staging — and its "conversion" is the decode table alone.
    // 16384 (with 8192): that node texture is RGBA16F in linear light — what the box pre-reduction of a plan with shrink factors from 4
    // leaves (resampler.rs: downsample.wgsl into an Rgba16Float texture): the f16 texels ARE the operand's hi halves, lo = 0.
    constexpr bool FX = (FL & 262144) != 0;  // the source is a 4:2:0 frame, converted exactly in the wave (header)
    static_assert(!FX || (NKS_T != 0 && SMR_WAVE_PIPE && !(FL & (8192 | 16384 | 32768 | 65536 | 131072 | 2048))), "plane-source builds: class builds of the pipelined loop only");
    constexpr bool RG = (FL & 8192) != 0, RH = RG && (FL & 16384) != 0;
    // 131072 (with 8192): the node texture as 12-byte groups of four pixels (smr_convert_420.h rgb12): lane (m, q) loads the group of texels
    // 4 q .. 4 q + 3 — a dword per channel
    constexpr bool RP = RG && !RH && (FL & 131072) != 0;
    static_assert(!RP || !(FL & 65536), "RGB12 nodes carry no alpha");

    // 32768: a single-axis plan (ResampledChild with one pass, resampler.rs:123-145 — only the width changes): pass 1's f32 sums are
    // encoded and stored as they are, row for row; there is no f16 rounding and no pass 2.  The job's vertical band (scale 1) only
    // supplies the chunk ranges of the pieces.  (Height-only plans run on the transposed frame.)
    constexpr bool SA = (FL & 32768) != 0;
    // 65536 (with 8192): the RGBA8 node texture has an alpha channel (premultiplied: a text run, an image, a nested layout node, a
    // BGRA / ARGB frame): alpha is a fourth channel through both passes — linear, not sRGB: a / 255 as an f16 pair from its own 256-entry
    // table, unorm8 on the way out (resample.wgsl filters all four channels of the premultiplied texel).
    constexpr bool AL = RG && (FL & 65536) != 0;
    constexpr int NCH = AL ? 4 : 3;
    const u32 alpha_tab = b_off + (u32)J_b_bytes;  // LDS: behind the pass-1 band (the raw staging area of the other builds)
    constexpr int RG_N = RG ? NKS_N : 1;
    uint4 rg[RG_N];  // block j of the chunk at hand; refilled with the next chunk's block j as soon as it has been converted
    uint4 rg2[RH ? RG_N : 1];  // (RGBA16F: texels 2, 3 of the block; rg holds 0, 1)
    const u32 *s_thr = (const u32 *)(smem + W_OFF_THR);  // (the encode buckets)
    const uint4 *Bs = (const uint4 *)(smem + b_off);

    // ---- pair geometry
    const int4 hm = J.h_meta[pair];
    const int base = hm.x;
    const int nks = min(((hm.y - base) >> 4) + 1, NKS);  // k-steps of this pair's window
    const int klo[2] = {hm.z & 0xff, hm.w & 0xff}, khi[2] = {(hm.z >> 8) & 0xff, (hm.w >> 8) & 0xff};  // (0xff / 0xff: no such tile)
    const int tx0 = (J.single ? 16 : 32) * pair;
    const int d_w = J.dst.w, d_h = J.dst.h;
    u8 *const d_ptr = J.dst.ptr;
    const u32 d_pitch = J.dst.pitch;

    // ---- staging: loads of the next chunk's raw footprint into registers, landed in LDS after the chunk at hand is converted
    const int ys = w_ys(NKS), cs = w_cs(NKS);
    u32 *const rawY = (u32 *)(smem + raw_off), *const rawU = rawY + 16 * ys, *const rawV = rawU + 9 * cs;
    const u8 *const y_ptr = J.yp.ptr, *const u_ptr = J.up.ptr, *const v_ptr = J.vp.ptr;
    const u32 y_pitch = J.yp.pitch, u_pitch = J.up.pitch, v_pitch = J.vp.pitch;
    const int sw = J.src_w, sh = J.src_h, cw = J.up.w, chh = J.up.h;
    const bool nv = NV && J.nv12 != 0;  // (uniform)
    // luma: lane (row 4 rb + lq, dword 16 cb + l16) of the chunk, rb < 4, cb < ceil(NKS / 4)
    constexpr int NCB = (NKS_N + 3) / 4, NYL = 4 * NCB;
    u32 py[NYL];
    const int sw4 = (sw + 3) & ~3;
    // chroma: 18 (plane, row) tasks; LPRC lanes per row
    constexpr int CSN = 2 * NKS_N + 2, LPRC = CSN <= 16 ? 16 : 32, RPLC = 64 / LPRC, NCL = (18 + RPLC - 1) / RPLC;
    u32 pc[NCL], pc_hi[NV ? NCL : 1];
    const int c_d = lane & (LPRC - 1), c_t = lane / LPRC;  // dword within the staged row, task within the load
    const int cc0 = ((base >> 1) - 1) & ~3;               // first staged chroma column (4-aligned, may be -4)
    const int c_col0 = cc0 + 4 * c_d;                      // first chroma column of this lane's staged dword
    const int c_col0c = w_clampi(c_col0, 0, (cw - 1) & ~3);  // ... of the dword actually loaded (clamp-to-edge)
    const bool c_live = c_d < cs;
    const bool c_edge = c_live && (c_col0 < 0 || c_col0 + 3 > cw - 1);
    // per-lane constants of the loads: luma column offsets; per chroma load the plane base, its pitch and the row within the chunk
    u32 y_col[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) y_col[cb] = (u32)min(base + 4 * (16 * cb + l16), sw4 - 4);
    const u8 *c_base[NCL];
    u32 c_pitch[NCL];
    int c_row[NCL];
#pragma unroll
    for (int k = 0; k < NCL; k++) {
        const int rt = min(RPLC * k + c_t, 17);  // (plane, row) task; tasks past the 18th repeat the last one
        const int plane = rt >= 9 ? 1 : 0;
        c_row[k] = rt - 9 * plane;
        c_base[k] = plane ? v_ptr : u_ptr;
        c_pitch[k] = plane ? v_pitch : u_pitch;
    }
    const u32 c_colb = nv ? 2u * (u32)c_col0c : (u32)c_col0c;
    auto issue = [&](int c) {
        if (RG || FX) return;  // (rg_load, block by block; fx_issue)
        if (SMR_WAVE_ABL & 256) c = J.v_meta[vt0].x;  // profiling: always the same rows (cache hits)
        if (SMR_WAVE_ABL & 128) {                       // profiling: no global loads
#pragma unroll
            for (int k = 0; k < NYL; k++) py[k] = 0x80808080u + (u32)c;
#pragma unroll
            for (int k = 0; k < NCL; k++) pc[k] = 0x80808080u + (u32)c;
            return;
        }
        // (32-bit offsets from the plane bases: a plane is far below 4 GiB; row * pitch is a 24-bit multiply-add with the lane's column)
        const int r0 = 16 * c - 1 + lq;
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
            const u32 row = (u32)min(max(r0 + 4 * rb, 0), sh - 1);
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) py[rb * NCB + cb] = *(const u32 *)(y_ptr + dev_mad24(row, y_pitch, y_col[cb]));
        }
        const int i0 = 8 * c - 1;  // chroma row of the chunk's first row pair
#pragma unroll
        for (int k = 0; k < NCL; k++) {
            const u32 off = dev_mad24((u32)min(max(i0 + c_row[k], 0), chh - 1), c_pitch[k], c_colb);
            pc[k] = *(const u32 *)(c_base[k] + off);
            if (nv) pc_hi[NV ? k : 0] = *(const u32 *)(c_base[k] + off + 4);  // (four chroma texels = eight interleaved bytes; the plane's four are picked when they land)
        }
    };
    auto land = [&]() {
        if (RG || FX) return;
        if (SMR_WAVE_ABL & 64) {  // profiling: no LDS writes (one keeps the loads alive)
            u32 x = 0;
#pragma unroll
            for (int k = 0; k < NYL; k++) x ^= py[k];
#pragma unroll
            for (int k = 0; k < NCL; k++) x ^= pc[k];
            if (x == 0x12345u) rawY[lane] = x;
            return;
        }
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
                if (16 * cb + l16 < 4 * NKS) rawY[(4 * rb + lq) * ys + 16 * cb + l16] = py[rb * NCB + cb];
#pragma unroll
        for (int k = 0; k < NCL; k++) {
            const int rt = RPLC * k + c_t;
            u32 v = pc[k];
            if (nv) v = dev_perm(pc_hi[NV ? k : 0], v, rt >= 9 ? 0x07050301u : 0x06040200u);  // V : U bytes
            if (c_edge) {
                u32 o = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) o |= ((v >> (8 * (w_clampi(c_col0 + b, 0, cw - 1) - c_col0c))) & 0xffu) << (8 * b);
                v = o;
            }
            if (c_live && rt < 18) {
                const int plane = rt >= 9 ? 1 : 0, r = rt - 9 * plane;
                (plane ? rawV : rawU)[r * cs + c_d] = v;
            }
        }
    };

    // ---- conversion: lane (m, q) = (l16, lq) converts chunk row m, texels base + 16 j + 4 q .. + 3 for k-step j
    const MConv K = J.conv;
    const int bi0 = ((base >> 1) - 1 - cc0) + 2 * lq;  // byte of the staged chroma row where this lane's neighbourhood starts (k-step 0)
    const u32 shb = (u32)(bi0 & 3);
    const u32 *const yrow = rawY + l16 * ys + lq;
    const u32 *const urow = rawU + (l16 >> 1) * cs + (bi0 >> 2), *const vrow = rawV + (l16 >> 1) * cs + (bi0 >> 2);
    // Row 0 of a chunk is an odd luma row: 3/4 of chroma row p = row / 2 (weights A); odd chunk rows take 3/4 of row p + 1 (B).
    constexpr u32 WA13 = 0x03010903u, WA31 = 0x01030309u, WB13 = 0x09030301u, WB31 = 0x03090103u;
    const u32 w13 = (l16 & 1) ? WB13 : WA13, w31 = (l16 & 1) ? WB31 : WA31;

    // RGBA source: block j of chunk c (texels past the row's end or rows outside the frame only ever meet zero weights: any readable
    // address will do) ...
    auto rg_load = [&](int c, int j) {
        const u32 row = (u32)min(max(16 * c - 1 + l16, 0), sh - 1);
        const u32 col = (u32)min(base + 16 * j + 4 * lq, sw4 - 4);
        if (RH) {
            // (texels past the row's end are the surface's padding — arbitrary bits, possibly an f16 NaN or infinity, and 0 x NaN is
            //  not 0: they are replaced by zero.  The byte builds need no such care: every byte decodes to a finite value.)
            const u8 *p = y_ptr + dev_mad24(row, y_pitch, 8u * col);
            uint4 t = *(const uint4 *)p, t2 = *(const uint4 *)(p + 16);
            const int nvalid = sw - (int)col;  // >= 1
            if (nvalid < 2) { t.z = 0u; t.w = 0u; }
            if (nvalid < 3) { t2.x = 0u; t2.y = 0u; }
            if (nvalid < 4) { t2.z = 0u; t2.w = 0u; }
            rg2[RH ? (j < RG_N ? j : 0) : 0] = t2;
            return t;
        }
        if (RP) {
            const u32 *p = (const u32 *)(y_ptr + dev_mad24(row, y_pitch, 3u * col));  // (col is a multiple of 4: group col / 4 at byte 12 (col / 4))
            return make_uint4(p[0], p[1], p[2], 0u);
        }
        return *(const uint4 *)(y_ptr + dev_mad24(row, y_pitch, 4u * col));
    };
    // an RGB12 group (a dword of four codes per channel) -> decode table (entries 256 .. 511 of the LUT are the codes themselves)
    auto decode_rgb12 = [&](u32 p0, u32 p1, u32 p2, uint4 (&a)[4]) {
        const u32 pc[3] = {p0, p1, p2};
        u32 o[3][4];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            o[ch][0] = dev_lds_u32(((pc[ch] << 2) & 0x3fcu) + 1024u);
            o[ch][1] = dev_lds_u32(((pc[ch] >> 6) & 0x3fcu) + 1024u);
            o[ch][2] = dev_lds_u32(((pc[ch] >> 14) & 0x3fcu) + 1024u);
            o[ch][3] = dev_lds_u32(((pc[ch] >> 22) & 0x3fcu) + 1024u);
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) a[ch] = make_uint4(o[ch][0], o[ch][1], o[ch][2], o[ch][3]);
    };
    // ... and its texel bytes -> decode table
    auto convert_rgba = [&](int j, uint4 (&a)[4]) {
        const uint4 t = rg[RG ? (j < RG_N ? j : 0) : 0];
        if (RH) {  // texel = (r | g << 16, b | a << 16): hi = the f16 itself, lo = 0
            const uint4 t2 = rg2[RH ? (j < RG_N ? j : 0) : 0];
            a[0] = make_uint4(t.x & 0xffffu, t.z & 0xffffu, t2.x & 0xffffu, t2.z & 0xffffu);
            a[1] = make_uint4(t.x >> 16, t.z >> 16, t2.x >> 16, t2.z >> 16);
            a[2] = make_uint4(t.y & 0xffffu, t.w & 0xffffu, t2.y & 0xffffu, t2.w & 0xffffu);
            if (AL) a[3] = make_uint4(t.y >> 16, t.w >> 16, t2.y >> 16, t2.w >> 16);
            return;
        }
        const u32 px[4] = {t.x, t.y, t.z, t.w};
        u32 o[4][4];
        if (RP) {  // px[ch] = the four texels of channel ch
            decode_rgb12(px[0], px[1], px[2], a);
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (SMR_WAVE_ABL & 1) {  // profiling: no table gathers
                o[0][k] = ((px[k] << 2) & 0x3fcu) + 1024u; o[1][k] = ((px[k] >> 6) & 0x3fcu) + 1024u; o[2][k] = ((px[k] >> 14) & 0x3fcu) + 1024u;
                if (AL) o[3][k] = (px[k] >> 22) & 0x3fcu;
                continue;
            }
            o[0][k] = dev_lds_u32(((px[k] << 2) & 0x3fcu) + 1024u);
            o[1][k] = dev_lds_u32(((px[k] >> 6) & 0x3fcu) + 1024u);
            o[2][k] = dev_lds_u32(((px[k] >> 14) & 0x3fcu) + 1024u);
            if (AL) o[3][k] = dev_lds_u32(((px[k] >> 22) & 0x3fcu) + alpha_tab);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) a[ch] = make_uint4(o[ch][0], o[ch][1], o[ch][2], o[ch][3]);
    };

    // ---- plane-source builds (FX): the chunk's 4 x 4 blocks of the exact converter.  Lane (bc, br) = (l16, lq) owns block row br of the chunk
    //      (rows 16 c + 4 br .. + 3) and block columns bc (+ 16 for windows beyond 4 k-steps) of the pair's window; the blocks' luma dwords
    //      and chroma window rows are requested a chunk ahead (right after the previous chunk was converted: the registers are free then)
    //      and converted at the top of the chunk into the wave's RGB12 chunk in LDS: row r of the chunk at dword r * fx_rs, 3 dwords per group.
    constexpr int FXB = FX ? (NKS_N + 3) / 4 : 1;  // blocks per lane and chunk
    constexpr bool FXNV = FX && NV;                // (a launch of NV12 jobs: launch_wave groups them)
    ConvJob CJ;
    CJ.yp = J.yp; CJ.up = J.up; CJ.vp = J.vp;
    CJ.dst.ptr = nullptr; CJ.dst.pitch = 0u; CJ.dst.w = J.src_w; CJ.dst.h = J.src_h;
    CJ.full = J.full; CJ.nv = FXNV ? 1 : 0; CJ.sx = 1; CJ.sy = 1; CJ.rgb12 = 1; CJ.packed = 0;
    Cv420Win fx_win[FXB];
    int fx_g[FXB];
    u32 fx_y[FXB][4];
    Cv420Raw<FXNV> fx_c[FXB][4];
    const u32 fx_rs = 12u * (u32)NKS + 1u;  // dwords per chunk row (+ 1: rows fall on different banks)
    u32 *const fx_node = (u32 *)(smem + raw_off);
    const float *const fx_ylut = (const float *)(smem + b_off + (u32)J_b_bytes), *const fx_nlut = fx_ylut + 256;
    if (FX) {
#pragma unroll
        for (int sb = 0; sb < FXB; sb++) {
            fx_g[sb] = min((base >> 2) + 16 * sb + l16, (sw >> 2) - 1);  // (columns past the row's end only ever meet zero weights: the last block again)
            fx_win[sb] = cv420_window<FXNV>(CJ, fx_g[sb]);
        }
    }
    auto fx_issue = [&](int c) {
        if (!FX) return;
#pragma unroll
        for (int sb = 0; sb < FXB; sb++) {
            if (16 * sb >= 4 * NKS) continue;  // (uniform: a narrower job of the wide class)
            const int P = 4 * c + lq;
            cv420_load_luma(CJ, fx_g[sb], P, fx_y[sb]);
#pragma unroll
            for (int k = 0; k < 4; k++) fx_c[sb][k] = cv420_load_chroma<FXNV, true>(CJ, fx_win[sb], 2 * P - 1 + k);  // (the build that reads nothing behind a row's last column)
        }
    };
    auto fx_convert = [&](int c) {
        if (!FX) return;
#pragma unroll
        for (int sb = 0; sb < FXB; sb++) {
            if (16 * sb >= 4 * NKS) continue;
            float H[4][2][4];
#pragma unroll
            for (int k = 0; k < 4; k++) cv420_hrow<FXNV>(fx_c[sb][k], fx_win[sb], fx_nlut, H[k]);
            u32 *const dst = fx_node + (u32)(4 * lq) * fx_rs + 3u * (u32)(16 * sb + l16);
            const u32 rs = fx_rs;
            const bool live = 16 * sb + l16 < 4 * NKS;  // (a narrower job of the wide class: its chunk rows hold 4 NKS groups)
            auto sink = [=](int r, int, u32 r4, u32 g4, u32 b4, const u32 (&)[4]) {
                if (live) { dst[(u32)r * rs] = r4; dst[(u32)r * rs + 1u] = g4; dst[(u32)r * rs + 2u] = b4; }
            };
            // (all four rows: rows past the frame's end repeat its last row — they only ever meet zero weights, like the columns)
            if (CJ.full) cv420_rows_to<true, true, true>(CJ, fx_g[sb], 4 * c + lq, fx_y[sb], H, fx_nlut, sink);
            else cv420_rows_to<true, false, true>(CJ, fx_g[sb], 4 * c + lq, fx_y[sb], H, fx_ylut, sink);
        }
    };
    const u32 *const fx_rd = fx_node + (u32)l16 * fx_rs + 3u * (u32)lq;  // this lane's group of k-step 0 (k-step j: + 12 j)

    // ---- pass-2 state: the ring of f16 rows (two chunks per register quad) and the weights of the next tile to finish
    //      (one register vector per tile and channel, written at a uniform runtime index: register-indexed moves, not a select per slot)
    typedef u32 ring_t __attribute__((ext_vector_type(4 * KV_N)));
    ring_t ring[2][4];
#pragma unroll
    for (int i = 0; i < W_NTI; i++)
#pragma unroll
        for (int c = 0; c < NCH; c++) ring[i][c] = (ring_t)(0u);
    auto ring_quad = [&](int i, int ch, int p) { return make_uint4(ring[i][ch][4 * p], ring[i][ch][4 * p + 1], ring[i][ch][4 * p + 2], ring[i][ch][4 * p + 3]); };
    uint4 bvh[KV_N], bvl[KV_N];
    const uint4 *const v_frag = J.v_frag;
    const bool dj = DIRECT && Dg != nullptr && J.layer >= 0;  // (uniform)
    const int d_ox = J.ox, d_oy = J.oy, d_layer = J.layer;
    u32 cls_next[2] = {0xffu, 0xffu};
    auto fetch_bv = [&](int t) {
#pragma unroll
        for (int p = 0; p < KV_N; p++)
            if (p < KV) {
                bvh[p] = v_frag[(((size_t)t * KV + p) * 2) * 64 + lane];
                bvl[p] = v_frag[(((size_t)t * KV + p) * 2 + 1) * 64 + lane];
            }
        if (dj) {  // direct output: the class of the 128x16 output tile this lane's four pixels of tile row t fall into
#pragma unroll
            for (int i = 0; i < W_NTI; i++) {
                const int x = tx0 + 16 * i + 4 * lq, y = 16 * t + l16;
                const int X = d_ox + x, Y = d_oy + y;
                const bool in = y < d_h && x < d_w && X >= 0 && Y >= 0 && X < Dg->yp.w && Y < Dg->yp.h;
                const u32 cl = Dg->cls[in ? (Y >> 4) * Dg->tiles_x + (X >> 7) : 0];
                cls_next[i] = in ? cl : 0xffu;
            }
        }
    };

    // ---- narrow class builds: the pair's pass-1 band (hi / lo fragments of both tiles, every k-step that is not known to be zero) is
    //      loop-invariant: it lives in registers for the whole piece instead of being re-read from LDS chunk after chunk
    constexpr bool B_REGS = FIXED_NKS && SMR_WAVE_PIPE && SMR_WAVE_B_REGS;
    uint4 breg[B_REGS ? 2 : 1][B_REGS ? NKS_N : 1][2];
    if (B_REGS) {
        const uint4 *const src = J.h_frag + (size_t)pair * 2 * J.NKS * 2 * 64;
#pragma unroll
        for (int i = 0; i < W_NTI; i++)
#pragma unroll
            for (int j = 0; j < NKS_N; j++) {
                const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                const bool skip = K01 && ((i == 0 && j == 3) || (i == 1 && j == 0));  // (known zero: never read)
                const bool have = !skip && j < J.NKS;  // (uniform: a job of the class with a narrower window meets zero weights beyond it)
                const int jj = min(j, J.NKS - 1);  // (always a valid address; the value is dropped below: a select of addresses would go through scratch)
                const uint4 fh = skip ? z : src[((i * J.NKS + jj) * 2) * 64 + lane], fl = skip ? z : src[((i * J.NKS + jj) * 2 + 1) * 64 + lane];
                breg[B_REGS ? i : 0][B_REGS ? j : 0][0] = make_uint4(have ? fh.x : 0u, have ? fh.y : 0u, have ? fh.z : 0u, have ? fh.w : 0u);
                breg[B_REGS ? i : 0][B_REGS ? j : 0][1] = make_uint4(have ? fl.x : 0u, have ? fl.y : 0u, have ? fl.z : 0u, have ? fl.w : 0u);
            }
    } else {
        breg[0][0][0] = breg[0][0][1] = make_uint4(0u, 0u, 0u, 0u);
    }

    // Deferred tile stores (node-texture builds without direct output).  The memory counter of gfx950 counts loads AND stores, in order:
    // the wait at the top of a chunk — for the chunk's blocks, requested a whole chunk ago — also waited for the stores of the tile row that
    // had just been finished, i.e. for a full write round trip per tile row (the build without stores ran 3.8 us shorter).  A finished tile
    // row's pixels therefore stay in registers (8) until that wait is over and leave right behind it: by the next wait they are a chunk old.
    constexpr bool DEFER = (RG || FX) && !DIRECT && !SA && NKS_T != 0 && (SMR_WAVE_DEFER_STORES != 0);  // (the generic builds sit at the register limit: immediate stores there)
    u32 pend_px[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    int pend_vt = -1;  // (uniform) the tile row waiting to be stored, -1 = none
    auto store_rows = [&](int vt_s, const u32 (&p)[2][4]) {
#pragma unroll
        for (int i = 0; i < W_NTI; i++) {
            if (klo[i] == 0xff) continue;
            const int y = 16 * vt_s + l16, x = tx0 + 16 * i + 4 * lq;  // lane holds columns x .. x + 3 of output row y
            if (y < d_h && x < d_w && (!(SMR_WAVE_ABL & 16) || p[i][0] == 0x12345678u)) {  // (16: profiling, all work but no store traffic)
                u8 *op = d_ptr + dev_mad24((u32)y, d_pitch, (u32)x * 4u);  // (a tile is far below 4 GiB)
                if (x + 3 < d_w) {
                    *(uint4 *)op = make_uint4(p[i][0], p[i][1], p[i][2], p[i][3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (x + k < d_w) ((u32 *)op)[k] = p[i][k];
                }
            }
        }
    };
    int vt = vt0;
    int2 vm = J.v_meta[vt0];
    int2 vm_next = J.v_meta[min(vt0 + 1, vt1)];  // (read a tile ahead: a scalar load the loop never waits for)
    const int c_first = vm.x, c_last = J.v_meta[vt1].y;
    if (!SA) fetch_bv(vt0);
    if (RG) {
#pragma unroll
        for (int j = 0; j < RG_N; j++)
            if (j < NKS) rg[j] = rg_load(c_first, j);
    }
    issue(c_first);
    fx_issue(c_first);
    finish_prologue();
    dev_wait_vmcnt0();
    land();
    dev_wave_lds_sync();
    if (c_first < c_last) issue(c_first + 1);
    int slot = w_posmod(c_first, 2 * KV);  // ring slot of the chunk at hand

    W_MARK(0);
    for (int c = c_first; c <= c_last; c++) {
        // (node-texture builds: the blocks of chunk c + 1 are requested while chunk c is converted, i.e. BEFORE this chunk's tile rows read their
        //  pass-2 weights, and the memory counter is in order.  The weights were requested a tile row ago; a wait for them must not include the
        //  blocks requested a moment ago — a full memory latency per tile row, 37 % of a wave's time when it did (profiles/r04_wave_timing.txt).
        //  Three things keep the waits where they cost nothing: the block loads are unconditional (the last chunk re-requests itself: the compiler can
        //  count the loads in flight), the first tile row of a chunk is its own copy of the code (below), and everything outstanding is waited for
        //  here, at the top of the chunk, where it is old — this chunk's blocks, the next tile row's weights.)
        if ((RG || FX) && SMR_WAVE_RG_EARLY_WAIT) dev_wait_vmcnt0();
        if (FX) {  // the chunk's node texels: converted, into LDS; the next chunk's blocks requested
            dev_wave_lds_sync();  // (every lane has read its groups of the previous chunk)
            fx_convert(c);
            dev_wave_lds_sync();
            fx_issue(min(c + 1, c_last));  // (unconditional: the compiler can count the loads in flight)
        }
        if (DEFER && pend_vt >= 0) {  // (uniform) the tile row finished in the previous chunk
            store_rows(pend_vt, pend_px);
            pend_vt = -1;
        }
        // ---- conversion + pass 1 of chunk c: k-step by k-step, straight into the accumulators of the tiles it feeds
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < W_NTI; i++)
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) acc[i][ch] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (NKS_T && SMR_WAVE_PIPE) {
            // ---- class build: software pipeline over the k-steps.  Step j: the LDS reads of block j + 1 and of the weights of step
            //      j - 1 go out, block j is converted (vector ALU + table gathers), then the MFMAs of step j - 1 are issued — their
            //      operands arrived long ago, and the tail of block j's gathers lands behind them.
            struct Raw { u32 yy, u0, u1, u2, u3, v0, v1, v2, v3; };
            auto read_raw = [&](int j, Raw &r) {
                if (FX) { r.yy = fx_rd[12 * j]; r.u0 = fx_rd[12 * j + 1]; r.u1 = fx_rd[12 * j + 2]; return; }  // (the group's R, G, B dwords)
                if (RG) return;
                r.yy = yrow[4 * j];
                r.u0 = urow[2 * j]; r.u1 = urow[2 * j + 1]; r.u2 = urow[cs + 2 * j]; r.u3 = urow[cs + 2 * j + 1];
                r.v0 = vrow[2 * j]; r.v1 = vrow[2 * j + 1]; r.v2 = vrow[cs + 2 * j]; r.v3 = vrow[cs + 2 * j + 1];
            };
            auto read_b = [&](int j, uint4 (&bq)[2][2]) {
#pragma unroll
                for (int i = 0; i < W_NTI; i++) {
                    if (B_REGS) {  // (a register rename)
                        bq[i][0] = breg[B_REGS ? i : 0][B_REGS ? j : 0][0];
                        bq[i][1] = breg[B_REGS ? i : 0][B_REGS ? j : 0][1];
                        continue;
                    }
                    bq[i][0] = Bs[((i * NKS + j) * 2) * 64 + lane];
                    bq[i][1] = Bs[((i * NKS + j) * 2 + 1) * 64 + lane];
                }
            };
            auto convert = [&](int j, const Raw &r, uint4 (&a)[4]) {
                if (FX) { decode_rgb12(r.yy, r.u0, r.u1, a); return; }
                if (RG) {
                    convert_rgba(j, a);
                    rg[RG ? (j < RG_N ? j : 0) : 0] = rg_load(min(c + 1, c_last), j);  // (unconditional: the compiler can then count the loads in flight — see the chunk loop)
                    return;
                }
                const u32 ua = dev_alignbyte(r.u1, r.u0, shb), ub = dev_alignbyte(r.u3, r.u2, shb);
                const u32 va = dev_alignbyte(r.v1, r.v0, shb), vb = dev_alignbyte(r.v3, r.v2, shb);
                m_convert_px<(SMR_WAVE_ABL & 1) != 0>(K, r.yy, ua, ub, va, vb, w13, w31, a);  // (ABL & 1: profiling, no table gathers)
            };
            auto mfmas = [&](int j, const uint4 (&a)[4], const uint4 (&bq)[2][2]) {
                if (SMR_WAVE_ABL & 2) {  // profiling: no pass-1 MFMAs (the operands stay alive)
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        acc[0][ch][0] += __uint_as_float(a[ch].x ^ a[ch].y ^ bq[0][0].x); acc[W_NTI - 1][ch][1] += __uint_as_float(a[ch].z ^ a[ch].w ^ bq[W_NTI - 1][1].y);
                    }
                    return;
                }
#if SMR_WAVE_SETPRIO
                __builtin_amdgcn_s_setprio(SMR_WAVE_SETPRIO);
#endif
#pragma unroll
                for (int i = 0; i < W_NTI; i++) {
                    if (K01 && ((i == 0 && j == 3) || (i == 1 && j == 0))) continue;
                    if (NKS_T > 4 && (j < klo[i] || j > khi[i])) continue;  // (uniform: wide windows — most (tile, k-step) fragments are zero)
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++)
                        acc[i][ch] = dev_mfma_16x16x32_f16(__builtin_bit_cast(f16x8, a[ch]), __builtin_bit_cast(f16x8, bq[i][0]), acc[i][ch]);
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++)
                        acc[i][ch] = dev_mfma_16x16x32_f16(__builtin_bit_cast(f16x8, a[ch]), __builtin_bit_cast(f16x8, bq[i][1]), acc[i][ch]);
                }
#if SMR_WAVE_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            };
            Raw cur, nxt;
            uint4 a[4], a_prev[4], bq[2][2];
            read_raw(0, cur);
#pragma unroll
            for (int j = 0; j < NKS_T; j++) {
                if (FIXED_NKS || j < NKS) {  // (uniform)
                    if (j + 1 < NKS_T && (FIXED_NKS || j + 1 < NKS)) read_raw(j + 1, nxt);
                    if (j > 0) read_b(j - 1, bq);
                    if (SMR_WAVE_PIPE_FENCE) dev_sched_barrier();
                    convert(j, cur, a);
                    if (SMR_WAVE_PIPE_FENCE) dev_sched_barrier();
                    if (j > 0) mfmas(j - 1, a_prev, bq);
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) a_prev[ch] = a[ch];
                    cur = nxt;
                }
            }
            const int j_last = FIXED_NKS ? NKS_T - 1 : NKS - 1;
            read_b(j_last, bq);
            if (SMR_WAVE_PIPE_FENCE) dev_sched_barrier();
            W_MARK(1);
            // ---- the next chunk's footprint (issued a whole conversion ago; the one after it goes out now) while the weights arrive
            if (c < c_last) {
                if (SMR_WAVE_EXPLICIT_WAIT) dev_wait_vmcnt0();
                dev_wave_lds_sync();  // (every lane has read its blocks of chunk c)
                land();
                dev_wave_lds_sync();
            }
            W_MARK(2);
            if (SMR_WAVE_PIPE_FENCE) dev_sched_barrier();
            mfmas(j_last, a_prev, bq);
        } else {
#pragma unroll
            for (int j = 0; j < NKS_N; j++) {
                if (NKS_T || j < nks) {  // (uniform; the class build converts its NKS_T k-steps unconditionally: texels past a window meet zero weights)
                    uint4 bq[2][2];
                    if (SMR_WAVE_PREFETCH_B) {
    #pragma unroll
                        for (int i = 0; i < W_NTI; i++) {
                            bq[i][0] = Bs[((i * NKS + j) * 2) * 64 + lane];
                            bq[i][1] = Bs[((i * NKS + j) * 2 + 1) * 64 + lane];
                        }
                    }
                    u32 yy = 0u, ua = 0u, ub = 0u, va = 0u, vb = 0u;
                    if (!RG) {
                        yy = yrow[4 * j];
                        ua = dev_alignbyte(urow[2 * j + 1], urow[2 * j], shb); ub = dev_alignbyte(urow[cs + 2 * j + 1], urow[cs + 2 * j], shb);
                        va = dev_alignbyte(vrow[2 * j + 1], vrow[2 * j], shb); vb = dev_alignbyte(vrow[cs + 2 * j + 1], vrow[cs + 2 * j], shb);
                    }
                    uint4 a[4];
                    if (RG) {
                        convert_rgba(j, a);
                        rg[RG ? (j < RG_N ? j : 0) : 0] = rg_load(min(c + 1, c_last), j);  // (unconditional: the compiler can then count the loads in flight — see the chunk loop)
                    } else if (SMR_WAVE_ABL & 4) {
                        a[0] = make_uint4(yy, ua, ub, va); a[1] = make_uint4(vb, yy, ua, ub); a[2] = make_uint4(va, vb, yy, ua);
                    } else {
                        m_convert_px<(SMR_WAVE_ABL & 1) != 0>(K, yy, ua, ub, va, vb, w13, w31, a);
                    }
                    if (SMR_WAVE_ABL & 2) {
    #pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            acc[0][ch][0] += __uint_as_float(a[ch].x ^ a[ch].y); acc[1][ch][0] += __uint_as_float(a[ch].z ^ a[ch].w);
                        }
                        continue;
                    }
    #pragma unroll
                    for (int i = 0; i < W_NTI; i++) {
                        if (K01 && NKS_T && ((i == 0 && j == 3) || (i == 1 && j == 0))) continue;
                        if (NKS_T || (j >= klo[i] && j <= khi[i])) {  // (uniform)
                            if (!SMR_WAVE_PREFETCH_B) {
                                bq[i][0] = Bs[((i * NKS + j) * 2) * 64 + lane];
                                bq[i][1] = Bs[((i * NKS + j) * 2 + 1) * 64 + lane];
                            }
    #pragma unroll
                            for (int ch = 0; ch < NCH; ch++)
                                acc[i][ch] = dev_mfma_16x16x32_f16(__builtin_bit_cast(f16x8, a[ch]), __builtin_bit_cast(f16x8, bq[i][0]), acc[i][ch]);
    #pragma unroll
                            for (int ch = 0; ch < NCH; ch++)
                                acc[i][ch] = dev_mfma_16x16x32_f16(__builtin_bit_cast(f16x8, a[ch]), __builtin_bit_cast(f16x8, bq[i][1]), acc[i][ch]);
                        }
                    }
                }
            }
            // ---- the next chunk's footprint: issued a whole conversion ago; the one after it goes out now
            if (c < c_last && !(SMR_WAVE_ABL & 32)) {
                if (SMR_WAVE_EXPLICIT_WAIT) dev_wait_vmcnt0();
                dev_wave_lds_sync();  // (every lane has read its blocks of chunk c)
                land();
                dev_wave_lds_sync();
            }
        }
        if (SA) {
            // ---- single-axis plan: the chunk's rows are output rows (lane holds rows 4 lq .. + 3 of output column l16 of either tile)
            const int y_lo = 16 * vt0, y_hi = min(16 * vt1 + 15, d_h - 1), perp = J.perp;
#pragma unroll
            for (int i = 0; i < W_NTI; i++) {
                if (klo[i] == 0xff) continue;
                const int x = tx0 + 16 * i + l16;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int y = 16 * c - 1 + 4 * lq + k - perp;
                    const u32 px = AL ? (w_encode_px(acc[i][0][k], acc[i][1][k], acc[i][2][k], s_thr) & 0xffffffu) | (unorm8(acc[i][AL ? 3 : 0][k]) << 24)
                                      : w_encode_px(acc[i][0][k], acc[i][1][k], acc[i][2][k], s_thr);
                    if (y >= y_lo && y <= y_hi && x < d_w) *(u32 *)(d_ptr + dev_mad24((u32)y, d_pitch, (u32)x * 4u)) = px;
                }
            }
            while (vt <= vt1 && vm.y == c) {
                vt++;
                if (vt <= vt1) { vm = vm_next; vm_next = J.v_meta[min(vt + 1, vt1)]; }
            }
            if (c + 1 < c_last) issue(c + 2);
            continue;
        }
        // ---- the chunk's 16 rows of H, rounded to f16 (resampler.rs:25-28), into ring slot c mod 2 KV: lane holds rows 4 lq .. + 3 of
        //      output column l16 of either tile
        u32 h16[2][4][2];
#pragma unroll
        for (int i = 0; i < W_NTI; i++)
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                const __half2 h0 = __floats2half2_rn(acc[i][ch][0], acc[i][ch][1]), h1 = __floats2half2_rn(acc[i][ch][2], acc[i][ch][3]);
                const u32 lo = *(const u32 *)&h0, hi = *(const u32 *)&h1;
                h16[i][ch][0] = lo;
                h16[i][ch][1] = hi;
            }
        // (slot is uniform: one taken branch with twelve moves, not a select per slot and value — the empty asm keeps the compiler from
        //  turning the branches back into selects)
#pragma unroll
        for (int s = 0; s < 2 * KV_N; s++)
            if (s == slot) {
#pragma unroll
                for (int i = 0; i < W_NTI; i++)
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) {
                        ring[i][ch][2 * s] = h16[i][ch][0];
                        ring[i][ch][2 * s + 1] = h16[i][ch][1];
                    }
#ifndef SMR_EMU
                asm volatile("" ::: "memory");
#endif
            }
        slot = slot + 1 == 2 * KV ? 0 : slot + 1;
        W_MARK(3);
        // ---- pass 2 + encode + store of every output tile whose window ends with chunk c.
        //      The first tile row of a chunk and any further one (up-scaling: several tile rows per chunk) are two copies of the code: the
        //      weights of the first were requested a tile row ago and need no wait (node-texture builds: the wait at the top of the chunk;
        //      fused builds: the chunk's own loads go out after this loop), those of a further row were requested a moment ago.  As ONE loop the
        //      compiler has to wait at its header for the youngest loads of either path — behind the next chunk's blocks (see above).
        auto tile_row = [&](auto first_of_chunk) {
#ifndef SMR_EMU
            if (decltype(first_of_chunk)::value) asm volatile("; first tile row of the chunk");  // (keeps the two copies from being merged back)
#endif
            // pass 2 of BOTH tiles first, then the next tile row's weights are requested (the registers are free), then both encodes with all
            // their lookups in flight together, then the stores.  (Measured neutral on configs[2] against one tile after the other — 34.5 vs
            // 33.8 us: the kernel's floor is its memory traffic, profiles/r04_wave_ablation.txt — and kept for the earlier weight request.)
            f32x4 o[2][4];
            u32 px[2][4];
#pragma unroll
            for (int i = 0; i < W_NTI; i++) {
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) o[i][ch] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (klo[i] == 0xff || (SMR_WAVE_ABL & 8)) continue;  // (uniform: no second tile in the last pair of an odd tile count)
#pragma unroll
                for (int p = 0; p < KV_N; p++) {
                    if (p < KV) {
#pragma unroll
                        for (int ch = 0; ch < NCH; ch++)
                            o[i][ch] = dev_mfma_16x16x32_f16(__builtin_bit_cast(f16x8, ring_quad(i, ch, p)), __builtin_bit_cast(f16x8, bvh[p]), o[i][ch]);
#pragma unroll
                        for (int ch = 0; ch < NCH; ch++)
                            o[i][ch] = dev_mfma_16x16x32_f16(__builtin_bit_cast(f16x8, ring_quad(i, ch, p)), __builtin_bit_cast(f16x8, bvl[p]), o[i][ch]);
                    }
                }
            }
            W_MARK(4);
            const int vt_now = vt;
            u32 cls_now[2] = {cls_next[0], cls_next[1]};
            // (the weights are consumed: the next tile row's go out now, under the encode)
            vt++;
            if (vt <= vt1) {
                vm = vm_next;
                vm_next = J.v_meta[min(vt + 1, vt1)];
                if (!(SMR_WAVE_ABL & 512)) fetch_bv(vt);  // (512: profiling, every tile with the first tile's weights)
            }
#pragma unroll
            for (int i = 0; i < W_NTI; i++) {
                if (klo[i] == 0xff) continue;
                if (SMR_WAVE_ABL & 8) {
#pragma unroll
                    for (int k = 0; k < 4; k++) px[i][k] = ring[i][k & 1][k] ^ ring[i][2][4 * KV_N - 1 - k];
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    px[i][k] = AL ? (w_encode_px(o[i][0][k], o[i][1][k], o[i][2][k], s_thr) & 0xffffffu) | (unorm8(o[i][AL ? 3 : 0][k]) << 24)
                                  : w_encode_px(o[i][0][k], o[i][1][k], o[i][2][k], s_thr);
            }
#ifndef SMR_EMU
#pragma unroll
            for (int i = 0; i < W_NTI; i++)
#pragma unroll
                for (int k = 0; k < 4; k++) asm volatile("" : "+v"(px[i][k]));  // (all the lookups in flight together: not sunk into the store branches)
#endif
            W_MARK(6);
            if (DEFER) {
                if (pend_vt >= 0) store_rows(pend_vt, pend_px);  // (a second tile row of the same chunk: the older one leaves now)
#pragma unroll
                for (int i = 0; i < W_NTI; i++)
#pragma unroll
                    for (int k = 0; k < 4; k++) pend_px[i][k] = px[i][k];
                pend_vt = vt_now;
                return;
            }
#pragma unroll
            for (int i = 0; i < W_NTI; i++) {
                if (klo[i] == 0xff) continue;
                // lane holds columns x .. x + 3 of output row y
                const int y = 16 * vt_now + l16, x = tx0 + 16 * i + 4 * lq;
                bool direct = false;
                if (dj) {
                    // (m_direct_yuv: the arithmetic of k_compose_output's copy tiles on the bytes above; every lane takes part in its
                    //  lane swaps, the lanes of a direct tile store)
                    direct = cls_now[i] == (u32)d_layer;
                    const bool odd = (l16 & 1) != 0;
                    u32 mine, other;
#if SMR_DIRECT_ABL & 1  // (profiling builds: no conversion arithmetic)
                    const u32 yq = px[i][0] ^ px[i][1];
                    mine = px[i][2]; other = px[i][3];
#else
                    const u32 yq = m_direct_yuv(px[i], odd, &mine, &other);
#endif
                    if (direct && (!(SMR_DIRECT_ABL & 2) || yq == 0x12345678u)) m_direct_store(Dg, d_ox + x, d_oy + y, odd, yq, mine, other);  // (2: profiling, no direct stores)
                }
                if (!direct && y < d_h && x < d_w && (!(SMR_WAVE_ABL & 16) || px[i][0] == 0x12345678u)) {  // (16: profiling, all work but no store traffic)
                    u8 *op = d_ptr + dev_mad24((u32)y, d_pitch, (u32)x * 4u);  // (a tile is far below 4 GiB)
                    if (x + 3 < d_w) {
                        *(uint4 *)op = make_uint4(px[i][0], px[i][1], px[i][2], px[i][3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (x + k < d_w) ((u32 *)op)[k] = px[i][k];
                    }
                }
            }
        };
        if (vt <= vt1 && vm.y == c) {
            tile_row(std::true_type{});
            while (vt <= vt1 && vm.y == c) tile_row(std::false_type{});
        }
        W_MARK(7);
        // ---- the footprint of chunk c + 2 goes out now: it has the whole next conversion to arrive, and the loop's waits for the pass-2
        //      weights above never stand behind it (the memory counter is in order)
        if (c + 1 < c_last && !(SMR_WAVE_ABL & 32)) issue(c + 2);
        W_MARK(5);
    }
    if (DEFER && pend_vt >= 0) store_rows(pend_vt, pend_px);
#if SMR_WAVE_TIMING
    if (timing && lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(dbg + i, tph[i]);
        atomicAdd(dbg + 8, (unsigned long long)(c_last - c_first + 1));
        atomicAdd(dbg + 9, (unsigned long long)(vt1 - vt0 + 1));
        atomicAdd(dbg + 10, 1ull);
    }
#endif
#undef W_MARK
}

template <int NKS_T, int KV_T, int FL>
__global__ __launch_bounds__(W_THREADS, (NKS_T == 4 && (FL & 8192) && !(FL & (2048 | 16384 | 32768 | 65536))) ? SMR_WAVE_MIN_WAVES_RG : SMR_WAVE_MIN_WAVES) void k_ingest_wave(const WArgs args, const float *__restrict__ tables, const u32 *__restrict__ lut) {
#ifdef SMR_EMU
    u8 *smem = emu_smem;
#else
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
#endif
    const int tid = threadIdx.x;
    const int wave = dev_readfirstlane(tid >> 6);
#if defined(SMR_PRIO_WAVE) && !defined(SMR_EMU)
    __builtin_amdgcn_s_setprio(SMR_PRIO_WAVE);
#endif
#if SMR_WAVE_STAGGER && !defined(SMR_EMU)
    if (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 4) & 1u) __builtin_amdgcn_s_sleep(SMR_WAVE_STAGGER);  // HW_ID.wave_id (this wave's slot in its SIMD)
#endif
    // XCD-aware order: workgroup ids that share an XCD are neighbours in unit space (piece-group major, pair fastest: neighbouring
    // pairs read the same source lines at the same time)
    const int per_xcd = ((int)gridDim.x + 7) >> 3;
    const int v = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    const int total = args.wg_prefix[args.n_jobs];
    if (v >= total) return;
    int j = 0;
    while (j + 1 < args.n_jobs && args.wg_prefix[j + 1] <= v) j++;
    const WJob &J = args.jobs[j];
    const int local = v - args.wg_prefix[j];
    const int group = local / J.n_pairs, pair = local - group * J.n_pairs;
    // ---- prologue: the decode LUT ((hi | lo << 16), clamp folded in), the encode tables and the pair's pass-1 band go to LDS.  Every
    //      load is issued before the first store, and the piece's own first loads (its weights, its first chunk) go out in between:
    //      the workgroup pays one memory latency, not four.
    const int NKS = (NKS_T > 0 && NKS_T <= 4) ? NKS_T : J.NKS;  // (as in wave_piece)
    const uint4 *src = J.h_frag + (size_t)pair * 2 * J.NKS * 2 * 64;
    uint4 *Bs = (uint4 *)(smem + W_OFF_B);
    constexpr int NL = (M_LUT_ENTRIES + W_THREADS - 1) / W_THREADS, NT = ((SMR_WAVE_ENC1 ? SMR_ENC_ENTRIES : SMR_TABLE_FLOATS - 256) + W_THREADS - 1) / W_THREADS;
    constexpr bool B_REGS = NKS_T > 0 && NKS_T <= 4 && SMR_WAVE_PIPE && SMR_WAVE_B_REGS;  // (as in wave_piece: no LDS band)
    constexpr int NB = NKS_T && !B_REGS ? (2 * NKS_T * 2 * 64 + W_THREADS - 1) / W_THREADS : 1;
    u32 r_lut[NL];
    u32 r_thr[NT];
    uint4 r_b[NB];
#pragma unroll
    for (int k = 0; k < NL; k++) r_lut[k] = lut[min(max(tid + k * W_THREADS - 256, 0), 255)];
#pragma unroll
    for (int k = 0; k < NT; k++)
        r_thr[k] = SMR_WAVE_ENC1 ? lut[256 + min(tid + k * W_THREADS, SMR_ENC_ENTRIES - 1)] : __float_as_uint(tables[256 + min(tid + k * W_THREADS, SMR_TABLE_FLOATS - 257)]);
    if (NKS_T && !B_REGS) {
#pragma unroll
        for (int k = 0; k < NB; k++) {  // (a job of the class with a narrower window: its band in the class's layout, zero beyond)
            const int i = tid + k * W_THREADS, t = i / (NKS * 128), r = i - t * (NKS * 128);
            r_b[k] = make_uint4(0u, 0u, 0u, 0u);
            if (t < 2 && r < J.NKS * 128) r_b[k] = src[t * J.NKS * 128 + r];
        }
    }
    auto finish_prologue = [&]() {
#pragma unroll
        for (int k = 0; k < NL; k++)
            if (tid + k * W_THREADS < M_LUT_ENTRIES) ((u32 *)smem)[tid + k * W_THREADS] = r_lut[k];
#pragma unroll
        for (int k = 0; k < NT; k++)
            if (tid + k * W_THREADS < (SMR_WAVE_ENC1 ? SMR_ENC_ENTRIES : SMR_TABLE_FLOATS - 256)) ((u32 *)(smem + W_OFF_THR))[tid + k * W_THREADS] = r_thr[k];
        if (B_REGS) {
        } else if (NKS_T) {
#pragma unroll
            for (int k = 0; k < NB; k++)
                if (tid + k * W_THREADS < 2 * NKS * 2 * 64) Bs[tid + k * W_THREADS] = r_b[k];
        } else {
            for (int i = tid; i < 2 * NKS * 2 * 64; i += W_THREADS) Bs[i] = src[i];
        }
        if (FL & 262144) {  // plane-source builds: the exact converter's tables behind the band — y' of a limited-range luma byte, byte / 255 (smr_convert_420.h)
            float *yl = (float *)(smem + W_OFF_B + args.b_bytes), *nl = yl + 256;
            for (int i = tid; i < 256; i += W_THREADS) {
                yl[i] = cv420_luma_of_byte((u32)i, false);
                nl[i] = unorm_of_byte((u32)i);
            }
        }
        if ((FL & 8192) && !(FL & 16384) && (FL & 65536)) {  // alpha builds: a / 255 (correctly rounded, as unorm_of_byte) as (hi | lo << 16)
            for (int i = tid; i < 256; i += W_THREADS) {
                const float v = unorm_of_byte((u32)i);
                const __half2 h = __floats2half2_rn(v, 0.0f);                       // hi
                const __half2 l = __floats2half2_rn(v - __half22float2(h).x, 0.0f);  // lo = v - hi
                ((u32 *)(smem + W_OFF_B + args.b_bytes))[i] = ((*(const u32 *)&h) & 0xffffu) | ((*(const u32 *)&l) << 16);
            }
        }
        __syncthreads();
    };
    if (SMR_WAVE_ABL & 1024) {  // profiling: launch + prologue only
        finish_prologue();
        if (tid == 0 && Bs[5].x == 0x12345678u) J.dst.ptr[0] = 1;
        return;
    }
    const int piece = group * W_WAVES + wave;
    const int vt0 = (int)(((long long)piece * J.n_vtiles) / J.pieces), vt1 = (int)(((long long)(piece + 1) * J.n_vtiles) / J.pieces) - 1;
    if (vt0 <= vt1)
        wave_piece<NKS_T, KV_T, FL>(J, args.direct, pair, vt0, vt1, smem, (u32)W_OFF_B, (u32)(W_OFF_B + args.b_bytes + ((FL & 262144) ? W_FX_TABLE_BYTES : 0) + wave * args.raw_bytes), args.dbg,
                                    finish_prologue, args.b_bytes);
    else
        finish_prologue();
}

#endif  // __HIPCC__

}  // namespace

#ifndef SMR_EMU
namespace {

// ------------------------------------------------------------------ host side: shared helpers
// the cached band of (scale, offset, n_dst, n_src, axis), if any: a frame of a scene at rest finds all of its bands here and
// never walks the tile geometry on the host
smr_ctx::MfmaTable *find_mfma_table(smr_ctx *ctx, float scale, float offset, int n_dst, int n_src, int axis) {
    for (auto &t : ctx->mfma_tables)
        if (t.dev && t.n_dst == n_dst && t.n_src == n_src && t.axis == axis && t.scale == scale && t.offset == offset) return &t;
    return nullptr;
}
bool mfma_plane_ok(const SurfView &p, u32 bytes) { return (p.pitch % 4) == 0 && (((uintptr_t)p.ptr) % 4) == 0 && p.pitch >= ((bytes + 3u) & ~3u); }
// the fused colour conversion (k_ingest_wave reading Y'CbCr planes) is opt-in: its conversion is within one code of the WGSL pass, not equal
// to it, and a linear-light filter can amplify a flipped code (include/smr.h) — the default converts exactly, then resamples the node
inline bool fused_conversion(const smr_ctx *ctx) { return ctx->ingest_impl == SMR_INGEST_LAB_FUSED; }

// ------------------------------------------------------------------ vertical-first plans: the same kernel on the transposed problem
// The reference filters the axis with the stronger shrink first (resampler.rs:123-145); for aspect-preserving fits that order hangs
// on the rounding of the tile size, so a tile that resizes flips between the two from frame to frame.  The kernel below filters
// horizontally first.  Filtering an image vertically first is filtering its transpose horizontally first — every tap loop runs along
// one axis — so a vertical-first plan is run as: transpose the node (or the planes), the horizontal-first kernel into a transposed
// tile, transpose the tile back.  Same quantisation points in the same places as the reference's order.
template <typename T>
__global__ __launch_bounds__(256) void k_transpose(const u8 *__restrict__ src, u32 spitch, int w, int h, u8 *__restrict__ dst, u32 dpitch) {
    __shared__ T tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (bx + tx < w && by + r < h) tile[r][tx] = *(const T *)(src + (size_t)(by + r) * spitch + (size_t)(bx + tx) * sizeof(T));
    __syncthreads();
    for (int r = ty; r < 32; r += 8)  // dst row = src column bx + r, dst column = src row by + tx
        if (by + tx < h && bx + r < w) *(T *)(dst + (size_t)(bx + r) * dpitch + (size_t)(by + tx) * sizeof(T)) = tile[tx][r];
}

template <typename T>
int launch_transpose(smr_ctx *ctx, const smr_surface *src, smr_surface *dst) {  // dst is src->h x src->w
    hipLaunchKernelGGL(k_transpose<T>, dim3((src->w + 31) / 32, (src->h + 31) / 32), dim3(256), 0, ctx->stream, (const u8 *)src->ptr, (u32)src->pitch,
                       (int)src->w, (int)src->h, (u8 *)dst->ptr, (u32)dst->pitch);
    SMR_HIP(ctx, hipGetLastError());
    return SMR_OK;
}

struct MTransposeBack {
    smr_surface *tile_t;  // what the kernel writes (tile->h x tile->w)
    smr_surface *tile;    // what the caller asked for
};

// ------------------------------------------------------------------ host side: band cache
struct WaveBand {
    const void *meta;
    const uint4 *frag;
    int K;        // k-steps per tile band (axis 2) / per window (axis 3)
    int nks;      // axis 2: k-steps of the widest pair window
    bool k01;     // axis 2: see wave_band_geometry
    int n_units;  // pairs / tiles
};

int get_wave_band(smr_ctx *ctx, float scale, float offset, int n_dst, int n_src, int axis, WaveBand *out) {
    const int n_tiles = (n_dst + 15) / 16;
    const int n_units = axis == 2 ? (n_tiles + 1) / 2 : n_tiles;  // (axis 4: one tile per unit)
    smr_ctx::MfmaTable *hit = find_mfma_table(ctx, scale, offset, n_dst, n_src, axis), *victim = nullptr;
    if (!hit) {
        int K, nks;
        bool k01 = false;
        wave_band_geometry(scale, offset, n_dst, n_src, axis, &K, &nks, &k01);
        for (auto &t : ctx->mfma_tables)  // never evict a table the current call already handed to a job that is not launched yet
            if (t.last_call != ctx->weight_call && (!victim || t.last_use < victim->last_use)) victim = &t;
        if (!victim || ctx->mfma_tables.size() < 64) {
            ctx->mfma_tables.emplace_back();
            victim = &ctx->mfma_tables.back();
        }
        const size_t meta_bytes = ((size_t)n_units * (!w_axis_vertical(axis) ? sizeof(int4) : sizeof(int2)) + 15) & ~(size_t)15;
        const size_t frags = !w_axis_vertical(axis) ? (size_t)n_units * 2 * K * 2 : (size_t)n_units * K * 2;  // (axis 2 / 4: K = k-steps of the widest window)
        const size_t need = meta_bytes + frags * 64 * sizeof(uint4);
        for (size_t i = ctx->pending_bands.size(); i-- > 0;)
            if (victim->dev && ctx->pending_bands[i].meta == victim->dev) ctx->pending_bands.erase(ctx->pending_bands.begin() + (long)i);
        if (victim->bytes < need) {
            if (victim->dev) {
                SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a queued kernel may still read it
                (void)hipFree(victim->dev);
                victim->dev = nullptr;
                victim->bytes = 0;
            }
            const size_t want = (need + 4095) & ~(size_t)4095;
            SMR_HIP(ctx, hipMalloc(&victim->dev, want));
            victim->bytes = want;
        }
        victim->scale = scale; victim->offset = offset; victim->n_dst = n_dst; victim->n_src = n_src; victim->axis = axis;
        victim->K = K; victim->max_span = k01 ? 1 : 0; victim->meta_bytes = meta_bytes; victim->ngm = nks;
        smr_ctx::PendingBand pb;
        pb.scale = scale; pb.offset = offset; pb.taps = w_taps(scale); pb.n_dst = n_dst; pb.n_src = n_src; pb.axis = axis; pb.K = K;
        pb.n_tiles = n_units; pb.meta = victim->dev; pb.frag = (u8 *)victim->dev + meta_bytes;
        ctx->pending_bands.push_back(pb);
        hit = victim;
    }
    hit->last_use = ++ctx->weight_clock;
    hit->last_call = ctx->weight_call;
    out->meta = hit->dev;
    out->frag = (const uint4 *)((const u8 *)hit->dev + hit->meta_bytes);
    out->K = hit->K;
    out->nks = hit->ngm;
    out->k01 = hit->max_span != 0;
    out->n_units = n_units;
    return SMR_OK;
}

// builds the bands (axis 2 / 3 / 4) the call is missing, one launch for all of them
int flush_wave_builds(smr_ctx *ctx) {
    std::vector<smr_ctx::PendingBand> rest, mine;
    for (const auto &p : ctx->pending_bands) (p.axis >= 2 ? mine : rest).push_back(p);
    size_t i = 0;
    while (i < mine.size()) {
        WWBatch args;
        memset(&args, 0, sizeof(args));
        int units = 0;
        for (; i < mine.size() && args.n < MAX_WWBUILDS; i++) {
            const smr_ctx::PendingBand &p = mine[i];
            WWBuild &b = args.b[args.n++];
            b.scale = p.scale; b.offset = p.offset; b.taps = p.taps; b.n_dst = p.n_dst; b.n_src = p.n_src; b.axis = p.axis; b.K = p.K;
            b.unit0 = units; b.meta = p.meta; b.frag = (uint4 *)p.frag;
            units += p.n_tiles;
        }
        hipLaunchKernelGGL(k_build_wave_weights, dim3((unsigned)units), dim3(64), 0, ctx->stream, args);
        SMR_HIP(ctx, hipGetLastError());
    }
    ctx->pending_bands = rest;
    return SMR_OK;
}

// ------------------------------------------------------------------ host side: jobs
// What k_ingest_wave covers: planar 4:2:0 (limited or full range) or NV12 with even luma size and dword-aligned planes, a separable
// two-pass plan with the horizontal pass first and no box pre-reduction (vertical-first plans come back on the transposed frame), 16-byte aligned tile rows, k-step counts within the kernel's limits.
bool can_fuse_wave(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile) {
    if (!fused_conversion(ctx)) return false;
    const bool nv12 = f && f->format == SMR_FRAME_NV12;
    if (!f || !f->planes[0] || !f->planes[1] || (!nv12 && !f->planes[2])) return false;
    if (f->format != SMR_FRAME_PLANAR_YUV420 && f->format != SMR_FRAME_PLANAR_YUVJ420 && !nv12) return false;
    if (f->width % 2 || f->height % 2 || f->width < 8 || f->height < 2) return false;
    if (!(plan.kind == 2 && plan.levels[0] == 0 && plan.levels[1] == 0 && plan.axis[0] == 0 && plan.axis[1] == 1)) return false;
    if (!mfma_plane_ok(view_of(f->planes[0]), f->width)) return false;
    if (nv12) {  // (the last staged dword pair may start up to 3 texels before the row's end: 8 bytes must be readable there)
        const SurfView uv = view_of(f->planes[1]);
        if ((uv.pitch % 4) || (((uintptr_t)uv.ptr) % 4) || uv.pitch < ((f->width + 7u) & ~7u)) return false;
    } else if (!mfma_plane_ok(view_of(f->planes[1]), f->width / 2) || !mfma_plane_ok(view_of(f->planes[2]), f->width / 2)) {
        return false;
    }
    if (((uintptr_t)tile->ptr % 16) || (tile->pitch % 16)) return false;
    int NKS, KV, unused;
    if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[0], plan.offset[0], (int)tile->w, (int)f->width, 2)) NKS = t->K;
    else wave_band_geometry(plan.scale[0], plan.offset[0], (int)tile->w, (int)f->width, 2, &NKS, &unused);
    if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[1], plan.offset[1], (int)tile->h, (int)f->height, 3)) KV = t->K;
    else wave_band_geometry(plan.scale[1], plan.offset[1], (int)tile->h, (int)f->height, 3, &KV, &unused);
    return NKS <= W_NKS_MAX && KV <= W_KV_MAX;
}

int make_wave_job(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile, WJob *out) {
    WaveBand bh, bv;
    int rc = get_wave_band(ctx, plan.scale[0], plan.offset[0], (int)tile->w, (int)f->width, 2, &bh);
    if (rc != SMR_OK) return rc;
    rc = get_wave_band(ctx, plan.scale[1], plan.offset[1], (int)tile->h, (int)f->height, 3, &bv);
    if (rc != SMR_OK) return rc;
    WJob &J = *out;
    J.yp = view_of(f->planes[0]); J.up = view_of(f->planes[1]); J.vp = f->planes[2] ? view_of(f->planes[2]) : J.up;
    J.dst = view_of(tile);
    J.src_w = (int)f->width; J.src_h = (int)f->height;
    J.conv = m_conv_constants(f->format == SMR_FRAME_PLANAR_YUVJ420);
    J.h_meta = (const int4 *)bh.meta; J.h_frag = bh.frag; J.NKS = bh.K; J.n_pairs = bh.n_units;
    J.k01 = bh.k01 ? 1 : 0;
    J.n_htiles = ((int)tile->w + 15) / 16;
    J.v_meta = (const int2 *)bv.meta; J.v_frag = bv.frag; J.KV = bv.K; J.n_vtiles = bv.n_units;
    J.pieces = W_WAVES;
    J.nv12 = f->format == SMR_FRAME_NV12 ? 1 : 0;
    if (J.nv12) J.vp = J.up;
    J.layer = -1; J.ox = 0; J.oy = 0;
    J.perp = 0; J.single = 0;
    return SMR_OK;
}

// The plane-source builds (FL & 262144): a planar 4:2:0 / NV12 frame the exact block converter takes (smr_conv_rgb12_ok: 4:2:0, width a multiple of 4
// from 8, even height, dword-aligned planes), a two-pass horizontal-first plan without box pre-reduction, inside one of the class builds'
// windows (vertical bands on the origin-0 chunk grid: axis 5), 16-byte aligned tile rows.  *cls: 0 <4, 2> | 1 <8, 3> | 2 <8, 2>.
bool can_fuse_planes(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile, int *cls = nullptr) {
    if (!ctx->plane_source || ctx->ingest_impl == SMR_INGEST_VALU_F32 || !f || !smr_conv_rgb12_ok(ctx, f)) return false;
    if (!(plan.kind == 2 && plan.levels[0] == 0 && plan.levels[1] == 0 && plan.axis[0] == 0 && plan.axis[1] == 1)) return false;
    if (((uintptr_t)tile->ptr % 16) || (tile->pitch % 16) || f->height < 4) return false;
    int NKS, KV, unused;
    if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[0], plan.offset[0], (int)tile->w, (int)f->width, 2)) NKS = t->K;
    else wave_band_geometry(plan.scale[0], plan.offset[0], (int)tile->w, (int)f->width, 2, &NKS, &unused);
    if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[1], plan.offset[1], (int)tile->h, (int)f->height, 5)) KV = t->K;
    else wave_band_geometry(plan.scale[1], plan.offset[1], (int)tile->h, (int)f->height, 5, &KV, &unused);
    const int c = (NKS <= 4 && KV == 2) ? 0 : (NKS <= 8 && KV == 3) ? 1 : (NKS <= 8 && KV == 2) ? 2 : -1;
    if (cls) *cls = c;
    return c >= 0;
}
int make_wave_job_planes(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile, WJob *out) {
    WaveBand bh, bv;
    int rc = get_wave_band(ctx, plan.scale[0], plan.offset[0], (int)tile->w, (int)f->width, 2, &bh);
    if (rc != SMR_OK) return rc;
    rc = get_wave_band(ctx, plan.scale[1], plan.offset[1], (int)tile->h, (int)f->height, 5, &bv);
    if (rc != SMR_OK) return rc;
    WJob &J = *out;
    memset(&J, 0, sizeof(J));
    J.yp = view_of(f->planes[0]); J.up = view_of(f->planes[1]); J.vp = f->planes[2] ? view_of(f->planes[2]) : J.up;
    J.dst = view_of(tile);
    J.src_w = (int)f->width; J.src_h = (int)f->height;
    J.conv = m_conv_constants(true);  // (unused by these builds)
    J.h_meta = (const int4 *)bh.meta; J.h_frag = bh.frag; J.NKS = bh.K; J.n_pairs = bh.n_units;
    J.k01 = bh.k01 ? 1 : 0;
    J.n_htiles = ((int)tile->w + 15) / 16;
    J.v_meta = (const int2 *)bv.meta; J.v_frag = bv.frag; J.KV = bv.K; J.n_vtiles = bv.n_units;
    J.pieces = W_WAVES;
    J.nv12 = f->format == SMR_FRAME_NV12 ? 1 : 0;
    J.full = f->format == SMR_FRAME_PLANAR_YUVJ420 ? 1 : 0;
    J.layer = -1;
    return SMR_OK;
}

// A single-axis plan written as the two-pass plan the job builders understand: the scaled axis first, then a scale-1 "pass" along the
// other axis that only carries the perpendicular crop offset (its band gives the pieces their chunk ranges; the 32768 builds do not
// run it).  The caller sets WJob::perp and launches the single-axis build.
inline smr_resample_plan single_axis_as_two_pass(const smr_resample_plan &plan) {
    smr_resample_plan p = plan;
    p.kind = 2;
    p.axis[1] = 1 - plan.axis[0];
    p.scale[1] = 1.0f;
    p.offset[1] = (float)plan.perp_offset[0];
    p.perp_offset[0] = p.perp_offset[1] = 0;
    return p;
}

// An RGBA8 node texture with alpha == 1 as the source (k_ingest_wave's 8192 builds): a frame of a format the fused conversion does not
// read (4:2:2, 4:4:4, packed YUV, BGRA / ARGB) after the exact converter, or an opaque surface.  Horizontal-first Lanczos plans with
// the kernel's window limits (shrink factors up to ~3.5).
// (bpp 8: an RGBA16F surface that is already box-reduced — the plan's levels then describe what was done to get it;
//  bpp 3: an RGB12 node — src describes the node in PIXELS (w x h), its rows hold 3 w bytes: make_wave_job_rgba's `rgb12`)
// *single (may be null): the pair windows are too wide but one tile per unit fits (axis 4 bands) — make_wave_job_rgba's `single`
bool can_fuse_wave_rgba(smr_ctx *ctx, const SurfView &src, const smr_resample_plan &plan, const smr_surface *tile, int bpp = 4, bool *single = nullptr) {
    if (single) *single = false;
    if (ctx->ingest_impl == SMR_INGEST_VALU_F32) return false;
    if (!(plan.kind == 2 && (bpp == 8 || (plan.levels[0] == 0 && plan.levels[1] == 0)) && plan.axis[0] == 0 && plan.axis[1] == 1)) return false;
    if (src.w < 8 || src.h < 2 || (((uintptr_t)src.ptr) % 16) || (src.pitch % 16) || src.pitch < (((size_t)src.w + 3u) & ~(size_t)3u) * (size_t)bpp) return false;
    if (((uintptr_t)tile->ptr % 16) || (tile->pitch % 16)) return false;
    int NKS, KV, unused;
    if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[0], plan.offset[0], (int)tile->w, src.w, 2)) NKS = t->K;
    else wave_band_geometry(plan.scale[0], plan.offset[0], (int)tile->w, src.w, 2, &NKS, &unused);
    if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[1], plan.offset[1], (int)tile->h, src.h, 3)) KV = t->K;
    else wave_band_geometry(plan.scale[1], plan.offset[1], (int)tile->h, src.h, 3, &KV, &unused);
    if (NKS > W_NKS_MAX && single && KV <= W_KV_MAX) {
        if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[0], plan.offset[0], (int)tile->w, src.w, 4)) NKS = t->K;
        else wave_band_geometry(plan.scale[0], plan.offset[0], (int)tile->w, src.w, 4, &NKS, &unused);
        *single = NKS <= W_NKS_MAX;
        return *single;
    }
    return NKS <= W_NKS_MAX && KV <= W_KV_MAX;
}

int make_wave_job_rgba(smr_ctx *ctx, const SurfView &src, const smr_resample_plan &plan, const smr_surface *tile, WJob *out, bool single = false, bool rgb12 = false) {
    WaveBand bh, bv;
    int rc = get_wave_band(ctx, plan.scale[0], plan.offset[0], (int)tile->w, src.w, single ? 4 : 2, &bh);
    if (rc != SMR_OK) return rc;
    rc = get_wave_band(ctx, plan.scale[1], plan.offset[1], (int)tile->h, src.h, 3, &bv);
    if (rc != SMR_OK) return rc;
    WJob &J = *out;
    memset(&J, 0, sizeof(J));
    J.yp = src; J.up = src; J.vp = src;
    J.dst = view_of(tile);
    J.src_w = src.w; J.src_h = src.h;
    J.conv = m_conv_constants(true);
    J.h_meta = (const int4 *)bh.meta; J.h_frag = bh.frag; J.NKS = bh.K; J.n_pairs = bh.n_units;
    J.k01 = bh.k01 ? 1 : 0;
    J.n_htiles = ((int)tile->w + 15) / 16;
    J.v_meta = (const int2 *)bv.meta; J.v_frag = bv.frag; J.KV = bv.K; J.n_vtiles = bv.n_units;
    J.pieces = W_WAVES;
    J.layer = -1;
    J.single = single ? 1 : 0;
    if (single) J.k01 = 0;
    return SMR_OK;
}

// A job for a vertical-first plan: the same kernel on the transposed frame (fused conversion only).  `slot0`: four surface-cache slots of
// the caller's for the transposed planes and tile.  *ok = false: not a case for this route.  The plane transposes are enqueued here; the caller
// launches the job with its others and then runs launch_transpose<u32> on every MTransposeBack.
int make_wave_job_transposed(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, smr_surface *tile, size_t slot0, WJob *out, bool *ok,
                             MTransposeBack *back) {
    *ok = false;
    if (!fused_conversion(ctx)) return SMR_OK;
    if (!f || !f->planes[0] || !f->planes[1]) return SMR_OK;
    if (!(plan.kind == 2 && plan.levels[0] == 0 && plan.levels[1] == 0 && plan.axis[0] == 1 && plan.axis[1] == 0)) return SMR_OK;
    const bool nv12 = f->format == SMR_FRAME_NV12;
    if (f->format != SMR_FRAME_PLANAR_YUV420 && f->format != SMR_FRAME_PLANAR_YUVJ420 && !nv12) return SMR_OK;
    if (!nv12 && !f->planes[2]) return SMR_OK;
    if (f->width % 2 || f->height % 2) return SMR_OK;
    const u32 cw = f->width / 2, ch = f->height / 2;
    smr_frame ft;
    memset(&ft, 0, sizeof(ft));
    ft.format = f->format; ft.width = f->height; ft.height = f->width;
    ft.planes[0] = smr_cached_surface(ctx, slot0, f->height, f->width, SMR_PX_R8);
    ft.planes[1] = smr_cached_surface(ctx, slot0 + 1, ch, cw, nv12 ? SMR_PX_RG8 : SMR_PX_R8);
    if (!nv12) ft.planes[2] = smr_cached_surface(ctx, slot0 + 2, ch, cw, SMR_PX_R8);
    smr_surface *tile_t = smr_cached_surface(ctx, slot0 + 3, tile->h, tile->w, SMR_PX_RGBA8);
    if (!ft.planes[0] || !ft.planes[1] || (!nv12 && !ft.planes[2]) || !tile_t) return SMR_ERR_OOM;
    smr_resample_plan pt = plan;  // the first pass of the plan (the source's rows) is the transposed frame's horizontal pass
    pt.axis[0] = 0; pt.axis[1] = 1;
    if (!can_fuse_wave(ctx, &ft, pt, tile_t)) return SMR_OK;
    if (int rc = make_wave_job(ctx, &ft, pt, tile_t, out)) return rc;
    if (int rc = launch_transpose<u8>(ctx, f->planes[0], ft.planes[0])) return rc;
    if (nv12) {
        if (int rc = launch_transpose<u16>(ctx, f->planes[1], ft.planes[1])) return rc;
    } else {
        if (int rc = launch_transpose<u8>(ctx, f->planes[1], ft.planes[1])) return rc;
        if (int rc = launch_transpose<u8>(ctx, f->planes[2], ft.planes[2])) return rc;
    }
    back->tile_t = tile_t;
    back->tile = tile;
    *ok = true;
    return SMR_OK;
}

// ... and a vertical-first plan on an RGBA8 node texture: the node transposed in, the tile transposed back
int make_wave_job_rgba_transposed(smr_ctx *ctx, const SurfView &src, const smr_resample_plan &plan, smr_surface *tile, size_t slot0, WJob *out, bool *ok,
                                  MTransposeBack *back) {
    *ok = false;
    if (ctx->ingest_impl == SMR_INGEST_VALU_F32) return SMR_OK;
    if (!(plan.kind == 2 && plan.levels[0] == 0 && plan.levels[1] == 0 && plan.axis[0] == 1 && plan.axis[1] == 0)) return SMR_OK;
    smr_surface *node_t = smr_cached_surface(ctx, slot0, (u32)src.h, (u32)src.w, SMR_PX_RGBA8);
    smr_surface *tile_t = smr_cached_surface(ctx, slot0 + 3, tile->h, tile->w, SMR_PX_RGBA8);
    if (!node_t || !tile_t) return SMR_ERR_OOM;
    smr_resample_plan pt = plan;
    pt.axis[0] = 0; pt.axis[1] = 1;
    bool single = false;
    if (!can_fuse_wave_rgba(ctx, view_of(node_t), pt, tile_t, 4, &single)) return SMR_OK;
    if (int rc = make_wave_job_rgba(ctx, view_of(node_t), pt, tile_t, out, single)) return rc;
    smr_surface node;  // non-owning alias of the node view
    node.ptr = src.ptr; node.pitch = src.pitch; node.w = (u32)src.w; node.h = (u32)src.h; node.fmt = SMR_PX_RGBA8;
    if (int rc = launch_transpose<u32>(ctx, &node, node_t)) return rc;
    back->tile_t = tile_t;
    back->tile = tile;
    *ok = true;
    return SMR_OK;
}

// builds: generic | the benchmark scenes' class (pair windows of <= 4 k-steps, pass-2 windows of 2: scales around 1.5) | that class with
//         its two always-zero weight fragments skipped | the north-star target's class (windows of <= 8 k-steps, pass-2 windows of 3:
//         scales around 3);  each plain | direct output (2048) | NV12-capable (4096) | both
typedef void (*WaveKernel)(const WArgs, const float *, const u32 *);
// (the builds that read the PLANES and convert on the fly — the fused conversion, ingest implementation 5 — exist in laboratory builds only:
//  a product build instantiates the node-texture builds below and nothing else)
#ifdef SMR_LAB
constexpr WaveKernel W_KERNELS[] = {k_ingest_wave<0, 0, 0>,    k_ingest_wave<4, 2, 0>,    k_ingest_wave<4, 2, 1>,    k_ingest_wave<8, 3, 0>,
                                    k_ingest_wave<0, 0, 2048>, k_ingest_wave<4, 2, 2048>, k_ingest_wave<4, 2, 2049>, k_ingest_wave<8, 3, 2048>,
                                    k_ingest_wave<0, 0, 4096>, k_ingest_wave<4, 2, 4096>, k_ingest_wave<4, 2, 4097>, k_ingest_wave<8, 3, 4096>,
                                    k_ingest_wave<0, 0, 6144>, k_ingest_wave<4, 2, 6144>, k_ingest_wave<4, 2, 6145>, k_ingest_wave<8, 3, 6144>};
constexpr int W_NKERNELS = (int)(sizeof(W_KERNELS) / sizeof(W_KERNELS[0]));
// scales around 2 (windows of 5 .. 8 k-steps, pass-2 windows of 2: what the narrow class and the 4K class leave between them —
// a 4x4 grid of 1080p inputs on a 4K output, tiles in mid-transition): plain | direct output | NV12-capable | both
constexpr WaveKernel W_KERNELS_82[] = {k_ingest_wave<8, 2, 0>, k_ingest_wave<8, 2, 2048>, k_ingest_wave<8, 2, 4096>, k_ingest_wave<8, 2, 6144>};
#endif
// RGBA8 node textures as the source: the same four classes
constexpr WaveKernel W_KERNELS_RGBA[] = {k_ingest_wave<0, 0, 8192>, k_ingest_wave<4, 2, 8192>, k_ingest_wave<4, 2, 8193>, k_ingest_wave<8, 3, 8192>};
// ... RGB12 node textures (the default route of 4:2:0 frames), plain and with direct output
constexpr WaveKernel W_KERNELS_RGB12[] = {k_ingest_wave<0, 0, 8192 + 131072>, k_ingest_wave<4, 2, 8192 + 131072>, k_ingest_wave<4, 2, 8193 + 131072>,
                                          k_ingest_wave<8, 3, 8192 + 131072>};
constexpr WaveKernel W_KERNELS_RGB12_DIRECT[] = {k_ingest_wave<0, 0, 8192 + 131072 + 2048>, k_ingest_wave<4, 2, 8192 + 131072 + 2048>,
                                                 k_ingest_wave<4, 2, 8193 + 131072 + 2048>, k_ingest_wave<8, 3, 8192 + 131072 + 2048>};
// ... plane-source builds (262144: the frame's planes, converted exactly in the wave — the default route of 4:2:0 frames): <4, 2> | <4, 2> without its two
// always-zero fragments | <8, 3> | <8, 2>; planar and (+ 4096) NV12
constexpr WaveKernel W_KERNELS_PLANES[] = {k_ingest_wave<4, 2, 262144>, k_ingest_wave<4, 2, 262145>, k_ingest_wave<8, 3, 262144>, k_ingest_wave<8, 2, 262144>};
constexpr WaveKernel W_KERNELS_PLANES_NV[] = {k_ingest_wave<4, 2, 262144 + 4096>, k_ingest_wave<4, 2, 262145 + 4096>, k_ingest_wave<8, 3, 262144 + 4096>,
                                              k_ingest_wave<8, 2, 262144 + 4096>};
// ... with direct output (2048: the tile's copy-class pixels leave as Y'CbCr, smr_fused.hip)
constexpr WaveKernel W_KERNELS_RGBA_DIRECT[] = {k_ingest_wave<0, 0, 8192 + 2048>, k_ingest_wave<4, 2, 8192 + 2048>, k_ingest_wave<4, 2, 8193 + 2048>,
                                                k_ingest_wave<8, 3, 8192 + 2048>};
// ... with an alpha channel (premultiplied RGBA8: text, images, nested layout nodes, BGRA / ARGB frames): four channels
constexpr WaveKernel W_KERNELS_RGBA_ALPHA[] = {k_ingest_wave<0, 0, 8192 + 65536>, k_ingest_wave<4, 2, 8192 + 65536>, k_ingest_wave<4, 2, 8193 + 65536>,
                                               k_ingest_wave<8, 3, 8192 + 65536>};
// ... and RGBA16F ones (box-pre-reduced plans: residual scales of 2 .. 4; above ~3.2 as one-tile units, WJob::single)
// node textures at scales around 2 (8 k-steps, pass-2 window 2: a 4 x 4 grid of 1080p inputs on a 4K output, 4 x 1080p on 1080p): the wide class's
// software pipeline instead of the generic build's predicated loops
constexpr WaveKernel W_KERNEL_RGBA_82 = k_ingest_wave<8, 2, 8192>, W_KERNEL_RGB12_82 = k_ingest_wave<8, 2, 8192 + 131072>;
constexpr WaveKernel W_KERNEL_RGBA16F = k_ingest_wave<0, 0, 8192 + 16384>, W_KERNEL_RGBA16F_ALPHA = k_ingest_wave<0, 0, 8192 + 16384 + 65536>;
// ... and single-axis plans (generic build): planar | NV12-capable | RGBA8 node texture
#ifdef SMR_LAB
constexpr WaveKernel W_KERNELS_SA[] = {k_ingest_wave<0, 0, 32768>, k_ingest_wave<0, 0, 32768 + 4096>, k_ingest_wave<0, 0, 32768 + 8192>,
                                       k_ingest_wave<0, 0, 32768 + 8192 + 65536>};
#else
constexpr WaveKernel W_KERNELS_SA[] = {nullptr, nullptr, k_ingest_wave<0, 0, 32768 + 8192>, k_ingest_wave<0, 0, 32768 + 8192 + 65536>};
#endif

int launch_wave(smr_ctx *ctx, std::vector<WJob> &jobs, const MDirect *direct = nullptr, bool rgba = false, bool f16 = false, bool sa = false,
                bool alpha = false, bool rgb12 = false, bool planes = false) {
    if (!ctx->wave_attr_set) {  // per device, hence per ctx
        std::vector<WaveKernel> all;
#ifdef SMR_LAB
        all.insert(all.end(), W_KERNELS, W_KERNELS + W_NKERNELS);
        all.insert(all.end(), W_KERNELS_82, W_KERNELS_82 + 4);
#endif
        all.insert(all.end(), W_KERNELS_RGBA, W_KERNELS_RGBA + 4);
        all.insert(all.end(), W_KERNELS_RGBA_DIRECT, W_KERNELS_RGBA_DIRECT + 4);
        all.insert(all.end(), W_KERNELS_PLANES, W_KERNELS_PLANES + 4);
        all.insert(all.end(), W_KERNELS_PLANES_NV, W_KERNELS_PLANES_NV + 4);
        all.insert(all.end(), W_KERNELS_RGB12, W_KERNELS_RGB12 + 4);
        all.insert(all.end(), W_KERNELS_RGB12_DIRECT, W_KERNELS_RGB12_DIRECT + 4);
        all.insert(all.end(), W_KERNELS_RGBA_ALPHA, W_KERNELS_RGBA_ALPHA + 4);
        all.push_back(W_KERNEL_RGBA_82);
        all.push_back(W_KERNEL_RGB12_82);
        all.push_back(W_KERNEL_RGBA16F);
        all.push_back(W_KERNEL_RGBA16F_ALPHA);
        all.insert(all.end(), W_KERNELS_SA, W_KERNELS_SA + 4);
        for (WaveKernel k : all) {
            if (!k) continue;
            SMR_HIP(ctx, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipFuncAttributes fa;
            SMR_HIP(ctx, hipFuncGetAttributes(&fa, (const void *)k));
            // the decode LUT is addressed by its LDS offset 0: holds while the kernel declares no static LDS
            if (fa.sharedSizeBytes != 0) return smr_fail(ctx, SMR_ERR_INTERNAL, "k_ingest_wave: %zu B of static LDS in front of the dynamic segment", (size_t)fa.sharedSizeBytes);
        }
        ctx->wave_attr_set = true;
    }
    if (int rc = flush_wave_builds(ctx)) return rc;
    StageScope scope(ctx, SMR_STAGE_FUSED_INGEST);
    for (size_t j0 = 0; j0 < jobs.size(); j0 += MAX_WJOBS_PER_LAUNCH) {
        const size_t nj = jobs.size() - j0 < (size_t)MAX_WJOBS_PER_LAUNCH ? jobs.size() - j0 : (size_t)MAX_WJOBS_PER_LAUNCH;
        WArgs args;
        memset(&args, 0, sizeof(args));
        bool cls432 = true, cls83 = true, cls82 = true, any_nv = false, k01 = true;
        int nks_max = 1;
        long long tile_rows = 0;  // sum over jobs of pairs x tile rows: the unit of work
        for (size_t j = 0; j < nj; j++) {
            const WJob &J = jobs[j0 + j];
            cls432 = cls432 && J.NKS <= 4 && J.KV == 2;
            cls83 = cls83 && J.NKS <= 8 && J.KV == 3;
            cls82 = cls82 && J.NKS <= 8 && J.KV == 2;
            any_nv = any_nv || J.nv12;
            k01 = k01 && J.k01;
            nks_max = J.NKS > nks_max ? J.NKS : nks_max;
            tile_rows += (long long)J.n_pairs * J.n_vtiles;
        }
        if (f16 || sa) cls432 = cls83 = cls82 = false;  // (one generic build)
        if (planes && !(cls432 || cls83 || cls82)) return smr_fail(ctx, SMR_ERR_INTERNAL, "k_ingest_wave: a plane-source job outside the class builds");
        const bool planes82 = planes && cls82 && !cls432;
        const bool node82 = cls82 && !cls432 && rgba && !alpha && !direct && ctx->wave_node82;  // (node textures: its own two builds)
        if (!planes && (rgba || cls432)) cls82 = false;
        int ki = cls432 ? (k01 ? 2 : 1) : (cls83 ? 3 : 0);
        if (!rgba && !sa) {
            if (direct) ki += 4;
            if (any_nv) ki += 8;
        }
        const int cls_nks = cls432 ? 4 : 0;  // (the wide class lays its LDS out for the jobs' own k-step counts)
        const int sa_i = rgba ? (alpha ? 3 : 2) : (any_nv ? 1 : 0);
#ifndef SMR_LAB
        if (!rgba && !planes) return smr_fail(ctx, SMR_ERR_INTERNAL, "k_ingest_wave: the one-code-per-stage conversion builds exist in laboratory builds only");
        constexpr WaveKernel W_KERNELS_82[4] = {nullptr, nullptr, nullptr, nullptr}, W_KERNELS[16] = {};
#endif
        const int planes_i = cls432 ? (k01 ? 1 : 0) : (planes82 ? 3 : 2);
        const WaveKernel kern = planes ? (any_nv ? W_KERNELS_PLANES_NV[planes_i] : W_KERNELS_PLANES[planes_i])
                                : node82 ? (rgb12 ? W_KERNEL_RGB12_82 : W_KERNEL_RGBA_82)
                                : cls82 ? W_KERNELS_82[(direct ? 1 : 0) + (any_nv ? 2 : 0)]
                                : sa ? W_KERNELS_SA[sa_i]
                                   : f16 ? (alpha ? W_KERNEL_RGBA16F_ALPHA : W_KERNEL_RGBA16F)
                                         : alpha ? W_KERNELS_RGBA_ALPHA[ki]
                                                 : rgb12 ? (direct ? W_KERNELS_RGB12_DIRECT[ki] : W_KERNELS_RGB12[ki])
                                                         : rgba ? (direct ? W_KERNELS_RGBA_DIRECT[ki] : W_KERNELS_RGBA[ki]) : W_KERNELS[ki];
        if (planes) ki = 800 + planes_i + (any_nv ? 4 : 0);
        else if (node82) ki = rgb12 ? 710 : 700;  // (occupancy cache keys)
        else if (cls82) ki = 500 + (direct ? 1 : 0) + (any_nv ? 2 : 0);
        else if (sa) ki = 300 + sa_i;
        else if (rgba) ki += f16 ? (alpha ? 250 : 200) : alpha ? 400 : rgb12 ? (direct ? 650 : 600) : (direct ? 150 : 100);
        ctx->kernel_launches[rgba ? SMR_KERNEL_INGEST_WAVE_RGBA : SMR_KERNEL_INGEST_WAVE]++;  // (slot 0: the builds that read the frame's planes)
        // (the narrow class keeps its pass-1 band in registers: no LDS for it)
        args.b_bytes = (cls_nks && SMR_WAVE_PIPE && SMR_WAVE_B_REGS) ? 0 : w_band_bytes(cls_nks ? cls_nks : nks_max);
        // (node-texture builds stage nothing: the area behind the band only holds the alpha builds' 1 KB table)
        args.raw_bytes = planes ? w_fx_node_bytes(cls_nks ? cls_nks : nks_max) : rgba ? (alpha ? 1024 / W_WAVES : 0) : (w_raw_bytes(cls_nks ? cls_nks : nks_max) + 15) & ~15;
        const size_t lds = (size_t)W_OFF_B + args.b_bytes + (planes ? W_FX_TABLE_BYTES : 0) + (size_t)W_WAVES * args.raw_bytes;
        if (lds > 160 * 1024) return smr_fail(ctx, SMR_ERR_INTERNAL, "k_ingest_wave: %zu B of LDS", lds);
        // as many waves as are resident at once (registers and LDS), one piece each: every wave starts and ends with the launch
        int per_cu = 0;
        for (auto &o : ctx->mfma_occupancy)
            if (o.kernel == 1000 + ki && o.lds == lds) per_cu = o.per_cu;
        if (!per_cu) {
            SMR_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, W_THREADS, lds));
            if (per_cu < 1) per_cu = 1;
            ctx->mfma_occupancy.push_back({1000 + ki, lds, per_cu});
        }
        const int reserve = ctx->ingest_reserve_cus >= 0 && ctx->ingest_reserve_cus < ctx->cu_count ? ctx->ingest_reserve_cus : 0;
        // (SMR_OPT_SHARED_DEVICE: four workgroups = two waves per SIMD, the rest of the register file for the other lanes' kernels)
        const int wg_cap = ctx->ingest_wg_per_cu > 0 ? ctx->ingest_wg_per_cu : ctx->shared_device ? 4 : 64;
        const long long slots = (long long)(per_cu > wg_cap ? wg_cap : per_cu) * (ctx->cu_count - reserve) * W_WAVES;  // resident waves
        // pieces per column pair: the same number of tile rows per wave for every job, rounded so that the launch fits the resident set
        double rows_per_wave = (double)tile_rows / (double)slots;
        // (a piece re-converts its window's head, so short pieces cost conversions — but a launch that cannot fill the chip is bound by
        //  the length of a wave's chain, not by throughput: configs[4]'s single-input node 34 -> 24 us with one tile row per wave
        //  instead of two, profiles/r03_min_rows.txt)
        const double min_rows = ctx->ingest_min_rows > 0 ? (double)ctx->ingest_min_rows : 1.0;
        if (rows_per_wave < min_rows) rows_per_wave = min_rows;
        int total = 0;
        for (int pass = 0; pass < 8; pass++) {
            total = 0;
            for (size_t j = 0; j < nj; j++) {
                WJob &J = jobs[j0 + j];
                int p = (int)floor((double)J.n_vtiles / rows_per_wave);
                p = p / W_WAVES * W_WAVES;
                if (p < W_WAVES) p = W_WAVES;
                J.pieces = p;
                total += J.n_pairs * (p / W_WAVES);
            }
            if ((long long)total * W_WAVES <= slots || rows_per_wave > 1e6) break;
            rows_per_wave *= 1.0 + 1.0 / 16.0;
        }
        total = 0;
        for (size_t j = 0; j < nj; j++) {
            args.jobs[j] = jobs[j0 + j];
            args.wg_prefix[j] = total;
            total += jobs[j0 + j].n_pairs * (jobs[j0 + j].pieces / W_WAVES);
        }
        args.wg_prefix[nj] = total;
        args.n_jobs = (int)nj;
        args.direct = direct;
        const int blocks = (total + 7) & ~7;
        if (ctx->debug_ingest)
            fprintf(stderr, "k_ingest_wave[%d]: %zu jobs, lds %zu B (%d WG/CU), %d workgroups of %d waves, job0: NKS %d KV %d pairs %d vtiles %d pieces %d\n", ki,
                    nj, lds, per_cu, total, W_WAVES, args.jobs[0].NKS, args.jobs[0].KV, args.jobs[0].n_pairs, args.jobs[0].n_vtiles,
                    args.jobs[0].pieces);
#if SMR_WAVE_TIMING
        args.dbg = (unsigned long long *)smr_scratch(ctx, 7, 128);
        if (args.dbg) SMR_HIP(ctx, hipMemsetAsync(args.dbg, 0, 128, ctx->stream));
#endif
        if (blocks > 0) hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(W_THREADS), lds, ctx->stream, args, ctx->d_tables, ctx->d_lut16);
        SMR_HIP(ctx, hipGetLastError());
#if SMR_WAVE_TIMING
        if (args.dbg) {
            unsigned long long h[16];
            SMR_HIP(ctx, hipMemcpyAsync(h, args.dbg, 128, hipMemcpyDeviceToHost, ctx->stream));
            SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const double w = 1e3 * (double)(h[10] ? h[10] : 1);
            fprintf(stderr, "k_ingest_wave kcycles per timed wave (%llu waves, %.1f chunks, %.1f tile rows each): prologue %.1f | k-steps %.1f land %.1f last-mfma+ring %.1f pass2-mfma %.1f encode %.1f store+fetch %.1f issue %.1f\n",
                    h[10], (double)h[8] / (double)(h[10] ? h[10] : 1), (double)h[9] / (double)(h[10] ? h[10] : 1), h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[6] / w,
                    h[7] / w, h[5] / w);
        }
#endif
    }
    return SMR_OK;
}

}  // namespace
#endif  // !SMR_EMU
