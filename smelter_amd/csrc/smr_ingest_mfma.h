// smr_ingest_mfma.h — wave A of the hot path on the matrix cores: k_ingest_mfma (included by smr_fused.hip only).
//
// Same job as k_ingest_resample (smr_fused_ingest.h): planar 4:2:0 frame -> dst-sized sRGB RGBA8 tile, i.e. the reference's
// planar_yuv_to_rgba pass (wgpu/format/planar_yuv_to_rgba.wgsl:35-58) followed by the two Lanczos3 passes of
// transformations/layout/resample.wgsl:31-87 with their Rgba16Float intermediate (layout/resampler.rs:25-28) — but the two
// separable filter passes are recast as banded GEMMs on v_mfma_f32_16x16x32_f16:
//
//   pass 1   H[r][x]  = sum_k  T[r][k]  * Wh[k][x]      M = 16 source rows (one chunk), N = 16 output columns, K = source columns
//   pass 2   O[x][y]  = sum_r  H[x][r]  * Wv[r][y]      M = 16 output columns, N = 16 output rows,            K = source rows
//
// * T = the node texture's sRGB-decoded linear value per texel, kept as an f16 pair (hi, lo = t - hi) interleaved along K, so the
//   product is exact to ~2^-22 although the matrix cores take f16 (one f16 alone costs ~1 % of the output bytes one LSB,
//   tools/mfma_precision_sim.py); the normalised pass-1 weights are f16 pairs too (hi, lo), applied by a second MFMA on the
//   same A operand — with single-f16 weights white-noise content lost 1.8 % of its bytes to a 1-LSB step and one pixel in 10^5
//   to two.  The reference's own quantisation points are kept: u8 node texture, f16 (RTNE)
//   between the passes, u8 sRGB tile.  Deviation from the oracle: <= 1 LSB, > 99.9 % of the bytes identical (tests/).
// * Clamp-to-edge is folded into the weight band (taps that clamp onto the same texel are summed), out-of-band entries are 0.
// * A 768-thread workgroup (12 waves) owns a 64-column strip of the tile over a range of 16-row output tiles and streams the
//   source rows through LDS in chunks of 16, as a two-stage pipeline with one barrier per chunk (mfma_piece below):
//     8 convert waves   stage the raw Y/U/V dwords of the next chunks (loads issued two chunks ahead), convert chunk k — one 4x1 pixel
//                       block per lane: chroma by v_dot4_u32_u8 (9/3/3/1 bilinear in 1/16 units, exact), BT.709 with folded
//                       constants (one FMA per term), u8 quantisation + sRGB decode + (hi, lo) split in one LDS lookup, 16-byte
//                       stores into T[k & 1]
//     4 filter waves    (one 16-column N tile each) pass 1 of chunk k - 1: 3 channels x KH k-steps x 2 MFMAs, result -> f16 -> a ring
//                       of rows Mh (LDS, [8-row granule][column]); then, for every 16-row output tile whose window is complete,
//                       pass 2: 3 x KV MFMAs, sRGB encode, 16-byte stores (or Y'CbCr: direct output)
//   A ring column is written and read by one wave only; T and the raw footprint are double-buffered across the barrier.
// * Vertical-first plans run on the transposed frame (make_mfma_job_transposed); NV12 frames and direct output are separate builds
//   of the kernel (template flags), so the plain build carries none of their registers.
#pragma once

#include "smr_ingest_common.h"

#include <cmath>
#include <vector>

namespace {

constexpr int M_NT = 4;             // 16-column N tiles per strip = filter waves of a workgroup
#ifndef SMR_MFMA_CONV_WAVES
#define SMR_MFMA_CONV_WAVES 8
#endif
constexpr int M_WAVES = M_NT + SMR_MFMA_CONV_WAVES;  // + convert waves (8: one 4x1 block per lane and chunk; 4 measured 10 % slower)
constexpr int M_THREADS = M_WAVES * 64;
constexpr int M_SW = 16 * M_NT;     // strip width
constexpr int M_CH = 16;            // source rows per chunk = M of pass 1
constexpr int M_KH_MAX = 6;         // k-steps of 32 (16 texels as hi/lo pairs) in pass 1
constexpr int M_KV_MAX = 4;         // k-steps of 32 rows in pass 2
constexpr int M_NG_MAX = 64;        // 4-texel column groups of a strip's source footprint (one luma row = one wave-wide load)
constexpr int M_WSPAN = 136;        // >= 32 * M_KV_MAX, >= 16 * M_KH_MAX

// ------------------------------------------------------------------ weight bands in MFMA B-operand layout (device cache)
// Per tile of 16 outputs: meta = (base, last) — base = first source texel of the K window (pass 1: multiple of 4; pass 2:
// = 1 mod 8), last = last texel with a non-zero weight — and K fragments of 64 lanes x 8 f16:
//   lane l holds W[k = 32 j + 8 (l >> 4) + e][n = l & 15], e = 0..7;  pass 1: texel = base + (k >> 1) (hi and lo share a weight);
//   every weight is an f16 pair (hi, lo): K fragments of each per tile.
__host__ __device__ inline int mfma_window_base(int lo, int axis) { return axis == 0 ? (lo & ~3) : (((lo - 1) & ~7) + 1); }

// One launch builds every band the call is missing (a tile that resizes needs two new ones per frame; sixteen such tiles used to
// be thirty-two launches of ~12 us each, back to back on the stream).
struct WBuild {
    float scale, offset;
    int taps, n_dst, n_src, axis, K, tile0;  // tile0: first workgroup of this band
    int2 *meta;
    uint4 *frag;
};
constexpr int MAX_WBUILDS = 32;
struct WBatch {
    WBuild b[MAX_WBUILDS];
    int n;
};

__global__ __launch_bounds__(64) void k_build_mfma_weights(const WBatch args) {
    __shared__ float s_w[16][M_WSPAN];
    __shared__ _Float16 s_q[16][M_WSPAN], s_r[16][M_WSPAN];
    int bi = 0;
    while (bi + 1 < args.n && args.b[bi + 1].tile0 <= (int)blockIdx.x) bi++;
    const WBuild &B = args.b[bi];
    const float scale = B.scale, offset = B.offset;
    const int taps = B.taps, n_dst = B.n_dst, n_src = B.n_src, axis = B.axis, K = B.K;
    int2 *__restrict__ meta = B.meta;
    uint4 *__restrict__ frag = B.frag;
    const int t = (int)blockIdx.x - B.tile0, lane = threadIdx.x;
    const int o0 = 16 * t, o1 = min(o0 + 15, n_dst - 1);
    const int lo = clampi(lanczos_first(o0, scale, offset), 0, n_src - 1);
    const int hi = clampi(lanczos_first(o1, scale, offset) + taps - 1, 0, n_src - 1);
    const int base = mfma_window_base(lo, axis);
    const int span = axis == 0 ? 16 * K : 32 * K;
    if (lane == 0) meta[t] = make_int2(base, hi);
    for (int i = lane; i < 16 * M_WSPAN; i += 64) {
        (&s_w[0][0])[i] = 0.0f;
        (&s_q[0][0])[i] = (_Float16)0.0f;
        (&s_r[0][0])[i] = (_Float16)0.0f;
    }
    __syncthreads();
    if (lane < 16 && o0 + lane < n_dst) {
        float w[MAX_TAPS];
        float ws;
        const int first = lanczos_weights(o0 + lane, scale, offset, taps, w, &ws);
        for (int i = 0; i < taps; i++) {
            const int idx = clampi(first + i, 0, n_src - 1) - base;
            if (idx >= 0 && idx < span) s_w[lane][idx] += w[i] / ws;
        }
        // two f16 terms per weight: hi = f16(w), lo = f16(w - hi) — the pair carries 22 bits, so no output's weights need
        // renormalising and a high-contrast neighbourhood cannot push a result over an f16 rounding boundary of the intermediate
        float sum = 0.0f;
        for (int i = 0; i < span; i++) {
            const _Float16 q = (_Float16)s_w[lane][i];
            s_q[lane][i] = q;
            s_r[lane][i] = (_Float16)(s_w[lane][i] - (float)q);
            sum += (float)q;
        }
        if (axis == 1) {
            // pass 2 applies the hi term alone: error feedback — the rounding residual of the whole row goes to the tap that can
            // absorb it best, so that the hi terms of every output sum to 1 (flat areas stay exact)
            const float r = 1.0f - sum;
            int best = -1;
            float best_err = 1e30f;
            for (int i = 0; i < span; i++) {
                if (s_w[lane][i] == 0.0f) continue;
                const float c = (float)s_q[lane][i] + r;
                const float err = fabsf((float)(_Float16)c - c);
                if (err < best_err) { best_err = err; best = i; }
            }
            if (best >= 0) s_q[lane][best] = (_Float16)((float)s_q[lane][best] + r);
        }
    }
    __syncthreads();
    // fragments: [tile][hi | lo][K][64 lanes].  Pass 1 multiplies the (t_hi, t_lo) texel pairs by (w_hi, w_hi) and, in a second
    // MFMA on the same A operand, by (w_lo, 0): t_hi w_hi + t_lo w_hi + t_hi w_lo.  Pass 2 (exact f16 rows) takes w_hi only.
    for (int j = 0; j < K; j++) {
        f16x8 v, r;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int kk = 32 * j + 8 * (lane >> 4) + e;
            v[e] = s_q[lane & 15][axis == 0 ? (kk >> 1) : kk];
            r[e] = axis == 0 ? ((kk & 1) ? (_Float16)0.0f : s_r[lane & 15][kk >> 1]) : s_r[lane & 15][kk];
        }
        frag[((size_t)t * 2 * K + j) * 64 + lane] = __builtin_bit_cast(uint4, v);
        frag[((size_t)t * 2 * K + K + j) * 64 + lane] = __builtin_bit_cast(uint4, r);
    }
}

struct MfmaBand {
    const int2 *meta;
    const uint4 *frag;
    int K;        // k-steps
    int n_tiles;
    int max_span; // max over tiles of (last - base + 1)
};

// host twin of the tile geometry (same f32 sequence as the device: lanczos_first is __host__ __device__)
void mfma_band_geometry(float scale, float offset, int n_dst, int n_src, int axis, int *K, int *max_span) {
    const int taps = host_taps(scale);
    const int n_tiles = (n_dst + 15) / 16;
    int span = 1;
    for (int t = 0; t < n_tiles; t++) {
        const int o0 = 16 * t, o1 = o0 + 15 < n_dst - 1 ? o0 + 15 : n_dst - 1;
        int lo = lanczos_first(o0, scale, offset), hi = lanczos_first(o1, scale, offset) + taps - 1;
        lo = lo < 0 ? 0 : (lo > n_src - 1 ? n_src - 1 : lo);
        hi = hi < 0 ? 0 : (hi > n_src - 1 ? n_src - 1 : hi);
        const int s = hi - mfma_window_base(lo, axis) + 1;
        span = s > span ? s : span;
    }
    *max_span = span;
    *K = axis == 0 ? (2 * span + 31) / 32 : (span + 31) / 32;
}

// the cached band of (scale, offset, n_dst, n_src, axis), if any: a frame of a scene at rest finds all of its bands here and
// never walks the tile geometry on the host
smr_ctx::MfmaTable *find_mfma_table(smr_ctx *ctx, float scale, float offset, int n_dst, int n_src, int axis) {
    for (auto &t : ctx->mfma_tables)
        if (t.dev && t.n_dst == n_dst && t.n_src == n_src && t.axis == axis && t.scale == scale && t.offset == offset) return &t;
    return nullptr;
}

int get_mfma_band(smr_ctx *ctx, float scale, float offset, int n_dst, int n_src, int axis, MfmaBand *out) {
    const int n_tiles = (n_dst + 15) / 16;
    smr_ctx::MfmaTable *hit = find_mfma_table(ctx, scale, offset, n_dst, n_src, axis), *victim = nullptr;
    if (!hit) {
        int K, span;
        mfma_band_geometry(scale, offset, n_dst, n_src, axis, &K, &span);
        // never evict a table the current call already handed to a job that is not launched yet
        for (auto &t : ctx->mfma_tables)
            if (t.last_call != ctx->weight_call && (!victim || t.last_use < victim->last_use)) victim = &t;
        if (!victim || ctx->mfma_tables.size() < 64) {
            ctx->mfma_tables.emplace_back();
            victim = &ctx->mfma_tables.back();
        }
        const size_t meta_bytes = ((size_t)n_tiles * sizeof(int2) + 15) & ~(size_t)15;
        const size_t need = meta_bytes + (size_t)n_tiles * 2 * K * 64 * sizeof(uint4);
        // (a band of an earlier call that was allocated but never needed — its job did not fit — is dropped with its table)
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
        victim->K = K; victim->max_span = span; victim->meta_bytes = meta_bytes; victim->ngm = 0;
        // (built by flush_mfma_builds, one launch for all the bands a call misses, before the kernel that reads them)
        smr_ctx::PendingBand pb;
        pb.scale = scale; pb.offset = offset; pb.taps = host_taps(scale); pb.n_dst = n_dst; pb.n_src = n_src; pb.axis = axis; pb.K = K;
        pb.n_tiles = n_tiles; pb.meta = victim->dev; pb.frag = (u8 *)victim->dev + meta_bytes;
        ctx->pending_bands.push_back(pb);
        hit = victim;
    }
    hit->last_use = ++ctx->weight_clock;
    hit->last_call = ctx->weight_call;
    out->meta = (const int2 *)hit->dev;
    out->frag = (const uint4 *)((const u8 *)hit->dev + hit->meta_bytes);
    out->K = hit->K;
    out->n_tiles = n_tiles;
    out->max_span = hit->max_span;
    return SMR_OK;
}

int flush_mfma_builds(smr_ctx *ctx) {
    // (bands in the layouts of k_ingest_wave — axis 2 / 3 — are built by flush_wave_builds and stay on the list)
    std::vector<smr_ctx::PendingBand> mine, rest;
    for (const auto &p : ctx->pending_bands) (p.axis < 2 ? mine : rest).push_back(p);
    size_t i = 0;
    while (i < mine.size()) {
        WBatch args;
        memset(&args, 0, sizeof(args));
        int tiles = 0;
        for (; i < mine.size() && args.n < MAX_WBUILDS; i++) {
            const smr_ctx::PendingBand &p = mine[i];
            WBuild &b = args.b[args.n++];
            b.scale = p.scale; b.offset = p.offset; b.taps = p.taps; b.n_dst = p.n_dst; b.n_src = p.n_src; b.axis = p.axis; b.K = p.K;
            b.tile0 = tiles; b.meta = (int2 *)p.meta; b.frag = (uint4 *)p.frag;
            tiles += p.n_tiles;
        }
        hipLaunchKernelGGL(k_build_mfma_weights, dim3((unsigned)tiles), dim3(64), 0, ctx->stream, args);
        SMR_HIP(ctx, hipGetLastError());
    }
    ctx->pending_bands = rest;
    return SMR_OK;
}

// ------------------------------------------------------------------ kernel
struct MJob {
    SurfView yp, up, vp;  // planar 4:2:0 source planes (chroma views carry the chroma size)
    SurfView dst;         // RGBA8 tile, dst-sized
    int src_w, src_h;
    // Y'CbCr -> 255 * R'G'B' + 1280.5 with the range expansion and the clamps of planar_yuv_to_rgba.wgsl:45-57 folded in:
    // luma in u8 units clamped to [ylo, yhi], chroma in 1/16 u8 units clamped to [clo, chi]
    float ky, krv, kgu, kgv, kbu, cr, cg, cb;
    float ylo, yhi;
    int clo, chi;
    const int2 *h_meta; const uint4 *h_frag;
    const int2 *v_meta; const uint4 *v_frag;
    int KH, KV, n_htiles, n_vtiles, strips_x;
    int ts;   // T row stride in dwords (= 8 mod 16: conflict-free ds_read_b128 of the A operand)
    int ngm;  // column groups (4 texels) the LDS is sized for
    int RG;   // ring depth in 8-row granules (even)
    int ablate;  // profiling only (SMR_ABLATE): 1 skip convert, 2 skip pass 1, 4 skip pass 2, 8 skip encode + store, 16 skip staging, 32 dispatch only
    // direct output (MDirect below): the layer this tile is blitted by, -1 = none, and its (even) position in the output frame
    int layer, ox, oy;
    int nv12;  // `up` is the interleaved UV plane of an NV12 frame (2 bytes per chroma texel), `vp` aliases it
};

constexpr int MAX_MJOBS_PER_LAUNCH = 16;
struct MArgs {
    MJob jobs[MAX_MJOBS_PER_LAUNCH];
    int unit_prefix[MAX_MJOBS_PER_LAUNCH + 1];  // units = strips_x * n_vtiles per job (strip-major)
    int n_jobs;
    int units_per_block;
    const MDirect *direct;  // device record (rides behind the layout list), nullptr = off
};

constexpr int M_OFF_THR = M_LUT_ENTRIES * 4;
constexpr int M_OFF_T = M_OFF_THR + (SMR_TABLE_FLOATS - 256) * 4;  // thr[257] + pad + encode estimate table
static_assert(M_OFF_THR % 16 == 0 && M_OFF_T % 16 == 0, "T must start on a 16-byte boundary");
__host__ __device__ inline int m_ncd(int ngm) { return ((2 * (ngm - 1) + 3) >> 2) + 2; }  // staged chroma dwords per row
constexpr int M_PIECE_TILES = 64;   // output tiles of one piece (their window table sits in LDS)

// One 4x1 pixel block (m_convert_px, smr_ingest_common.h) -> 3 channels x 16 bytes into T.
template <int ABL>
__device__ __forceinline__ void m_convert_block(const MConv &J, const u32 *__restrict__ lut, u32 yy, u32 ua, u32 ub, u32 va, u32 vb, u32 w13, u32 w31,
                                                u32 *__restrict__ Trow /* T + row * ts + 4g, channel stride 16 * ts */, int ts) {
    uint4 o[3];
    m_convert_px<(ABL & 256) != 0>(J, yy, ua, ub, va, vb, w13, w31, o);  // (the LUT sits at LDS offset 0: launch_mfma checks it)
    if (ABL & 512) {  // profiling: one dword instead of 48 bytes
        *Trow = o[0].x ^ o[0].y ^ o[0].z ^ o[0].w ^ o[1].x ^ o[1].y ^ o[1].z ^ o[1].w ^ o[2].x ^ o[2].y ^ o[2].z ^ o[2].w;
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) *(uint4 *)(Trow + (size_t)c * M_CH * ts) = o[c];
}

// LDS carve (bytes): decode LUT | encode tables | T[2] | ring | raw[2] | tile windows of the piece
struct MLds {
    int t, t_bytes, ring, raw, raw_bytes, vmeta, total;
};
__host__ __device__ inline MLds m_lds(int ts, int ngm, int RG) {
    MLds L;
    L.t = M_OFF_T;
    L.t_bytes = 3 * M_CH * ts * 4;
    L.ring = L.t + 2 * L.t_bytes;
    L.raw = L.ring + 3 * RG * M_SW * 16;
    L.raw_bytes = M_CH * 4 * ngm + 2 * 9 * 4 * m_ncd(ngm);
    L.vmeta = L.raw + 2 * L.raw_bytes;
    L.total = L.vmeta + M_PIECE_TILES * 8;
    return L;
}

// Output tiles [vt0, vt1] (16 rows each) of strip `strip` of job J.  KH_T / KV_T: the k-step counts when the whole launch shares
// them (0 = read them from the job: loops unrolled to the maximum and predicated); ABL: profiling build with phases compiled out.
//
// Every workgroup of a launch is resident at once, so the launch lasts as long as one workgroup's instruction streams.  The
// waves are therefore specialised and run as a two-stage pipeline, one barrier per chunk of 16 source rows:
//   convert waves (4)   chunk k:     land the raw footprint of chunk k + 1, issue the loads of chunk k + 2, convert chunk k
//                                    (one 4x1 block per lane and step) into T[k & 1]            — vector ALU + LDS gathers
//   filter waves (4)    chunk k - 1: pass 1 from T[(k - 1) & 1] into the ring columns of the wave's own N tile, then pass 2 +
//                                    sRGB encode + store of every output tile whose window is complete   — MFMA + LDS reads
// A ring column is written and read by one wave only; T and the raw footprint are double-buffered across the barrier.
template <int KH_T, int KV_T, int ABL>
__device__ __forceinline__ void mfma_piece(const MJob &J, const MDirect *__restrict__ Dg, int strip, int vt0, int vt1, u8 *smem, unsigned long long *dbg) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: keeps per-wave addressing on the scalar unit)
    const bool is_conv = wave >= M_NT;
    constexpr int ablate = ABL;  // compile-time switch set (profiling builds only)
    // profiling build (ABL & 64): shader cycles per phase, wave 0 (filter) -> dbg[0..7], wave M_NT (convert) -> dbg[8..15]
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    const bool timing = (ablate & 64) && dbg && (wave == 0 || wave == M_NT);
    auto mark = [&](int ph) {
        if (timing) {
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): charge LDS latency to the phase that issued it
            const unsigned long long now = __builtin_readcyclecounter();
            tph[ph] += now - tlast;
            tlast = now;
        }
    };
    const int KH = KH_T ? KH_T : J.KH, KV = KV_T ? KV_T : J.KV;
    constexpr int KH_N = KH_T ? KH_T : M_KH_MAX, KV_N = KV_T ? KV_T : M_KV_MAX;
    const int ts = J.ts, RG = J.RG;
    const MLds L = m_lds(ts, J.ngm, RG);
    const u32 *s_lut = (const u32 *)smem;
    const float *s_thr = (const float *)(smem + M_OFF_THR);
    u32 *T = (u32 *)(smem + L.t);
    uint4 *Mh = (uint4 *)(smem + L.ring);
    int2 *s_vmeta = (int2 *)(smem + L.vmeta);  // (window base, last row) of the piece's output tiles
    const int ys = 4 * J.ngm, cs = 4 * m_ncd(J.ngm);

    // ---- strip geometry
    const int nt0 = strip * M_NT, ntn = min(M_NT, J.n_htiles - nt0);
    const int cbase = J.h_meta[nt0].x & ~7;  // luma column of T column 0 (chroma staging wants it = 0 mod 8)
    // columns converted per chunk: up to the last texel that carries a weight.  A K window may run past them — into the row
    // padding, the next row, the other T buffer or the ring: all finite, all under zero weights.
    const int ngroups = min((J.h_meta[nt0 + ntn - 1].y - cbase + 4) >> 2, J.ngm);
    const int R_lo = J.v_meta[vt0].x, R_hi = J.v_meta[vt1].y;
    const int n_chunks = (R_hi - R_lo) / M_CH + 1;
    if (vt0 + tid <= vt1) s_vmeta[tid] = J.v_meta[vt0 + tid];
    __syncthreads();  // tables, zeroed ring / T padding and the tile windows are in place

    if (is_conv) {
        // ================================================================== convert waves
        const int cwave = wave - M_NT, ctid = tid - M_NT * 64;
        constexpr int CW = M_WAVES - M_NT, CT = CW * 64;
        const MConv K = {J.ky, J.krv, J.kgu, J.kgv, J.kbu, J.cr, J.cg, J.cb, J.ylo, J.yhi, (float)J.clo, (float)J.chi};
        const u8 *const y_ptr = J.yp.ptr, *const u_ptr = J.up.ptr, *const v_ptr = J.vp.ptr;
        const u32 y_pitch = J.yp.pitch, u_pitch = J.up.pitch, v_pitch = J.vp.pitch;
        const int sw = J.src_w, sh = J.src_h, cw = J.up.w, chh = J.up.h;
        const int ncd = m_ncd(ngroups);
        // staging: luma = one row per wave and step (64 lanes x 4 B), chroma = 18 (plane, row) tasks of <= 64 dwords.
        // Every lane always loads (dead lanes re-read the last live dword: same cache line, no extra traffic).
        constexpr int NY = M_CH / CW, NC = (18 + CW - 1) / CW;
        constexpr bool NV = (ABL & 4096) != 0;  // the build that also reads NV12 frames (decoder hand-off: Y + interleaved UV)
        const bool nv = NV && J.nv12 != 0;      // (uniform)
        u32 py[NY], pc[NC], pc_hi[NV ? NC : 1];
        const int sw4 = (sw + 3) & ~3;
        const bool y_live = lane < ngroups && cbase + 4 * lane < sw4;
        const bool c_live = lane < ncd;
        const int c_col0 = (cbase >> 1) - 4 + 4 * lane;                 // first chroma column of this lane's staged dword
        const int c_col0c = clampi(c_col0, 0, (cw - 1) & ~3);           // ... of the dword actually loaded (clamp-to-edge)
        const bool c_edge = c_live && (c_col0 < 0 || c_col0 + 3 > cw - 1);
        // (uniform row base in scalar registers + a 32-bit lane offset: the loads need no vector address arithmetic)
        const u32 y_off = (u32)min(cbase + 4 * lane, sw4 - 4), c_off = (u32)c_col0c;
        auto issue = [&](int base) {
            if (ablate & 1024) base = R_lo;  // profiling: always the same rows (cache / TLB hits)
#pragma unroll
            for (int k = 0; k < NY; k++) {
                const u8 *rowp = y_ptr + (size_t)clampi(base + cwave + CW * k, 0, sh - 1) * y_pitch;
                py[k] = *(const u32 *)(rowp + y_off);
            }
            const int i0 = (base - 1) >> 1;  // chroma row of the chunk's first row pair
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int rt = min(cwave + CW * k, 17);  // (plane, row) task, uniform per wave; tasks past the 18th repeat the last one
                const int plane = rt >= 9 ? 1 : 0, r = rt - 9 * plane;
                const u8 *rowp = (plane ? v_ptr : u_ptr) + (size_t)clampi(i0 + r, 0, chh - 1) * (plane ? v_pitch : u_pitch);
                if (nv) {  // four chroma texels = eight interleaved bytes; the plane's four are picked when they land
                    pc[k] = *(const u32 *)(rowp + 2 * c_off);
                    pc_hi[NV ? k : 0] = *(const u32 *)(rowp + 2 * c_off + 4);
                } else {
                    pc[k] = *(const u32 *)(rowp + c_off);
                }
            }
        };
        auto land = [&](u8 *raw) {
            u8 *rawY = raw, *rawU = rawY + M_CH * ys, *rawV = rawU + 9 * cs;
            if (y_live) {
#pragma unroll
                for (int k = 0; k < NY; k++) *(u32 *)(rawY + (cwave + CW * k) * ys + 4 * lane) = (ablate & 128) ? 0x80808080u : py[k];
            }
            if (nv) {
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    const int rt = min(cwave + CW * k, 17);
                    pc[k] = __builtin_amdgcn_perm(pc_hi[NV ? k : 0], pc[k], rt >= 9 ? 0x07050301u : 0x06040200u);  // V : U bytes
                }
            }
            if (c_edge) {
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    u32 o = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) o |= ((pc[k] >> (8 * (clampi(c_col0 + b, 0, cw - 1) - c_col0c))) & 0xffu) << (8 * b);
                    pc[k] = o;
                }
            }
            if (c_live) {
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    const int rt = cwave + CW * k;
                    if (rt < 18) {
                        const int plane = rt >= 9 ? 1 : 0, r = rt - 9 * plane;
                        *(u32 *)((plane ? rawV : rawU) + r * cs + 4 * lane) = (ablate & 128) ? 0x80808080u : pc[k];
                    }
                }
            }
        };
        // convert tasks of this thread: (chunk row, column group g) for id = ctid + k * CT, 16 * ngroups <= 1024 of them.
        // Row 0 of a chunk is an odd luma row: 3/4 of chroma row p = row / 2 (weights A); odd chunk rows take 3/4 of row p + 1 (B).
        constexpr u32 WA13 = 0x03010903u, WA31 = 0x01030309u, WB13 = 0x09030301u, WB31 = 0x03090103u;
        constexpr int NCV = 1024 / CT;
        int crow[NCV], cgrp[NCV];
#pragma unroll
        for (int k = 0; k < NCV; k++) {
            crow[k] = (ctid + k * CT) / ngroups;
            cgrp[k] = ctid + k * CT - crow[k] * ngroups;
        }
        u8 *const raw0 = smem + L.raw;
        if (!(ablate & 16)) {
            issue(R_lo);
            land(raw0);
            if (n_chunks > 1) issue(R_lo + M_CH);
        }
        __syncthreads();  // (pairs with the filter waves' prologue barrier) chunk 0's footprint is visible to every convert wave
        if (timing) tlast = __builtin_readcyclecounter();
        for (int k = 0; k <= n_chunks; k++) {
            mark(0);
            if (k < n_chunks) {
                // the one wait for memory of the chunk: the footprint of chunk k + 1, issued a whole chunk ago
                __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
                mark(1);
                if (k + 1 < n_chunks && !(ablate & 16)) land(raw0 + ((k + 1) & 1) * L.raw_bytes);
                mark(2);
                if (k + 2 < n_chunks && !(ablate & 48)) issue(R_lo + (k + 2) * M_CH);
                mark(3);
                const u8 *rawY = raw0 + (k & 1) * L.raw_bytes, *rawU = rawY + M_CH * ys, *rawV = rawU + 9 * cs;
                u32 *Tk = T + (size_t)(k & 1) * (L.t_bytes >> 2);
#pragma unroll
                for (int it = 0; it < NCV; it++) {
                    const int row = crow[it], g = cgrp[it];
                    if (row >= M_CH || (ablate & 1)) break;
                    const int p = row >> 1;
                    const u32 yy = *((const u32 *)(rawY + row * ys) + g);
                    const int bi = 2 * g + 3;
                    const u32 *ru = (const u32 *)(rawU + p * cs) + (bi >> 2), *rv = (const u32 *)(rawV + p * cs) + (bi >> 2);
                    const u32 shb = (u32)(bi & 3);
                    const u32 ua = __builtin_amdgcn_alignbyte(ru[1], ru[0], shb), ub = __builtin_amdgcn_alignbyte(ru[(cs >> 2) + 1], ru[cs >> 2], shb);
                    const u32 va = __builtin_amdgcn_alignbyte(rv[1], rv[0], shb), vb = __builtin_amdgcn_alignbyte(rv[(cs >> 2) + 1], rv[cs >> 2], shb);
                    m_convert_block<ABL>(K, s_lut, yy, ua, ub, va, vb, (row & 1) ? WB13 : WA13, (row & 1) ? WB31 : WA31, Tk + (size_t)row * ts + 4 * g, ts);
                    mark(4 + (it > 0));
                }
            }
            mark(6);
            __syncthreads();  // T[k & 1] and the raw footprint of chunk k + 1 are complete
        }
        if (timing && lane == 0)
            for (int i = 0; i < 8; i++) atomicAdd(dbg + 8 + i, tph[i]);
        return;
    }

    // ====================================================================== filter waves: N tile `wave` of the strip
    const int l16 = lane & 15, lq = lane >> 4;
    const bool wave_on = wave < ntn;
    const int my_tile = nt0 + (wave_on ? wave : 0);
    const int colw = J.h_meta[my_tile].x - cbase;  // T column (dword) of this wave's K window
    uint4 bh[KH_N], bh2[KH_N];  // (w_hi, w_hi) and (w_lo, 0) per texel pair
#pragma unroll
    for (int j = 0; j < KH_N; j++)
        if (j < KH) {
            bh[j] = J.h_frag[((size_t)my_tile * 2 * KH + j) * 64 + lane];
            bh2[j] = J.h_frag[((size_t)my_tile * 2 * KH + KH + j) * 64 + lane];
        }
    u8 *const d_ptr = J.dst.ptr;
    const u32 d_pitch = J.dst.pitch;
    const int d_w = J.dst.w, d_h = J.dst.h;
    const uint4 *const v_frag = J.v_frag;
    const int tx0 = strip * M_SW + 16 * wave;
    int vt = vt0;
    int cg = 0;  // ring granule of the chunk being written
    // pass-2 weights of the next output tile, fetched a chunk ahead (they depend on the tile row only)
    // (pass 2 takes the hi term of its weights only: its input rows are exact f16 and its result is rounded once, to 8 bits —
    //  on white noise the lo term moved 0.1 % of the bytes, the lo term of pass 1, whose result is rounded to f16, 1.4 %)
    uint4 bv[KV_N];
    // direct output: the class of the 128x16 output tile this lane's four pixels of the next tile row fall into
    constexpr bool DIRECT = (ABL & 2048) != 0;  // the direct-output build (the plain one carries none of its registers)
    const bool dj = DIRECT && Dg != nullptr && J.layer >= 0;  // (uniform)
    const int d_ox = J.ox, d_oy = J.oy, d_layer = J.layer;
    u32 cls_next = 0xffu;
    auto fetch_bv = [&](int t) {
#pragma unroll
        for (int j = 0; j < KV_N; j++)
            if (j < KV) bv[j] = v_frag[((size_t)t * 2 * KV + j) * 64 + lane];
        if (dj) {
            const MDirect *Dp = Dg;
            asm volatile("" : "+s"(Dp));  // read the record where it is used: it must not sit in scalar registers for the whole piece
            const int X = d_ox + tx0 + 4 * lq, Y = d_oy + 16 * t + l16;
            const bool in = 16 * t + l16 < d_h && tx0 + 4 * lq < d_w && X >= 0 && Y >= 0 && X < Dp->yp.w && Y < Dp->yp.h;
            cls_next = Dp->cls[in ? (Y >> 4) * Dp->tiles_x + (X >> 7) : 0];
            cls_next = in ? cls_next : 0xffu;
        }
    };
    fetch_bv(vt0);
    __syncthreads();  // (pairs with the convert waves' prologue barrier)
    if (timing) tlast = __builtin_readcyclecounter();
    for (int k = 0; k <= n_chunks; k++) {
        mark(0);
        if (k >= 1 && wave_on) {
            const int base = R_lo + (k - 1) * M_CH;
            if (!(ablate & 2)) {
                // ---- pass 1 of chunk k - 1: its 16 rows x this wave's 16 output columns, three channels
                f32x4 acc[3];
#pragma unroll
                for (int c = 0; c < 3; c++) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const u32 *Ta = T + (size_t)((k - 1) & 1) * (L.t_bytes >> 2) + (size_t)l16 * ts + colw + 4 * lq;
#pragma unroll
                for (int j = 0; j < KH_N; j++) {
                    if (j < KH) {
                        uint4 a[3];
#pragma unroll
                        for (int c = 0; c < 3; c++) a[c] = *(const uint4 *)(Ta + (size_t)c * M_CH * ts + 16 * j);
                        // (two other channels' MFMAs between two on the same accumulator)
#pragma unroll
                        for (int c = 0; c < 3; c++)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[c]), __builtin_bit_cast(f16x8, bh[j]), acc[c], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < 3; c++)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[c]), __builtin_bit_cast(f16x8, bh2[j]), acc[c], 0, 0, 0);
                    }
                }
                mark(1);
                // lane holds rows 4 lq .. 4 lq + 3 of column l16: one half granule of the ring
                int rgw = cg + (lq >> 1);
                rgw -= rgw >= RG ? RG : 0;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const __half2 h0 = __floats2half2_rn(acc[c][0], acc[c][1]), h1 = __floats2half2_rn(acc[c][2], acc[c][3]);
                    *(uint2 *)((u8 *)(Mh + (size_t)(c * RG + rgw) * M_SW + 16 * wave + l16) + 8 * (lq & 1)) = make_uint2(*(const u32 *)&h0, *(const u32 *)&h1);
                }
            }
            mark(2);
            cg += 2;
            cg -= cg >= RG ? RG : 0;
            // ---- pass 2 + encode + store of every output tile whose window ends inside chunk k - 1
            const int e = base + M_CH - 1;
            while (vt <= vt1) {
                const int2 vm = s_vmeta[vt - vt0];
                if (vm.y > e) break;
                if (!(ablate & 4)) {
                    const int g0 = ((vm.x - R_lo) >> 3) % RG;
                    f32x4 acc[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < KV_N; j++) {
                        if (j < KV) {
                            int rg = g0 + 4 * j + lq;
                            rg -= rg >= RG ? RG : 0;
                            rg -= rg >= RG ? RG : 0;
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                const uint4 a = Mh[(size_t)(c * RG + rg) * M_SW + 16 * wave + l16];
                                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bv[j]), acc[c], 0, 0, 0);
                            }
                        }
                    }
                    // the next tile's weights, before the stores below enter the memory queue: on gfx950 a wait for a load drains
                    // the stores issued before it too (one counter)
                    mark(3);
                    const bool direct = cls_next == (u32)d_layer;  // (false everywhere unless dj: d_layer >= 0 never equals 0xff.. see host)
                    fetch_bv(min(vt + 1, vt1));
                    // lane holds columns tx0 + 4 lq .. + 3 of output row 16 vt + l16
                    const int y = 16 * vt + l16, x = tx0 + 4 * lq;
                    if (ablate & 8) {  // profiling: no encode, one dword store keeps the MFMAs alive
                        if (y < d_h && x < d_w) *(float *)(d_ptr + (size_t)y * d_pitch + (size_t)x * 4) = acc[0][0] + acc[1][1] + acc[2][2] + acc[0][3];
                    } else {
                        u32 px[4];
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            px[i] = srgb_encode8(acc[0][i], s_thr) | (srgb_encode8(acc[1][i], s_thr) << 8) | (srgb_encode8(acc[2][i], s_thr) << 16) | 0xff000000u;
                        if (dj) {
                            // (m_direct_yuv: the arithmetic of k_compose_output's copy tiles on the bytes above; every lane takes part
                            //  in its lane swaps, the lanes of a direct tile store)
                            const bool odd = (l16 & 1) != 0;
                            u32 mine, other;
                            const u32 yq = m_direct_yuv(px, odd, &mine, &other);
                            if (direct) {
                                const MDirect *Dp = Dg;
                                asm volatile("" : "+s"(Dp));
                                m_direct_store(Dp, d_ox + x, d_oy + y, odd, yq, mine, other);
                            }
                        }
                        if (!direct && y < d_h && x < d_w) {
                            u8 *o = d_ptr + (size_t)y * d_pitch + (size_t)x * 4;
                            if (x + 3 < d_w) {
                                *(uint4 *)o = make_uint4(px[0], px[1], px[2], px[3]);
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; i++)
                                    if (x + i < d_w) ((u32 *)o)[i] = px[i];
                            }
                        }
                    }
                    mark(4);
                }
                vt++;
            }
        }
        mark(5);
        __syncthreads();  // T[k & 1] (chunk k) is complete; this wave is done with T[(k - 1) & 1]
    }
    if (timing && lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(dbg + i, tph[i]);
}

template <int KH_T, int KV_T, int ABL>
// (the generic build keeps up to 6 + 4 k-steps of weights in registers: one workgroup per CU, 168 registers)
__global__ __launch_bounds__(M_THREADS, KH_T ? M_WAVES / 2 : M_WAVES / 4) void k_ingest_mfma(const MArgs args, const float *__restrict__ tables, const u32 *__restrict__ lut,
                                                              unsigned long long *dbg) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const int tid = threadIdx.x;
    if (ABL & 32) return;
    // tables: (hi | lo << 16) decode LUT, encode thresholds + estimate table
    for (int i = tid; i < M_LUT_ENTRIES; i += M_THREADS) ((u32 *)smem)[i] = lut[min(max(i - 256, 0), 255)];
    for (int i = tid; i < SMR_TABLE_FLOATS - 256; i += M_THREADS) ((float *)(smem + M_OFF_THR))[i] = tables[256 + i];
    const int total = args.unit_prefix[args.n_jobs];
    // XCD-aware order (as k_ingest_resample): ids that share an XCD are neighbours in the unit space
    const int per_xcd = (int)gridDim.x >> 3;
    const int v = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    int u = v * args.units_per_block;
    const int u_end = min(u + args.units_per_block, total);
    bool first = true;
    while (u < u_end) {
        int j = 0;
        while (j + 1 < args.n_jobs && args.unit_prefix[j + 1] <= u) j++;
        const MJob &J = args.jobs[j];
        const int local = u - args.unit_prefix[j];
        const int strip = local / J.n_vtiles, vt0 = local - strip * J.n_vtiles;
        const int vt1 = min(min(J.n_vtiles, vt0 + (u_end - u)), vt0 + M_PIECE_TILES) - 1;
        if (!first) __syncthreads();
        {
            // T (row padding included) and the ring must hold finite values wherever a zero weight meets them
            const MLds L = m_lds(J.ts, J.ngm, J.RG);
            u32 *z = (u32 *)(smem + L.t);
            for (int i = tid; i < (L.raw - L.t) / 4; i += M_THREADS) z[i] = 0u;
        }
        mfma_piece<KH_T, KV_T, ABL>(J, args.direct, strip, vt0, vt1, smem, dbg);
        u += vt1 - vt0 + 1;
        first = false;
    }
}

// ------------------------------------------------------------------ host side
bool mfma_plane_ok(const SurfView &p, u32 bytes) { return (p.pitch % 4) == 0 && (((uintptr_t)p.ptr) % 4) == 0 && p.pitch >= ((bytes + 3u) & ~3u); }

// What k_ingest_mfma covers: planar 4:2:0 (limited or full range) with even luma size and dword-aligned planes, separable
// plan, horizontal pass first, no box pre-reduction, 16-byte aligned tile rows, footprints that fit the LDS.
bool can_fuse_mfma(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile) {
    if (ctx->ingest_impl == SMR_INGEST_VALU_F32 || ctx->ingest_impl == SMR_INGEST_MFMA_F16_NODE) return false;
    const bool nv12 = f && f->format == SMR_FRAME_NV12;
    if (!f || !f->planes[0] || !f->planes[1] || (!nv12 && !f->planes[2])) return false;
#ifdef SMR_ABLATION_BUILDS
    if (nv12) return false;
#endif
    if (f->format != SMR_FRAME_PLANAR_YUV420 && f->format != SMR_FRAME_PLANAR_YUVJ420 && !nv12) return false;
    if (f->width % 2 || f->height % 2 || f->width < 8 || f->height < 2) return false;
    // Two filtered axes, horizontal pass first, no box pre-reduction.  (A plan the reference orders vertically first comes back
    // here on the transposed frame: make_mfma_job_transposed.  Running it with the passes swapped instead was measured: <= 1 LSB
    // and 99.6 % identical on camera-like content, but 2 LSB on white noise.)
    if (!(plan.kind == 2 && plan.levels[0] == 0 && plan.levels[1] == 0 && plan.axis[0] == 0 && plan.axis[1] == 1)) return false;
    const int hs = 0, vs = 1;  // plan slots of the horizontal / vertical pass
    if (!mfma_plane_ok(view_of(f->planes[0]), f->width)) return false;
    if (nv12) {  // (the last staged dword pair may start up to 3 texels before the row's end: 8 bytes must be readable there)
        const SurfView uv = view_of(f->planes[1]);
        if ((uv.pitch % 4) || (((uintptr_t)uv.ptr) % 4) || uv.pitch < ((f->width + 7u) & ~7u)) return false;
    } else if (!mfma_plane_ok(view_of(f->planes[1]), f->width / 2) || !mfma_plane_ok(view_of(f->planes[2]), f->width / 2)) {
        return false;
    }
    if (((uintptr_t)tile->ptr % 16) || (tile->pitch % 16)) return false;
    int KH, KV, sh_, sv_;
    if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[hs], plan.offset[hs], (int)tile->w, (int)f->width, 0)) KH = t->K;
    else mfma_band_geometry(plan.scale[hs], plan.offset[hs], (int)tile->w, (int)f->width, 0, &KH, &sh_);
    if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[vs], plan.offset[vs], (int)tile->h, (int)f->height, 1)) KV = t->K;
    else mfma_band_geometry(plan.scale[vs], plan.offset[vs], (int)tile->h, (int)f->height, 1, &KV, &sv_);
    if (KH > M_KH_MAX || KV > M_KV_MAX) return false;
    return true;
}

int make_mfma_job(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile, MJob *out, bool *fits) {
    MfmaBand bh, bv;
    const int hs = plan.axis[0] == 0 ? 0 : 1, vs = 1 - hs;  // plan slots of the horizontal / vertical pass (can_fuse_mfma)
    int rc = get_mfma_band(ctx, plan.scale[hs], plan.offset[hs], (int)tile->w, (int)f->width, 0, &bh);
    if (rc != SMR_OK) return rc;
    rc = get_mfma_band(ctx, plan.scale[vs], plan.offset[vs], (int)tile->h, (int)f->height, 1, &bv);
    if (rc != SMR_OK) return rc;
    MJob &J = *out;
    J.yp = view_of(f->planes[0]); J.up = view_of(f->planes[1]); J.vp = f->planes[2] ? view_of(f->planes[2]) : J.up;
    J.dst = view_of(tile);
    J.src_w = (int)f->width; J.src_h = (int)f->height;
    // planar_yuv_to_rgba.wgsl:45-57 with every constant folded; chroma arrives in 1/16 u8 units (16 * 255 * u)
    const MConv C = m_conv_constants(f->format == SMR_FRAME_PLANAR_YUVJ420);
    J.ky = C.ky; J.krv = C.krv; J.kgu = C.kgu; J.kgv = C.kgv; J.kbu = C.kbu; J.cr = C.cr; J.cg = C.cg; J.cb = C.cb;
    J.ylo = C.ylo; J.yhi = C.yhi; J.clo = (int)C.clo; J.chi = (int)C.chi;
    J.h_meta = bh.meta; J.h_frag = bh.frag; J.KH = bh.K; J.n_htiles = bh.n_tiles;
    J.v_meta = bv.meta; J.v_frag = bv.frag; J.KV = bv.K; J.n_vtiles = bv.n_tiles;
    J.strips_x = (bh.n_tiles + M_NT - 1) / M_NT;
    J.ablate = ctx->ablate;
    J.layer = -1; J.ox = 0; J.oy = 0;
    J.nv12 = f->format == SMR_FRAME_NV12 ? 1 : 0;
    if (J.nv12) J.vp = J.up;
    // LDS sizing: the widest strip footprint (host twin of the kernel's geometry), kept with the horizontal band
    smr_ctx::MfmaTable *th = find_mfma_table(ctx, plan.scale[hs], plan.offset[hs], (int)tile->w, (int)f->width, 0);
    const int taps_h = host_taps(plan.scale[hs]);
    int ngm = th && th->ngm > 0 ? th->ngm : 1;
    for (int s = 0; s < J.strips_x && !(th && th->ngm > 0); s++) {
        const int t0 = s * M_NT, t1 = (t0 + M_NT < bh.n_tiles ? t0 + M_NT : bh.n_tiles) - 1;
        int lo = lanczos_first(16 * t0, plan.scale[hs], plan.offset[hs]);
        lo = lo < 0 ? 0 : (lo > J.src_w - 1 ? J.src_w - 1 : lo);
        const int o1 = 16 * t1 + 15 < (int)tile->w - 1 ? 16 * t1 + 15 : (int)tile->w - 1;
        int hi = lanczos_first(o1, plan.scale[hs], plan.offset[hs]) + taps_h - 1;
        hi = hi < 0 ? 0 : (hi > J.src_w - 1 ? J.src_w - 1 : hi);
        const int g = (hi - (mfma_window_base(lo, 0) & ~7) + 4) >> 2;
        ngm = g > ngm ? g : ngm;
    }
    if (th) th->ngm = ngm;
    J.ngm = ngm;
    J.ts = ((4 * ngm + 7) & ~15) + 8;  // >= 4 * ngm and = 8 mod 16
    if (J.ts < 4 * ngm) J.ts += 16;
    // ring: the widest window plus the chunk that may land before the window's tile is resolved
    int rg = (bv.max_span + M_CH - 1 + 7) / 8;
    J.RG = rg < 2 ? 2 : rg;
    *fits = ngm <= M_NG_MAX && (size_t)m_lds(J.ts, J.ngm, J.RG).total <= 160 * 1024;
    return SMR_OK;
}

// ------------------------------------------------------------------ vertical-first plans: the same kernel on the transposed problem
// The reference filters the axis with the stronger shrink first (resampler.rs:123-145); for aspect-preserving fits that order hangs
// on the rounding of the tile size, so a tile that resizes flips between the two from frame to frame.  The kernel above filters
// horizontally first.  Filtering an image vertically first is filtering its transpose horizontally first — the colour conversion is
// per pixel, the chroma up-sampling weights are the same along both axes, every tap loop runs along one axis — so a vertical-first
// plan is run as: transpose the planes (3.1 MB for 1080p), the horizontal-first kernel on the transposed frame into a transposed
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

bool can_fuse_mfma(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile);
int make_mfma_job(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile, MJob *out, bool *fits);

// A job for a vertical-first plan (see above).  `slot0`: four surface-cache slots of the caller's for the transposed planes and tile.
// *ok = false: not a case for this route (the caller goes on to the general kernels).  The plane transposes are enqueued here; the
// caller launches the job with its others and then runs launch_transpose<u32> on every MTransposeBack.
int make_mfma_job_transposed(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, smr_surface *tile, size_t slot0, MJob *out, bool *ok,
                             MTransposeBack *back) {
    *ok = false;
    if (ctx->ingest_impl == SMR_INGEST_VALU_F32 || ctx->ingest_impl == SMR_INGEST_MFMA_F16_NODE || !f || !f->planes[0] || !f->planes[1]) return SMR_OK;
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
    if (!can_fuse_mfma(ctx, &ft, pt, tile_t)) return SMR_OK;
    bool fits = false;
    if (int rc = make_mfma_job(ctx, &ft, pt, tile_t, out, &fits)) return rc;
    if (!fits) return SMR_OK;
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

// the scale range of the benchmark scenes (1.5x .. 2x: 3 k-steps per pass-1 tile, 2 per pass-2 tile) gets its own build
typedef void (*MfmaKernel)(const MArgs, const float *, const u32 *, unsigned long long *);
#ifdef SMR_ABLATION_BUILDS  // tools/ablate_mfma.sh: the (3,2) build with phases compiled out (1 convert, 2 pass 1, 4 pass 2, 16 staging, 64 phase timers)
constexpr int M_ABL[] = {0, 0, 1, 2, 4, 5, 6, 7, 16, 23, 64, 128, 262, 518, 774, 1024, 1088, 3, 9, 13, 8, 256, 512, 768, 39};
constexpr MfmaKernel M_KERNELS[] = {k_ingest_mfma<0, 0, 0>,  k_ingest_mfma<3, 2, 0>,  k_ingest_mfma<3, 2, 1>,  k_ingest_mfma<3, 2, 2>,
                                    k_ingest_mfma<3, 2, 4>,  k_ingest_mfma<3, 2, 5>,  k_ingest_mfma<3, 2, 6>,  k_ingest_mfma<3, 2, 7>,
                                    k_ingest_mfma<3, 2, 16>, k_ingest_mfma<3, 2, 23>, k_ingest_mfma<3, 2, 64>, k_ingest_mfma<3, 2, 128>,
                                    k_ingest_mfma<3, 2, 262>, k_ingest_mfma<3, 2, 518>, k_ingest_mfma<3, 2, 774>,
                                    k_ingest_mfma<3, 2, 1024>, k_ingest_mfma<3, 2, 1088>, k_ingest_mfma<3, 2, 3>, k_ingest_mfma<3, 2, 9>, k_ingest_mfma<3, 2, 13>,
                                    k_ingest_mfma<3, 2, 8>, k_ingest_mfma<3, 2, 256>, k_ingest_mfma<3, 2, 512>, k_ingest_mfma<3, 2, 768>, k_ingest_mfma<3, 2, 39>};
#else
// builds: plain | direct output (2048) | NV12-capable (4096) | both
constexpr int M_ABL[] = {0, 0, 2048, 2048, 4096, 4096, 6144, 6144};
constexpr MfmaKernel M_KERNELS[] = {k_ingest_mfma<0, 0, 0>,    k_ingest_mfma<3, 2, 0>,    k_ingest_mfma<0, 0, 2048>, k_ingest_mfma<3, 2, 2048>,
                                    k_ingest_mfma<0, 0, 4096>, k_ingest_mfma<3, 2, 4096>, k_ingest_mfma<0, 0, 6144>, k_ingest_mfma<3, 2, 6144>};
#endif
constexpr int M_NKERNELS = (int)(sizeof(M_KERNELS) / sizeof(M_KERNELS[0]));

int launch_mfma(smr_ctx *ctx, std::vector<MJob> &jobs, const MDirect *direct = nullptr) {
    if (!ctx->mfma_attr_set) {  // per device, hence per ctx
        for (MfmaKernel k : M_KERNELS) {
            SMR_HIP(ctx, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipFuncAttributes fa;
            SMR_HIP(ctx, hipFuncGetAttributes(&fa, (const void *)k));
            // the convert waves address the decode LUT by its LDS offset 0: holds while the kernel declares no static LDS
            if (fa.sharedSizeBytes != 0) return smr_fail(ctx, SMR_ERR_INTERNAL, "k_ingest_mfma: %zu B of static LDS in front of the dynamic segment", (size_t)fa.sharedSizeBytes);
        }
        ctx->mfma_attr_set = true;
    }
    if (int rc = flush_mfma_builds(ctx)) return rc;
    StageScope scope(ctx, SMR_STAGE_FUSED_INGEST);
    ctx->kernel_launches[SMR_KERNEL_INGEST_MFMA_WG]++;
    for (size_t j0 = 0; j0 < jobs.size(); j0 += MAX_MJOBS_PER_LAUNCH) {
        const size_t nj = jobs.size() - j0 < (size_t)MAX_MJOBS_PER_LAUNCH ? jobs.size() - j0 : (size_t)MAX_MJOBS_PER_LAUNCH;
        MArgs args;
        memset(&args, 0, sizeof(args));
        size_t lds = 0;
        int total = 0;
        for (size_t j = 0; j < nj; j++) {
            const MJob &J = jobs[j0 + j];
            args.jobs[j] = J;
            args.unit_prefix[j] = total;
            total += J.strips_x * J.n_vtiles;
            const size_t b = (size_t)m_lds(J.ts, J.ngm, J.RG).total;
            lds = b > lds ? b : lds;
        }
        args.unit_prefix[nj] = total;
        args.n_jobs = (int)nj;
        args.direct = direct;
        bool all32 = true;
        for (size_t j = 0; j < nj; j++) all32 = all32 && args.jobs[j].KH == 3 && args.jobs[j].KV == 2;
        int ki = all32 ? 1 : 0;
#ifdef SMR_ABLATION_BUILDS
        for (int i = 2; i < M_NKERNELS; i++)
            if (all32 && ctx->ablate == M_ABL[i]) ki = i;
        if (direct) return smr_fail(ctx, SMR_ERR_INTERNAL, "direct output is not part of the ablation builds");
#else
        if (direct) ki += 2;
        for (size_t j = 0; j < nj; j++)
            if (args.jobs[j].nv12) { ki += 4; break; }
#endif
        const MfmaKernel kern = M_KERNELS[ki];
        // as many workgroups as are resident at once (LDS and registers), minus the share left to the other stream's compose kernel
        int per_cu = 0;
        for (auto &o : ctx->mfma_occupancy)
            if (o.kernel == ki && o.lds == lds) per_cu = o.per_cu;
        if (!per_cu) {
            SMR_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, M_THREADS, lds));
            if (per_cu < 1) per_cu = 1;
            ctx->mfma_occupancy.push_back({ki, lds, per_cu});
        }
        const int reserve = ctx->ingest_reserve_cus >= 0 && ctx->ingest_reserve_cus < ctx->cu_count ? ctx->ingest_reserve_cus : ctx->cu_count / 16;
        const int wg_cap = ctx->ingest_wg_per_cu > 0 ? ctx->ingest_wg_per_cu : 4;
        int blocks = (per_cu > wg_cap ? wg_cap : per_cu) * (ctx->cu_count - reserve);
        int upb = (total + blocks - 1) / blocks;
        if (upb < 2) upb = 2;  // a piece re-converts the rows of its vertical halo
        blocks = ((total + upb - 1) / upb + 7) & ~7;
        args.units_per_block = upb;
        if (ctx->debug_ingest)
            fprintf(stderr, "k_ingest_mfma: %zu jobs, lds %zu B (%d/CU), blocks %d x %d tiles, job0: KH %d KV %d ts %d ngm %d RG %d strips %d vtiles %d\n", nj, lds,
                    per_cu, blocks, upb, args.jobs[0].KH, args.jobs[0].KV, args.jobs[0].ts, args.jobs[0].ngm, args.jobs[0].RG, args.jobs[0].strips_x,
                    args.jobs[0].n_vtiles);
        unsigned long long *dbg = nullptr;
        if (M_ABL[ki] & 64) {  // profiling: per-phase cycle sums of wave 0 of every block, printed after the launch
            dbg = (unsigned long long *)smr_scratch(ctx, 7, 128);
            if (dbg) SMR_HIP(ctx, hipMemsetAsync(dbg, 0, 128, ctx->stream));
        }
        if (blocks > 0) hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(M_THREADS), lds, ctx->stream, args, ctx->d_tables, ctx->d_lut16, dbg);
        if (dbg) {
            unsigned long long h[16];
            SMR_HIP(ctx, hipMemcpyAsync(h, dbg, 128, hipMemcpyDeviceToHost, ctx->stream));
            SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const double b = 1e3 * blocks;
            fprintf(stderr, "k_ingest_mfma kcycles per block: filter: barrier %.1f pass1-mfma %.1f ring-write %.1f pass2-mfma %.1f encode+store %.1f tail %.1f | convert: barrier %.1f vmwait %.1f land %.1f issue %.1f task0 %.1f task1+ %.1f tail %.1f\n",
                    h[0] / b, h[1] / b, h[2] / b, h[3] / b, h[4] / b, h[5] / b, h[8] / b, h[9] / b, h[10] / b, h[11] / b, h[12] / b, h[13] / b, h[14] / b);
        }
        SMR_HIP(ctx, hipGetLastError());
    }
    return SMR_OK;
}

}  // namespace
