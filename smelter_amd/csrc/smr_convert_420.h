// smr_convert_420.h — the input converter of 4:2:0 frames (planar, NV12; limited or full range), second generation: a thread owns a
// 4 x 4 pixel block.  Included by smr_convert.hip (k_yuv420_to_rgba) and, for the CPU, by tests/emu/emu_convert.cpp (the very
// source, one call per block: bit for bit against the oracle without a GPU).
//
// What it computes is planar_yuv_to_rgba.wgsl:35-58 / nv12_to_rgba.wgsl:26-48 as the oracle restates them (oracle/smr_oracle.c
// sample_plane_bilinear + yuv_to_rgb_store), value for value — the same f32 operations in the same order, so the node texture's bytes
// are the reference's bytes — with the work that several pixels share done once:
//   * the sample position of luma column x in chroma texels is x / 2 - 1/4, so the 8-bit sub-texel weights are exactly 1/4 and 3/4
//     (k_yuv_to_rgba_batch's header, smr_convert.hip); a product with 1/4 is exact, hence a * w0 + b * w1 with its two roundings is
//     fma(x, 1/4, RN(y * 3/4)) — one multiply and one fused multiply-add whose single rounding is the sum's;
//   * a block of luma rows 4 P .. 4 P + 3 needs chroma rows 2 P - 1 .. 2 P + 2 (clamped like the sampler clamps): four rows of four
//     bytes per plane become unorm values once (byte / 255 as unorm_of_byte: the IEEE quotient), their horizontal lerps at the four
//     luma columns share the two 3/4 products of a row (2 multiplies + 4 FMAs), and the vertical lerps of two luma rows share the
//     3/4 product of the chroma row between them;
//   * the luma byte's range expansion (y - 16/255) / 0.8588 (correctly rounded division, clamp) is a function of the byte: a 256-entry
//     table in LDS, built per workgroup with the operations themselves;
//   * the two chroma range divisions are a multiply and a fused multiply-add with a two-term reciprocal — the IEEE quotient on the whole
//     domain, checked exhaustively (below); the clamps ride on the instructions that produce their operands.
// ~40 vector instructions per pixel against k_yuv_to_rgba_batch's 70.
//
// Output: the RGBA8 node texture (16 bytes per thread and row) or — ConvJob::rgb12, for node textures that only the matrix-core resampler
// reads — 12 bytes per four pixels: R0 R1 R2 R3 G0 G1 G2 G3 B0 B1 B2 B3 (alpha is 1 for every Y'CbCr frame, planar_yuv_to_rgba.wgsl:57, and
// is not stored).  The route is bound by the traffic of its intermediates (profiles/r04_wave_ablation.txt): a quarter less to write here
// and to read there, and a lane of the resampler still fetches its four texels with ONE load (three planes needed three: slower,
// profiles/r04_planar_nodes.txt).
#pragma once

#include "smr_convert_dev.h"

#ifdef SMR_EMU
#define cv_perm(hi, lo, sel) dev_perm((hi), (lo), (sel))
#define cv_alignbyte(hi, lo, sh) dev_alignbyte((hi), (lo), (sh))
#define cv_clamp01(x) dev_fmed3((x), 0.0f, 1.0f)
#define cv_mad24(a, b, c) dev_mad24((a), (b), (c))
#define cv_uniform(x) (x)
#else
#define cv_uniform(x) __builtin_amdgcn_readfirstlane(x)  // a value every lane of the wave holds alike, moved to a scalar register
#define cv_perm(hi, lo, sel) __builtin_amdgcn_perm((hi), (lo), (sel))
#define cv_alignbyte(hi, lo, sh) __builtin_amdgcn_alignbyte((hi), (lo), (sh))
#define cv_clamp01(x) __builtin_amdgcn_fmed3f((x), 0.0f, 1.0f)
#define cv_mad24(a, b, c) ((u32)__umul24((a), (b)) + (c))  // row * pitch + offset with both factors below 2^24: one full-rate v_mad_u32_u24 (a 32 x 32 multiply is a quarter-rate v_mad_u64_u32)
#endif

// ---- the unorm8 store of a row's twelve values: floor(clamp(x, 0, 1) * 255 + 0.5) (planar_yuv_to_rgba.wgsl:57 through the Rgba8Unorm target).
// v_cvt_pk_u8_f32 converts, saturates to [0, 255] and inserts the byte into a dword in ONE instruction — with the wave's f32 rounding mode:
// round-to-nearest-even by default (tools/ubench/cvt_pk_u8.hip: not the reference's truncation), but under round-toward-zero exactly the
// truncation (tools/ubench/cvt_pk_u8_mode.hip on the device: every f32 within 4 ulp of every integer and half-integer of [0, 258), a sweep of
// [-2, 298] and the specials — 4.2 M values, no difference from trunc(clamp(x, 0, 255))).  The operands t = x * 255 + 0.5 are computed before, in the
// default mode, WITHOUT the clamp (t <= 0.5 truncates / saturates to 0 like clamp's 0.5, t >= 255.5 saturates to 255 like clamp's 255.5; NaN
// cannot occur); the mode is switched inside one asm block around the twelve conversions, so no other float instruction can be scheduled into
// the switched region.  Three instructions per value (multiply, add, convert) instead of five (median, multiply, add, convert, shift-or).
// dst[c] |= byte i of channel c for the row's pixel i: RGB12 rows are (R x 4, G x 4, B x 4) dwords; RGBA8 rows take cv_store_px.
#ifdef SMR_EMU
static inline u32 cv_emu_u8(float t) { return t >= 255.0f ? 255u : (t > 0.0f ? (u32)t : 0u); }
#define CV_U8X4(d, t0, t1, t2, t3) do { (d) = cv_emu_u8(t0) | (cv_emu_u8(t1) << 8) | (cv_emu_u8(t2) << 16) | (cv_emu_u8(t3) << 24); } while (0)
static inline void cv_row_bytes_planar(const float (&t)[3][4], u32 &r4, u32 &g4, u32 &b4) {
    CV_U8X4(r4, t[0][0], t[0][1], t[0][2], t[0][3]); CV_U8X4(g4, t[1][0], t[1][1], t[1][2], t[1][3]); CV_U8X4(b4, t[2][0], t[2][1], t[2][2], t[2][3]);
}
static inline void cv_row_bytes_packed(const float (&t)[3][4], u32 (&px)[4]) {
    for (int i = 0; i < 4; i++) px[i] = cv_emu_u8(t[0][i]) | (cv_emu_u8(t[1][i]) << 8) | (cv_emu_u8(t[2][i]) << 16) | 0xff000000u;
}
#else
__device__ __forceinline__ void cv_row_bytes_planar(const float (&t)[3][4], u32 &r4, u32 &g4, u32 &b4) {
    u32 r = 0u, g = 0u, b = 0u;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 0, %0\n\tv_cvt_pk_u8_f32 %1, %7, 0, %1\n\tv_cvt_pk_u8_f32 %2, %11, 0, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\tv_cvt_pk_u8_f32 %1, %8, 1, %1\n\tv_cvt_pk_u8_f32 %2, %12, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\tv_cvt_pk_u8_f32 %1, %9, 2, %1\n\tv_cvt_pk_u8_f32 %2, %13, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, 3, %0\n\tv_cvt_pk_u8_f32 %1, %10, 3, %1\n\tv_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1"
                 : "+v"(r), "+v"(g), "+v"(b)
                 : "v"(t[0][0]), "v"(t[0][1]), "v"(t[0][2]), "v"(t[0][3]), "v"(t[1][0]), "v"(t[1][1]), "v"(t[1][2]), "v"(t[1][3]), "v"(t[2][0]), "v"(t[2][1]),
                   "v"(t[2][2]), "v"(t[2][3]));
    r4 = r; g4 = g; b4 = b;
}
__device__ __forceinline__ void cv_row_bytes_packed(const float (&t)[3][4], u32 (&px)[4]) {
    u32 p0 = 0xff000000u, p1 = 0xff000000u, p2 = 0xff000000u, p3 = 0xff000000u;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 0, %0\n\tv_cvt_pk_u8_f32 %1, %5, 0, %1\n\tv_cvt_pk_u8_f32 %2, %6, 0, %2\n\tv_cvt_pk_u8_f32 %3, %7, 0, %3\n\t"
                 "v_cvt_pk_u8_f32 %0, %8, 1, %0\n\tv_cvt_pk_u8_f32 %1, %9, 1, %1\n\tv_cvt_pk_u8_f32 %2, %10, 1, %2\n\tv_cvt_pk_u8_f32 %3, %11, 1, %3\n\t"
                 "v_cvt_pk_u8_f32 %0, %12, 2, %0\n\tv_cvt_pk_u8_f32 %1, %13, 2, %1\n\tv_cvt_pk_u8_f32 %2, %14, 2, %2\n\tv_cvt_pk_u8_f32 %3, %15, 2, %3\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1"
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                 : "v"(t[0][0]), "v"(t[0][1]), "v"(t[0][2]), "v"(t[0][3]), "v"(t[1][0]), "v"(t[1][1]), "v"(t[1][2]), "v"(t[1][3]), "v"(t[2][0]), "v"(t[2][1]),
                   "v"(t[2][2]), "v"(t[2][3]));
    px[0] = p0; px[1] = p1; px[2] = p2; px[3] = p3;
}
#endif

struct ConvJob {
    SurfView yp, up, vp, dst;
    int full, nv;  // full range (J420) | NV12 (interleaved chroma in `up`)
    int sx, sy;    // chroma subsampling: 4:2:0 = (1, 1), 4:2:2 = (1, 0), 4:4:4 = (0, 0)
    int rgb12;     // k_yuv420_to_rgba only: dst is an R8 surface of 3 w x h — the node texture as 12-byte groups of four pixels (R x 4, G x 4, B x 4)
    int packed;    // 0 planar / NV12 | 1 UYVY | 2 YUYV: `yp` is the (w / 2) x h plane of U Y0 V Y1 / Y0 U Y1 V groups | 3 BGRA | 4 ARGB: `yp` is the w x h plane, bytes permuted (bgra_to_rgba.wgsl / argb_to_rgba.wgsl:24-28)
};
constexpr int MAX_CONV_JOBS = 16;
struct ConvBatch {
    ConvJob j[MAX_CONV_JOBS];
    // k_yuv420_to_rgba: the launch's work in UNITS — a unit = 64 column groups (256 pixels) x one block row (4 rows) of one job, what a wave
    // computes per cv420_run step — dealt to the XCDs in BANDS (cv420_plan): a band = `band` consecutive block rows of a job across all its
    // column blocks, band g of the launch (jobs one after the other) belongs to XCD g % 8.  Inside an XCD the units are ordered band by band,
    // column block by column block, block rows fastest, and cut into equal contiguous shares, one per wave the XCD runs (workgroup b runs on
    // XCD b % 8 — observed, used for locality only: the partition is arithmetic on blockIdx and complete whatever the placement).
    // Why bands: a column block's chroma window reaches a byte or two into its neighbours' cache lines; neighbours on different XCDs made
    // every L2 fetch three lines for one (counters: 62 MB for 24.9 MB of planes with shares dealt without regard to the XCDs).
    int n;
    u32 band;                                  // block rows per band
    u32 rows[MAX_CONV_JOBS];                   // block rows of job j
    u32 cols[MAX_CONV_JOBS];                   // column blocks of job j
    u32 band0[MAX_CONV_JOBS];                  // launch-wide index of job j's first band
    u32 first_unit[8][MAX_CONV_JOBS + 1];      // per XCD x: job j owns units [first_unit[x][j], first_unit[x][j + 1]) of x's sequence
};

// Fills the partition tables of a launch of n jobs (j[].dst.w / .h set) that will run `waves` waves in all; returns the number of bands
// (the launch needs at least min(8, bands) workgroups: every XCD that owns units must run a wave).
inline u32 cv420_plan(ConvBatch &B, int n, u32 waves) {
    B.n = n;
    u32 total = 0;
    for (int j = 0; j < n; j++) {
        B.cols[j] = ((u32)B.j[j].dst.w + 255u) / 256u;
        B.rows[j] = ((u32)B.j[j].dst.h + 3u) / 4u;
        total += B.cols[j] * B.rows[j];
    }
    // Band height: every share of a band runs on the band's XCD whatever its length, so a band may be much taller than a share — and the
    // taller, the fewer chroma rows two XCDs both fetch (a band's window reaches one chroma row into the bands above and below: 2 rows per
    // band of 2 * band).  Against that, the XCDs' totals differ by up to one band: sixteen bands per XCD and more keeps that under a
    // sixteenth (8 x 1080p: bands of 16 block rows, 17 per XCD, 6 % of the chroma rows fetched twice).
    (void)waves;
    u32 cols_max = 1;
    for (int j = 0; j < n; j++) cols_max = B.cols[j] > cols_max ? B.cols[j] : cols_max;
    const u32 band = total / (128u * cols_max);
    B.band = band < 1u ? 1u : (band > 64u ? 64u : band);
    u32 g = 0;
    for (int x = 0; x < 8; x++) B.first_unit[x][0] = 0;
    for (int j = 0; j < n; j++) {
        B.band0[j] = g;
        const u32 bands = (B.rows[j] + B.band - 1u) / B.band;
        u32 units[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (u32 b = 0; b < bands; b++) {
            const u32 h = B.rows[j] - b * B.band < B.band ? B.rows[j] - b * B.band : B.band;
            units[(g + b) & 7u] += B.cols[j] * h;
        }
        for (int x = 0; x < 8; x++) B.first_unit[x][j + 1] = B.first_unit[x][j] + units[x];
        g += bands;
    }
    return g;
}

#ifndef CV_ABL
#define CV_ABL 0  // laboratory builds (tools/variant.sh NAME -DCV_ABL=n): 1 no stores | 2 no chroma loads | 4 no luma loads | 8 no per-pixel arithmetic
#endif

#ifdef __HIPCC__

// CV_TIMING (tools/variant.sh timing -DCV_TIMING): lane 0 of every wave stamps the shader clock at the phase boundaries of its run —
// each stamp behind an explicit wait for what the phase requested — into g_cv_stamps[wave][...]; tools/r05/conv_timing.py reads them back.
#if defined(CV_TIMING) && !defined(SMR_EMU)
#define CV_STAMP(st, i, waits) do { asm volatile(waits ::: "memory"); if ((st) && (threadIdx.x & 63) == 0) (st)[i] = __builtin_readcyclecounter(); asm volatile("" ::: "memory"); } while (0)
#else
#define CV_STAMP(st, i, waits) do { } while (0)
#endif

// y' of a luma byte: planar_yuv_to_rgba.wgsl:46 on byte / 255 (limited range), the byte's unorm value itself (full range)
__device__ __forceinline__ float cv420_luma_of_byte(u32 b, bool full) {
    const float y = unorm_of_byte(b);
    if (full) return y;
    constexpr float ky = 0.85882352941f;
    const float ry = 1.0f / ky;
    const float a = y - (16.0f / 255.0f), q = a * ry;
    return cv_clamp01(__builtin_fmaf(__builtin_fmaf(-q, ky, a), ry, q));
}

// ---- the pieces of a block (cv420_block below = one block; cv420_run = a vertical run of blocks that shares what neighbouring blocks share)

// Where column group g's chroma window (columns 2 g - 1 .. 2 g + 2, clamped to the plane like the sampler clamps) sits in a chroma row
struct Cv420Win {
    u32 base;       // byte offset of the dword the window's first byte lies in
    u32 sh;         // ... and the byte's offset inside that dword
    u32 right_fix;  // v_perm selector: the window's last column repeats the one before it at the plane's right edge
    bool left;      // the window starts left of the plane: columns 0 1 2 3 -> 0 0 1 2
    u32 lim;        // TIGHT builds: offset of the last dword that holds a column of the plane (0, 4 or 8 from base): nothing behind it is loaded
};
template <bool NV>
__device__ __forceinline__ Cv420Win cv420_window(const ConvJob &J, int g) {
    const int cw = J.dst.w >> 1;
    const int first = 2 * g - 1, first_ld = first < 0 ? 0 : first;
    const int byte0 = NV ? 2 * first_ld : first_ld, base = byte0 & ~3;
    const int nvalid = cw - first;  // window columns 0 .. nvalid - 1 exist (>= 3: the last block's window starts at cw - 3)
    Cv420Win W;
    W.base = (u32)base; W.sh = (u32)(byte0 - base); W.right_fix = nvalid >= 4 ? 0x03020100u : 0x02020100u; W.left = first < 0;
    // the last column the loaded window can hold is min(first_ld + 3, cw - 1): its last byte decides which dwords are worth loading
    const int last_col = first_ld + 3 < cw - 1 ? first_ld + 3 : cw - 1;
    W.lim = (u32)(((NV ? 2 * last_col + 1 : last_col) - base) & ~3);
    return W;
}

// The dwords of one chroma row that hold the window, as loaded (planar: U, U + 4, V, V + 4; NV12: three dwords of U V pairs)
template <bool NV>
struct Cv420Raw {
    u32 d[NV ? 3 : 4];
};
// TIGHT: a plane whose rows fill its pitch with nothing behind the last row (a wrapped decoder surface): the dwords behind the window's last
// column — read and ignored by the plain build, in the next row or the allocation's tail — are not requested; the dword before them is loaded
// again instead (its bytes land where the ignored ones would).  Same results, two more vector instructions per load; a kernel of its own.
template <bool NV, bool TIGHT = false>
__device__ __forceinline__ Cv420Raw<NV> cv420_load_chroma(const ConvJob &J, const Cv420Win &W, int crow) {
    const int ch = J.dst.h >> 1;
    const int cy = min(max(crow, 0), ch - 1);
    Cv420Raw<NV> R;
    const u8 *ur = J.up.ptr + cv_mad24((u32)cy, J.up.pitch, W.base);  // (one 32-bit offset from a uniform base: a plane is far below 4 GiB)
    if (CV_ABL & 2) {
#pragma unroll
        for (int k = 0; k < (NV ? 3 : 4); k++) R.d[k] = 0x01020304u * (u32)(crow + k) + W.base;
    } else if (NV) {
        const u32 o1 = TIGHT ? (W.lim < 4u ? W.lim : 4u) : 4u, o2 = TIGHT ? (W.lim < 8u ? W.lim : 8u) : 8u;
        R.d[0] = g_ld_u32(ur); R.d[1] = g_ld_u32(ur + o1); R.d[2] = g_ld_u32(ur + o2);
    } else {
        const u8 *vr = J.vp.ptr + cv_mad24((u32)cy, J.vp.pitch, W.base);
        const u32 o1 = TIGHT ? (W.lim < 4u ? W.lim : 4u) : 4u;
        R.d[0] = g_ld_u32(ur); R.d[1] = g_ld_u32(ur + o1); R.d[2] = g_ld_u32(vr); R.d[3] = g_ld_u32(vr + o1);
    }
    return R;
}
__device__ __forceinline__ void cv420_load_luma(const ConvJob &J, int g, int P, u32 yrow[4]) {
    const int h = J.dst.h;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (CV_ABL & 4) yrow[r] = 0x10203040u * (u32)(g + r) + (u32)P;
        else yrow[r] = g_ld_u32(J.yp.ptr + cv_mad24((u32)min(4 * P + r, h - 1), J.yp.pitch, 4u * (u32)g));
    }
}

// One chroma row of the window -> its horizontal lerps at the block's four luma columns, per plane: H[plane][column]
template <bool NV>
__device__ __forceinline__ void cv420_hrow(const Cv420Raw<NV> &R, const Cv420Win &W, const float *nlut, float H[2][4]) {
    u32 uw, vw;
    if (NV) {
        const u32 w0 = cv_alignbyte(R.d[1], R.d[0], W.sh), w1 = cv_alignbyte(R.d[2], R.d[1], W.sh);  // U V U V of two columns each
        uw = cv_perm(w1, w0, 0x06040200u);
        vw = cv_perm(w1, w0, 0x07050301u);
    } else {
        uw = cv_alignbyte(R.d[1], R.d[0], W.sh);
        vw = cv_alignbyte(R.d[3], R.d[2], W.sh);
    }
    if (W.left) {  // columns 0 1 2 3 -> 0 0 1 2
        uw = (uw << 8) | (uw & 0xffu);
        vw = (vw << 8) | (vw & 0xffu);
    }
    uw = cv_perm(0u, uw, W.right_fix);
    vw = cv_perm(0u, vw, W.right_fix);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const u32 q = c ? vw : uw;
        const float n0 = nlut[q & 0xffu], n1 = nlut[(q >> 8) & 0xffu], n2 = nlut[(q >> 16) & 0xffu], n3 = nlut[q >> 24];
        // a * (1 - fx) + b * fx with fx = .75, .25, .75, .25 (sample_plane_bilinear): the 1/4 products are exact
        const float m1 = n1 * 0.75f, m2 = n2 * 0.75f;
        H[c][0] = __builtin_fmaf(n0, 0.25f, m1);
        H[c][1] = __builtin_fmaf(n2, 0.25f, m1);
        H[c][2] = __builtin_fmaf(n1, 0.25f, m2);
        H[c][3] = __builtin_fmaf(n3, 0.25f, m2);
    }
}

// The four luma rows of block (g, P) from its luma dwords and the window's four chroma rows H[window row][plane][column]:
// row 4 P + r takes window rows (0, 1) with fy = .75, (1, 2) with .25, (1, 2) with .75, (2, 3) with .25 — top * (1 - fy) + bot * fy:
// the 3/4 product of window row 1 serves luma rows 0 and 1, that of row 2 serves rows 2 and 3.
// (c - 16/255) / 0.8784 as RN(a * y_hi + RN(a * y_lo)) with y_hi + y_lo = 1 / 0.8784 to 48 bits: the IEEE quotient for EVERY f32 a in
// [-16/255, 1] — all 636 524 221 of them checked against the division (tools/check_div_by_constant.py), as unorm_of_byte's form is
// for the 256 bytes.  One multiply and one fused multiply-add.
// ALLROWS: the caller knows that all four rows of the block exist (every block of a run but its last): no branch around the stores, so the
// compiler can count the stores in flight and wait for the NEXT block's loads — requested before them — with vmcnt(stores) instead of vmcnt(0)
// (the memory counter counts loads and stores alike: with the branch every block waited for its own stores' round trip)
// SINK: where a finished row goes — sink(r, y, r4, g4, b4, px): row r of the block = frame row y, as RGB12 dwords (RGB12) or four RGBA8 pixels.
// The default stores into the job's node texture; k_ingest_wave's plane-source builds (smr_ingest_wave.h) hand the rows to LDS instead.
struct Cv420StoreNode {
    const ConvJob &J;
    int g;
    bool rgb12;
    __device__ __forceinline__ void operator()(int, int y, u32 r4, u32 g4, u32 b4, const u32 (&px)[4]) const {
        if (rgb12) g_st_u32x3(J.dst.ptr + cv_mad24((u32)y, J.dst.pitch, 12u * (u32)g), r4, g4, b4);
        else g_st_u32x4(J.dst.ptr + cv_mad24((u32)y, J.dst.pitch, 16u * (u32)g), make_uint4(px[0], px[1], px[2], px[3]));
    }
};
template <bool RGB12, bool FULL, bool ALLROWS, typename SINK>
__device__ __forceinline__ void cv420_rows_to(const ConvJob &J, int g, int P, const u32 yrow[4], const float H[4][2][4], const float *ylut, const SINK &sink) {
    const int h = J.dst.h;
    constexpr bool full = FULL;  // (a template parameter: as a run-time flag it was a scalar branch per pixel)
    constexpr float kc = 0.87843137254f;
    constexpr float rc_hi = 1.0f / kc, rc_lo = (float)(1.0 / (double)kc - (double)rc_hi);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = 4 * P + r;
        if (!ALLROWS && y >= h) break;
        const int j34 = r < 2 ? 1 : 2, j14 = r == 0 ? 0 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
        const u32 y4 = yrow[r];
        u32 px[4] = {0u, 0u, 0u, 0u}, r4 = 0u, g4 = 0u, b4 = 0u;
        float t[3][4];  // x * 255 + 0.5 of the row's twelve values (cv_row_bytes_* truncates and packs them)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (CV_ABL & 8) {
                const u32 b = (y4 >> (8 * i)) & 0xffu;
                px[i] = b | (__float_as_uint(H[j14][0][i] + H[j34][1][i]) & 0xffff00u) | 0xff000000u;
                r4 |= b << (8 * i); g4 |= (__float_as_uint(H[j14][0][i]) & 0xffu) << (8 * i); b4 |= (__float_as_uint(H[j34][1][i]) & 0xffu) << (8 * i);
                continue;
            }
            float u = __builtin_fmaf(H[j14][0][i], 0.25f, H[j34][0][i] * 0.75f);
            float v = __builtin_fmaf(H[j14][1][i], 0.25f, H[j34][1][i] * 0.75f);
            const float yy = ylut[(y4 >> (8 * i)) & 0xffu];
            if (!full) {  // planar_yuv_to_rgba.wgsl:47-48: (c - 16/255) / 0.8784, clamp
                const float au = u - (16.0f / 255.0f), av = v - (16.0f / 255.0f);
                u = cv_clamp01(__builtin_fmaf(au, rc_hi, au * rc_lo));
                v = cv_clamp01(__builtin_fmaf(av, rc_hi, av * rc_lo));
            }
            const float um = u - 0.5f, vm = v - 0.5f;
            const float R = yy + 1.5748f * vm;
            const float G = yy - 0.1873f * um - 0.4681f * vm;
            const float B = yy + 1.8556f * um;
            t[0][i] = R * 255.0f + 0.5f;
            t[1][i] = G * 255.0f + 0.5f;
            t[2][i] = B * 255.0f + 0.5f;
        }
        if (!(CV_ABL & 8)) {
            if (RGB12) cv_row_bytes_planar(t, r4, g4, b4);
            else cv_row_bytes_packed(t, px);
        }
        if ((CV_ABL & 1) && (r4 ^ g4 ^ b4 ^ px[0] ^ px[3]) != 0x12345677u) continue;  // (never equal in practice: the values stay live)
        sink(r, y, r4, g4, b4, px);
    }
}
template <bool RGB12, bool FULL, bool ALLROWS = false>
__device__ __forceinline__ void cv420_rows(const ConvJob &J, int g, int P, const u32 yrow[4], const float H[4][2][4], const float *ylut) {
    cv420_rows_to<RGB12, FULL, ALLROWS>(J, g, P, yrow, H, ylut, Cv420StoreNode{J, g, RGB12});
}

// One 4 x 4 block: columns 4 g .. 4 g + 3, rows 4 P .. 4 P + 3 of job J (rows past the frame's height are not stored).
// ylut: 256 floats, cv420_luma_of_byte of every byte for this job's range.
// Requirements (cv420_job_ok on the host): 4:2:0, even height, width a multiple of 4, dword-aligned planes whose rows can be read a
// dword past the window, 16-byte aligned destination rows.
// nlut: 256 floats, unorm_of_byte of every byte (the chroma bytes' byte / 255: a table gather instead of a conversion and two multiply-adds)
// RGB12: the node texture as 12-byte groups (ConvJob::rgb12), else RGBA8 — separate instantiations: each packs its own bytes only
template <bool NV, bool RGB12, bool FULL, bool TIGHT = false>
__device__ __forceinline__ void cv420_block(const ConvJob &J, int g, int P, const float *ylut, const float *nlut) {
    // every load of the block is in flight before the first store (a row's luma load behind the previous row's store waited for that
    // store and for itself: four memory round trips per block instead of one)
    u32 yrow[4];
    cv420_load_luma(J, g, P, yrow);
    const Cv420Win W = cv420_window<NV>(J, g);
    Cv420Raw<NV> raw[4];
#pragma unroll
    for (int j = 0; j < 4; j++) raw[j] = cv420_load_chroma<NV, TIGHT>(J, W, 2 * P - 1 + j);  // chroma rows 2 P - 1 .. 2 P + 2
    float H[4][2][4];  // [window row][plane][luma column]
#pragma unroll
    for (int j = 0; j < 4; j++) cv420_hrow<NV>(raw[j], W, nlut, H[j]);
    cv420_rows<RGB12, FULL>(J, g, P, yrow, H, ylut);
}

// A vertical run of blocks: columns 4 g .. 4 g + 3, block rows P0 .. P0 + nb - 1 (those that exist).  The same values as cv420_block on
// each of them — the same operations on the same operands — with what two vertically adjacent blocks share done once and the memory
// latency of block P + 1 spent under the arithmetic of block P:
//   * block P + 1's chroma window rows 0, 1 are block P's rows 2, 3 (chroma rows 2 P + 1, 2 P + 2, clamped alike): their loads, byte / 255
//     gathers and horizontal lerps are kept — half of the chroma work of every block after the run's first;
//   * block P + 1's luma dwords and its two new chroma rows are requested BEFORE block P's rows are computed and stored, so they arrive
//     while the wave's vector ALU is busy (a one-block thread loads, waits, computes, stores: its waves all wait at the same time);
//     the run's last block requests nothing (the loop is peeled: inside it the requests are unconditional, so the compiler can count
//     what is outstanding).
template <bool NV, bool RGB12, bool FULL, bool TIGHT = false>
__device__ __forceinline__ void cv420_run(const ConvJob &J, int g, int P0, int nb, const float *ylut, const float *nlut, unsigned long long *st = nullptr) {
    const int Pend = min(P0 + nb, (J.dst.h + 3) >> 2);
    if (P0 >= Pend) return;
    const Cv420Win W = cv420_window<NV>(J, g);
    u32 yrow[4];
    cv420_load_luma(J, g, P0, yrow);
    float H[4][2][4];
    {
        Cv420Raw<NV> raw[4];
#pragma unroll
        for (int j = 0; j < 4; j++) raw[j] = cv420_load_chroma<NV, TIGHT>(J, W, 2 * P0 - 1 + j);
        CV_STAMP(st, 2, "s_waitcnt vmcnt(0)");  // the first block's loads have arrived
#pragma unroll
        for (int j = 0; j < 4; j++) cv420_hrow<NV>(raw[j], W, nlut, H[j]);
        CV_STAMP(st, 3, "s_waitcnt lgkmcnt(0)");  // its chroma window is converted
    }
    // every block but the run's last requests its successor before it is computed; the last one requests nothing (its successor belongs to
    // another wave, which fetched it at ITS start — microseconds ago: by now the lines have left the L2 and the request would go to memory
    // again: + 9 MB of fetch per 8 x 1080p launch when the loop requested unconditionally)
    int P = P0;
    for (; P + 1 < Pend; P++) {
        u32 ynext[4];
        cv420_load_luma(J, g, P + 1, ynext);
        const Cv420Raw<NV> n2 = cv420_load_chroma<NV, TIGHT>(J, W, 2 * P + 3), n3 = cv420_load_chroma<NV, TIGHT>(J, W, 2 * P + 4);
        cv420_rows<RGB12, FULL, true>(J, g, P, yrow, H, ylut);  // (P + 1 < Pend: not the frame's last block row — all four rows exist)
        if (P == P0) CV_STAMP(st, 4, "s_nop 0");  // the first block's rows are computed, its stores issued
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) { H[0][c][i] = H[2][c][i]; H[1][c][i] = H[3][c][i]; }
        cv420_hrow<NV>(n2, W, nlut, H[2]);
        cv420_hrow<NV>(n3, W, nlut, H[3]);
#pragma unroll
        for (int r = 0; r < 4; r++) yrow[r] = ynext[r];
    }
    cv420_rows<RGB12, FULL>(J, g, P, yrow, H, ylut);
    if (P == P0) CV_STAMP(st, 4, "s_nop 0");
}

// A job's geometry as VALUES in scalar registers.  Read in place (B.j[j].dst.pitch ...) the compiler treats the kernel-argument segment as
// memory it may re-read whenever that is cheaper than keeping a register: it did, behind every row's store — an s_load_dword plus
// s_waitcnt lgkmcnt(0) (scalar loads return out of order, so the wait also drains the table gathers in flight) per row, ~200 cycles
// each, in every wave at once.  A value that went through v_readfirstlane is a computed value: it stays where it is.
__device__ __forceinline__ SurfView cv420_view_in_registers(const SurfView &v) {
#ifdef SMR_EMU
    return v;
#else
    SurfView r;
    const unsigned long long p = (unsigned long long)(uintptr_t)v.ptr;
    const u32 lo = cv_uniform((u32)p), hi = cv_uniform((u32)(p >> 32));
    r.ptr = (u8 *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    r.pitch = cv_uniform(v.pitch);
    r.w = cv_uniform(v.w);
    r.h = cv_uniform(v.h);
    return r;
#endif
}
__device__ __forceinline__ ConvJob cv420_job_in_registers(const ConvJob &j) {
    ConvJob J = j;
    J.yp = cv420_view_in_registers(j.yp); J.up = cv420_view_in_registers(j.up); J.vp = cv420_view_in_registers(j.vp); J.dst = cv420_view_in_registers(j.dst);
    return J;
}

// Wave `wave` (0 .. 3) of workgroup `block` of a launch of `grid` workgroups: XCD x = block % 8 runs the workgroups x, x + 8, ...; its
// waves cut x's unit sequence (ConvBatch) into equal contiguous shares — total / waves units, the first total % waves waves one more.  A
// share is a vertical run of blocks inside one (band, column block) cell, or the end of a cell and the start of the next (in the next
// band or job, too).  ylut: the limited-range luma table; nlut: byte / 255, which is also the full-range luma table.
template <bool NV, bool TIGHT = false>
__device__ __forceinline__ void cv420_share(const ConvBatch &B, u32 block, u32 wave, u32 grid, u32 lane, const float *ylut, const float *nlut,
                                            unsigned long long *st = nullptr) {
    const u32 x = block & 7u;
    const u32 waves = ((grid - x + 7u) >> 3) * 4u, w = (block >> 3) * 4u + wave;  // of this XCD
    const u32 total = B.first_unit[x][B.n];
    const u32 share = total / waves, extra = total - share * waves;
    u32 lo = w * share + (w < extra ? w : extra);
    const u32 hi = lo + share + (w < extra ? 1u : 0u);
    int j = 0;
    while (lo < hi) {  // (uniform: one or two runs per share, more only across tiny jobs)
#pragma unroll 1
        while (lo >= B.first_unit[x][j + 1]) j++;
        const ConvJob J = cv420_job_in_registers(B.j[j]);
        const u32 rows = B.rows[j], cols = B.cols[j], band = B.band;
        // x's bands of job j: b0, b0 + 8, ...; all but the job's last band (which, if x owns it, is the last of them) hold cols * band units
        const u32 b0 = (x - B.band0[j]) & 7u;
        const u32 local = lo - B.first_unit[x][j];
        const u32 t = local / (cols * band), b = b0 + 8u * t;
        const u32 in_band = local - t * cols * band;
        const u32 band_rows = rows - b * band < band ? rows - b * band : band;
        const u32 col = in_band / band_rows, r = in_band - col * band_rows;
        const u32 nrun = hi - lo < band_rows - r ? hi - lo : band_rows - r;
        const int g = (int)(col * 64u + lane);
        if (4 * g < J.dst.w) {
            const int P0 = (int)(b * band + r);
            // (uniform branches: a job is one frame; range and node format are template parameters — as run-time flags they cost a scalar branch per pixel)
            if (J.rgb12) {
                if (J.full) cv420_run<NV, true, true, TIGHT>(J, g, P0, (int)nrun, nlut, nlut, st);
                else cv420_run<NV, true, false, TIGHT>(J, g, P0, (int)nrun, ylut, nlut, st);
            } else {
                if (J.full) cv420_run<NV, false, true, TIGHT>(J, g, P0, (int)nrun, nlut, nlut, st);
                else cv420_run<NV, false, false, TIGHT>(J, g, P0, (int)nrun, ylut, nlut, st);
            }
        }
        st = nullptr;  // (timing builds: only the share's first run is stamped phase by phase)
        lo += nrun;
    }
}

#endif  // __HIPCC__
