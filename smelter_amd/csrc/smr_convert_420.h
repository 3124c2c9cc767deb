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
#else
#define cv_perm(hi, lo, sel) __builtin_amdgcn_perm((hi), (lo), (sel))
#define cv_alignbyte(hi, lo, sh) __builtin_amdgcn_alignbyte((hi), (lo), (sh))
#define cv_clamp01(x) __builtin_amdgcn_fmed3f((x), 0.0f, 1.0f)
#define cv_mad24(a, b, c) ((u32)__umul24((a), (b)) + (c))  // row * pitch + offset with both factors below 2^24: one full-rate v_mad_u32_u24 (a 32 x 32 multiply is a quarter-rate v_mad_u64_u32)
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
    int order;      // (profiling: SMR_CONVERT_ORDER) 0 box in launch order | 1 an eighth of the box per XCD | 2 block rows round robin over the XCDs
    int gx, gy, n;  // k_yuv420_to_rgba: the launch as a gx x gy x n box of workgroups (its 1-D grid is that box walked XCD by XCD)
};

#ifndef CV_ABL
#define CV_ABL 0  // profiling builds (tools/variant_convert.sh): 1 no stores | 2 no chroma loads | 4 no luma loads | 8 no per-pixel arithmetic
#endif

#ifdef __HIPCC__

// y' of a luma byte: planar_yuv_to_rgba.wgsl:46 on byte / 255 (limited range), the byte's unorm value itself (full range)
__device__ __forceinline__ float cv420_luma_of_byte(u32 b, bool full) {
    const float y = unorm_of_byte(b);
    if (full) return y;
    constexpr float ky = 0.85882352941f;
    const float ry = 1.0f / ky;
    const float a = y - (16.0f / 255.0f), q = a * ry;
    return cv_clamp01(__builtin_fmaf(__builtin_fmaf(-q, ky, a), ry, q));
}

// One 4 x 4 block: columns 4 g .. 4 g + 3, rows 4 P .. 4 P + 3 of job J (rows past the frame's height are not stored).
// ylut: 256 floats, cv420_luma_of_byte of every byte for this job's range.
// Requirements (cv420_job_ok on the host): 4:2:0, even height, width a multiple of 4, dword-aligned planes whose rows can be read a
// dword past the window, 16-byte aligned destination rows.
// nlut: 256 floats, unorm_of_byte of every byte (the chroma bytes' byte / 255: a table gather instead of a conversion and two multiply-adds)
// RGB12: the node texture as 12-byte groups (ConvJob::rgb12), else RGBA8 — separate instantiations: each packs its own bytes only
template <bool NV, bool RGB12>
__device__ __forceinline__ void cv420_block(const ConvJob &J, int g, int P, const float *ylut, const float *nlut) {
    const int w = J.dst.w, h = J.dst.h, cw = w >> 1, ch = h >> 1;
    const bool full = J.full != 0;
    // ---- the four luma dwords first, with the chroma window's loads: every load of the block is in flight before the first store (a row's
    //      luma load behind the previous row's store waited for that store and for itself: four memory round trips per block instead of one)
    u32 yrow[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (CV_ABL & 4) yrow[r] = 0x10203040u * (u32)(g + r) + (u32)P;
        else yrow[r] = *(const u32 *)(J.yp.ptr + cv_mad24((u32)min(4 * P + r, h - 1), J.yp.pitch, 4u * (u32)g));
    }
    // ---- chroma window: columns 2 g - 1 .. 2 g + 2, rows 2 P - 1 .. 2 P + 2, clamped to the plane like the sampler clamps
    const int first = 2 * g - 1, first_ld = first < 0 ? 0 : first;
    const int byte0 = NV ? 2 * first_ld : first_ld, base = byte0 & ~3;
    const u32 sh = (u32)(byte0 - base);
    const int nvalid = cw - first;  // window columns 0 .. nvalid - 1 exist (>= 3: the last block's window starts at cw - 3)
    const u32 right_fix = nvalid >= 4 ? 0x03020100u : 0x02020100u;
    float H[2][4][4];  // [plane][window row][luma column]: the row's horizontal lerp at the block's four columns
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int cy = min(max(2 * P - 1 + j, 0), ch - 1);
        const u8 *ur = J.up.ptr + cv_mad24((u32)cy, J.up.pitch, (u32)base);  // (one 32-bit offset from a uniform base: a plane is far below 4 GiB)
        u32 uw, vw;
        if (CV_ABL & 2) {
            uw = 0x01020304u * (u32)(g + j) + (u32)cy; vw = uw ^ 0x55aa55aau;
        } else if (NV) {
            const u32 d0 = *(const u32 *)ur, d1 = *(const u32 *)(ur + 4), d2 = *(const u32 *)(ur + 8);
            const u32 w0 = cv_alignbyte(d1, d0, sh), w1 = cv_alignbyte(d2, d1, sh);  // U V U V of two columns each
            uw = cv_perm(w1, w0, 0x06040200u);
            vw = cv_perm(w1, w0, 0x07050301u);
        } else {
            const u8 *vr = J.vp.ptr + cv_mad24((u32)cy, J.vp.pitch, (u32)base);
            uw = cv_alignbyte(*(const u32 *)(ur + 4), *(const u32 *)ur, sh);
            vw = cv_alignbyte(*(const u32 *)(vr + 4), *(const u32 *)vr, sh);
        }
        if (first < 0) {  // columns 0 1 2 3 -> 0 0 1 2
            uw = (uw << 8) | (uw & 0xffu);
            vw = (vw << 8) | (vw & 0xffu);
        }
        uw = cv_perm(0u, uw, right_fix);
        vw = cv_perm(0u, vw, right_fix);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const u32 q = c ? vw : uw;
            const float n0 = nlut[q & 0xffu], n1 = nlut[(q >> 8) & 0xffu], n2 = nlut[(q >> 16) & 0xffu], n3 = nlut[q >> 24];
            // a * (1 - fx) + b * fx with fx = .75, .25, .75, .25 (sample_plane_bilinear): the 1/4 products are exact
            const float m1 = n1 * 0.75f, m2 = n2 * 0.75f;
            H[c][j][0] = __builtin_fmaf(n0, 0.25f, m1);
            H[c][j][1] = __builtin_fmaf(n2, 0.25f, m1);
            H[c][j][2] = __builtin_fmaf(n1, 0.25f, m2);
            H[c][j][3] = __builtin_fmaf(n3, 0.25f, m2);
        }
    }
    // ---- the four luma rows: row 4 P + r takes chroma window rows (0, 1) with fy = .75, (1, 2) with .25, (1, 2) with .75, (2, 3) with .25
    //      — top * (1 - fy) + bot * fy: the 3/4 product of window row 1 serves luma rows 0 and 1, that of row 2 serves rows 2 and 3
    // (c - 16/255) / 0.8784 as RN(a * y_hi + RN(a * y_lo)) with y_hi + y_lo = 1 / 0.8784 to 48 bits: the IEEE quotient for EVERY f32 a in
    // [-16/255, 1] — all 636 524 221 of them checked against the division (tools/check_div_by_constant.py), as unorm_of_byte's form is
    // for the 256 bytes.  One multiply and one fused multiply-add.
    constexpr float kc = 0.87843137254f;
    constexpr float rc_hi = 1.0f / kc, rc_lo = (float)(1.0 / (double)kc - (double)rc_hi);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = 4 * P + r;
        if (y >= h) break;
        const int j34 = r < 2 ? 1 : 2, j14 = r == 0 ? 0 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
        const u32 y4 = yrow[r];
        u32 px[4] = {0u, 0u, 0u, 0u}, r4 = 0u, g4 = 0u, b4 = 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (CV_ABL & 8) {
                const u32 b = (y4 >> (8 * i)) & 0xffu;
                px[i] = b | (__float_as_uint(H[0][j14][i] + H[1][j34][i]) & 0xffff00u) | 0xff000000u;
                r4 |= b << (8 * i); g4 |= (__float_as_uint(H[0][j14][i]) & 0xffu) << (8 * i); b4 |= (__float_as_uint(H[1][j34][i]) & 0xffu) << (8 * i);
                continue;
            }
            float u = __builtin_fmaf(H[0][j14][i], 0.25f, H[0][j34][i] * 0.75f);
            float v = __builtin_fmaf(H[1][j14][i], 0.25f, H[1][j34][i] * 0.75f);
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
            const u32 r8 = (u32)(int)(cv_clamp01(R) * 255.0f + 0.5f);
            const u32 g8 = (u32)(int)(cv_clamp01(G) * 255.0f + 0.5f);
            const u32 b8 = (u32)(int)(cv_clamp01(B) * 255.0f + 0.5f);
            if (RGB12) { r4 |= r8 << (8 * i); g4 |= g8 << (8 * i); b4 |= b8 << (8 * i); }
            else px[i] = r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
        }
        if ((CV_ABL & 1) && (r4 ^ g4 ^ b4 ^ px[0] ^ px[3]) != 0x12345677u) continue;  // (never equal in practice: the values stay live)
        if (RGB12) {
            u32 *d = (u32 *)(J.dst.ptr + cv_mad24((u32)y, J.dst.pitch, 12u * (u32)g));
            d[0] = r4; d[1] = g4; d[2] = b4;
        } else {
            *(uint4 *)(J.dst.ptr + cv_mad24((u32)y, J.dst.pitch, 16u * (u32)g)) = make_uint4(px[0], px[1], px[2], px[3]);
        }
    }
}

#endif  // __HIPCC__
