// smr_yuv_fast.h — the output conversion's fast path: RGBA bytes -> Y'CbCr bytes (rgba_to_yuv.wgsl:26-54, rgba_to_nv12.wgsl:24-52)
// with three fused multiply-adds per plane value instead of the reference's ~20 operations, and a guard that says when that is not enough.
//
// The reference computes, in f32 and in a fixed order, byte / 255 per channel, a 3-term dot product, the range compression
// (x * 219/255 + 16/255 or (x + .5) * 224/255 + 16/255), x * 255 + .5 and a truncation (yuv_byte() / unorm_of_byte(), smr_convert_dev.h).
// In real arithmetic that is an affine function of the bytes; its f32 evaluation deviates from the real value by the sequence's own
// roundings — at most 4.6e-5 code units (measured over the COMPLETE domain, below).  The fast path evaluates the affine function directly:
//
//      x = fma(R, k_r, fma(G, k_g, fma(B, k_b, k_0)))         k_0 carries the + 0.5 of the rounding and + DELTA
//      byte = trunc(x)          flag = fract(x) < 2 DELTA     <=>  the real value lies within DELTA of a code boundary
//
// A flagged value is recomputed with the reference sequence by the caller (a rare, divergent branch: 0.024 % of the values); an unflagged
// one is the reference's byte.  That is a theorem about finitely many inputs, and tools/check_yuv_fast.cpp proves it by enumeration with
// THIS header: luma over all 2^24 (R, G, B); chroma — a function of the 2x2 block's three channel means — over every combination of the
// byte sums (1021^3) times every f32 value the mean ((a + b) + (c + d)) / 4 of byte / 255 values can take for that sum (at most 3).
// For chroma R, G, B are the block's byte SUMS (exact in f32: at most 1020), the / 4 of the mean is folded into the coefficients.
//
// Included by the kernels (smr_convert_dev.h), the lane emulator (tests/emu) and the checker: one source.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__) && !defined(SMR_EMU)
#define YF_FN __device__ __forceinline__
#else
#define YF_FN static inline
#endif

namespace yuvfast {

constexpr double DELTA = 1.0 / 8192.0;  // half width of the guard band, code units (2^-13: 2.7 x the largest deviation of the sequence)

struct K {
    float kr, kg, kb, k0;
};
// plane 0: Y' from a pixel's bytes | 1, 2: Cb, Cr from a 2x2 block's byte sums
constexpr K constants(int plane) {
    const double c16 = (double)(16.0f / 255.0f);
    const double ky = (double)0.85882352941f, kc = (double)0.87843137254f;
    const double cr = plane == 0 ? (double)0.2126f : plane == 1 ? (double)-0.1146f : 0.5;
    const double cg = plane == 0 ? (double)0.7152f : plane == 1 ? (double)-0.3854f : (double)-0.4542f;
    const double cb = plane == 0 ? (double)0.0722f : plane == 1 ? 0.5 : (double)-0.0458f;
    const double off = plane == 0 ? 0.0 : 0.5, k = plane == 0 ? ky : kc, div = plane == 0 ? 1.0 : 4.0;
    return K{(float)(cr * k / div), (float)(cg * k / div), (float)(cb * k / div), (float)((off * k + c16) * 255.0 + 0.5 + DELTA)};
}

YF_FN float fract(float x) {
#if defined(__HIPCC__) && !defined(SMR_EMU)
    return __builtin_amdgcn_fractf(x);  // v_fract_f32
#else
    return x - floorf(x);  // (exact: x is far above the subnormals and below 2^23)
#endif
}

// r, g, b: the bytes (plane 0) / the block's byte sums (planes 1, 2) as floats.  Returns the plane's byte; *flag: recompute it exactly.
template <int PLANE>
YF_FN uint32_t convert(float r, float g, float b, bool *flag) {
    constexpr K k = constants(PLANE);
    const float x = __builtin_fmaf(r, k.kr, __builtin_fmaf(g, k.kg, __builtin_fmaf(b, k.kb, k.k0)));
    *flag = fract(x) < (float)(2.0 * DELTA);
    return (uint32_t)x;  // v_cvt_u32_f32: truncation (x lies in [16, 241])
}

}  // namespace yuvfast
