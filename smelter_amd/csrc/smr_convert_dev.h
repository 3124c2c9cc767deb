// smr_convert_dev.h — colour maths shared by the converter kernels and the fused kernels.
#pragma once

#include "smr_internal.h"
#include "smr_yuv_fast.h"

#ifdef __HIPCC__

// planar_yuv_to_rgba.wgsl:45-57: limited->full range (unless J), BT.709 matrix, clamp, unorm8 store.
__device__ __forceinline__ u32 yuv_to_rgb_px(float y, float u, float v, bool full) {
    if (!full) {
        y = clampf((y - (16.0f / 255.0f)) / 0.85882352941f, 0.0f, 1.0f);
        u = clampf((u - (16.0f / 255.0f)) / 0.87843137254f, 0.0f, 1.0f);
        v = clampf((v - (16.0f / 255.0f)) / 0.87843137254f, 0.0f, 1.0f);
    }
    float r = y + 1.5748f * (v - 0.5f);
    float g = y - 0.1873f * (u - 0.5f) - 0.4681f * (v - 0.5f);
    float b = y + 1.8556f * (u - 0.5f);
    return unorm8(r) | (unorm8(g) << 8) | (unorm8(b) << 16) | 0xff000000u;
}

// The same with the two range divisions as q0 = a * RN(1 / b), q = fma(fma(-q0, b, a), RN(1 / b), q0): the correctly rounded quotient
// (Markstein's correction step; neither divisor's significand is all ones) — checked against the division on 2 x 10^8 random operands
// per divisor and on every byte, and on the GPU against k_yuv_to_rgba frame by frame (tests/test_gpu_parity.py).
// The clamps are one v_med3_f32 each (operands are finite; a -0 where clampf gives +0 is absorbed by the sums that follow).
#ifndef SMR_EMU  // (k_yuv_to_rgba_batch only: not part of the kernels the lane emulator compiles)
__device__ __forceinline__ u32 yuv_to_rgb_px_cr(float y, float u, float v, bool full) {
    if (!full) {
        constexpr float ky = 0.85882352941f, kc = 0.87843137254f;
        const float ry = 1.0f / ky, rc = 1.0f / kc;
        const float ay = y - (16.0f / 255.0f), au = u - (16.0f / 255.0f), av = v - (16.0f / 255.0f);
        const float qy = ay * ry, qu = au * rc, qv = av * rc;
        y = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_fmaf(-qy, ky, ay), ry, qy), 0.0f, 1.0f);
        u = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_fmaf(-qu, kc, au), rc, qu), 0.0f, 1.0f);
        v = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_fmaf(-qv, kc, av), rc, qv), 0.0f, 1.0f);
    }
    const float r = y + 1.5748f * (v - 0.5f);
    const float g = y - 0.1873f * (u - 0.5f) - 0.4681f * (v - 0.5f);
    const float b = y + 1.8556f * (u - 0.5f);
    const u32 r8 = (u32)(int)(__builtin_amdgcn_fmed3f(r, 0.0f, 1.0f) * 255.0f + 0.5f);
    const u32 g8 = (u32)(int)(__builtin_amdgcn_fmed3f(g, 0.0f, 1.0f) * 255.0f + 0.5f);
    const u32 b8 = (u32)(int)(__builtin_amdgcn_fmed3f(b, 0.0f, 1.0f) * 255.0f + 0.5f);
    return r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
}
#endif

// rgba_to_yuv.wgsl:26-54 — one plane component from a (gamma-encoded, raw-byte) RGBA value.
__device__ __forceinline__ float yuv_component(float4 c, int plane) {
    float comp;
    if (plane == 0) {
        float y = c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f;
        comp = (y * 0.85882352941f) + (16.0f / 255.0f);
    } else if (plane == 1) {
        float u = c.x * -0.1146f + c.y * -0.3854f + c.z * 0.5f;
        comp = ((u + 0.5f) * 0.87843137254f) + (16.0f / 255.0f);
    } else {
        float v = c.x * 0.5f + c.y * -0.4542f + c.z * -0.0458f;
        comp = ((v + 0.5f) * 0.87843137254f) + (16.0f / 255.0f);
    }
    return clampf(comp, 0.0f, 1.0f);
}

__device__ __forceinline__ float4 unpack_unorm(u32 p) {
    return make_float4((float)(p & 0xff) / 255.0f, (float)((p >> 8) & 0xff) / 255.0f, (float)((p >> 16) & 0xff) / 255.0f,
                       (float)(p >> 24) / 255.0f);
}

// ---- the output conversion for inputs that are bytes (the fused kernels: every channel is byte / 255), with the operations that
//      cannot change the result left out.  Same values as yuv_component() / unorm8() / unpack_unorm() bit for bit:
//  * byte / 255: RN(a * rb + a * -2^-33) with rb = RN(1 / 255) is the IEEE quotient for all 256 bytes (checked exhaustively,
//    tests/test_oracle_golden.py) — one multiply by a power of two (exact) and one FMA instead of a division;
//  * the clamps of yuv_component() and unorm8() never act: with channels in [0, 1], y' = dot(c, (0.2126, 0.7152, 0.0722)) <= 1 + 2 ulp
//    gives 0.0627 <= y' * 0.8588 + 16/255 <= 0.9216, and |u'|, |v'| <= 0.5 (the negative and the positive coefficients each sum to
//    0.5) gives 0.0627 <= (u' + 0.5) * 0.8784 + 16/255 <= 0.9412: inside (0, 1), never NaN.
__device__ __forceinline__ float unorm_of_byte(u32 b) {
    const float a = (float)b;
    return __builtin_fmaf(a, 1.0f / 255.0f, a * -1.1641532182693481e-10f);
}
__device__ __forceinline__ u32 yuv_byte(float r, float g, float b, int plane) {
    float comp;
    if (plane == 0) {
        const float y = r * 0.2126f + g * 0.7152f + b * 0.0722f;
        comp = (y * 0.85882352941f) + (16.0f / 255.0f);
    } else if (plane == 1) {
        const float u = r * -0.1146f + g * -0.3854f + b * 0.5f;
        comp = ((u + 0.5f) * 0.87843137254f) + (16.0f / 255.0f);
    } else {
        const float v = r * 0.5f + g * -0.4542f + b * -0.0458f;
        comp = ((v + 0.5f) * 0.87843137254f) + (16.0f / 255.0f);
    }
    return (u32)(int)(comp * 255.0f + 0.5f);
}

// ---- the same bytes through the fast path (smr_yuv_fast.h: three FMAs per value, proven equal to the sequence above wherever its guard
//      flag is clear — tools/check_yuv_fast.cpp, the complete domain) with the sequence itself behind the flag: a divergent branch that
//      2.4 values in 10 000 take.  The callers hold the bytes as floats already (v_cvt_f32_ubyteN: extract + convert in one instruction).
// (the compiler otherwise speculates the side-effect-free reference sequence and selects: both paths on every value — seen in the first build's
//  instruction counts; an empty volatile asm cannot be speculated, so the sequence stays behind s_cbranch_execz)
#ifdef SMR_EMU
#define YUV_SLOW_PATH_STAYS_A_BRANCH() do { } while (0)
#else
#define YUV_SLOW_PATH_STAYS_A_BRANCH() asm volatile("; reference sequence (guard flag set)" ::: "memory")
#endif
// Y' of a pixel: px = the RGBA8 bytes, fr / fg / fb = its colour bytes as floats
__device__ __forceinline__ u32 yuv_luma_byte(u32 px, float fr, float fg, float fb) {
    bool flag;
    u32 y = yuvfast::convert<0>(fr, fg, fb, &flag);
    if (flag) {
        YUV_SLOW_PATH_STAYS_A_BRANCH();
        y = yuv_byte(unorm_of_byte(px & 0xffu), unorm_of_byte((px >> 8) & 0xffu), unorm_of_byte((px >> 16) & 0xffu), 0);
    }
    return y;
}
// (Cb | Cr << 8) of a 2x2 block: pa, pb = the upper row's pixels, pc, pd = the lower row's; sr / sg / sb = the block's byte sums as floats
// (exact: sums of integers up to 1020 in any order).  The reference's mean is ((a + b) + (c + d)) / 4 of the byte / 255 values.
__device__ __forceinline__ u32 yuv_chroma_bytes(u32 pa, u32 pb, u32 pc, u32 pd, float sr, float sg, float sb) {
    bool f1, f2;
    u32 u = yuvfast::convert<1>(sr, sg, sb, &f1), v = yuvfast::convert<2>(sr, sg, sb, &f2);
    if (f1 || f2) {
        YUV_SLOW_PATH_STAYS_A_BRANCH();
        const float mr = ((unorm_of_byte(pa & 0xffu) + unorm_of_byte(pb & 0xffu)) + (unorm_of_byte(pc & 0xffu) + unorm_of_byte(pd & 0xffu))) * 0.25f;
        const float mg = ((unorm_of_byte((pa >> 8) & 0xffu) + unorm_of_byte((pb >> 8) & 0xffu)) + (unorm_of_byte((pc >> 8) & 0xffu) + unorm_of_byte((pd >> 8) & 0xffu))) * 0.25f;
        const float mb = ((unorm_of_byte((pa >> 16) & 0xffu) + unorm_of_byte((pb >> 16) & 0xffu)) + (unorm_of_byte((pc >> 16) & 0xffu) + unorm_of_byte((pd >> 16) & 0xffu))) * 0.25f;
        u = yuv_byte(mr, mg, mb, 1);
        v = yuv_byte(mr, mg, mb, 2);
    }
    return u | (v << 8);
}

#endif
