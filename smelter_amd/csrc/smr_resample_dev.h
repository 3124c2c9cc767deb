// smr_resample_dev.h — Lanczos3 weight table shared by the general pass kernel and the fused
// ingest+resample kernel, so both evaluate exactly the same f32 sequence
// (smelter-render/src/transformations/layout/resample.wgsl:44-86).
#pragma once

#include "smr_internal.h"

constexpr int MAX_TAPS = 32;  // taps = ceil(6 * max(scale,1)) + 1 <= 25 once the box pre-reduce caps scale at 4

#ifdef __HIPCC__

// first source texel of output coordinate `out_coord` (resample.wgsl:47-49) — the same f32 sequence as lanczos_weights;
// host + device: the host sizes the fused kernels' footprints with it
__host__ __device__ __forceinline__ int lanczos_first(int out_coord, float scale, float offset) {
    float kernel_scale = scale > 1.0f ? scale : 1.0f;
    float support = 3.0f * kernel_scale;
    float center = offset + ((float)out_coord + 0.5f) * scale - 0.5f;
    return (int)ceilf(center - support);
}

__device__ __forceinline__ int lanczos_taps(float scale) {
    float kernel_scale = scale > 1.0f ? scale : 1.0f;
    int taps = (int)ceilf(2.0f * (3.0f * kernel_scale)) + 1;
    return taps > MAX_TAPS ? MAX_TAPS : taps;
}

// Weights of output coordinate `out_coord`: w[0..taps), returns first source index; *wsum = sum of weights.
__device__ __forceinline__ int lanczos_weights(int out_coord, float scale, float offset, int taps, float *w, float *wsum) {
    const float PI = 3.14159265359f;
    float kernel_scale = scale > 1.0f ? scale : 1.0f;
    float inv_k = 1.0f / kernel_scale;
    float support = 3.0f * kernel_scale;
    float center = offset + ((float)out_coord + 0.5f) * scale - 0.5f;
    float first = ceilf(center - support);
    float x0 = (first - center) * inv_k;
    float s1 = sinf(PI * x0), c1 = cosf(PI * x0);
    float s3 = sinf(PI * x0 / 3.0f), c3 = cosf(PI * x0 / 3.0f);
    float sd1 = sinf(PI * inv_k), cd1 = cosf(PI * inv_k);
    float sd3 = sinf(PI * inv_k / 3.0f), cd3 = cosf(PI * inv_k / 3.0f);
    float sum = 0.0f;
    for (int t = 0; t < taps; t++) {
        float xx = x0 + (float)t * inv_k;
        float weight = 0.0f;
        if (fabsf(xx) < 1e-5f) weight = 1.0f;
        else if (fabsf(xx) < 3.0f) weight = 3.0f * s1 * s3 / (PI * PI * xx * xx);
        w[t] = weight;
        sum = sum + weight;
        float ns1 = s1 * cd1 + c1 * sd1;
        c1 = c1 * cd1 - s1 * sd1;
        s1 = ns1;
        float ns3 = s3 * cd3 + c3 * sd3;
        c3 = c3 * cd3 - s3 * sd3;
        s3 = ns3;
    }
    *wsum = sum;
    return (int)first;
}

#endif
