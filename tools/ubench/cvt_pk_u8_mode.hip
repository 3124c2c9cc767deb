// Does v_cvt_pk_u8_f32 follow the wave's FP32 rounding mode (MODE[1:0])?  Under round-toward-zero it would be the truncation the reference's
// unorm8 store wants (floor(x * 255 + 0.5) on non-negative operands) with the clamp to [0, 255] and the byte insert in the same instruction:
// three instructions per stored value (multiply, add, convert) instead of five (median, multiply, add, convert, shift-or).
// hipcc --offload-arch=gfx950 -O2 -o /tmp/cvt_mode tools/ubench/cvt_pk_u8_mode.hip && /tmp/cvt_mode
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
__global__ void k(const float *in, unsigned *out, int n, int mode) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    unsigned r;
    if (mode == 0) {
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(x), "v"(0xaabbccddu));
    } else {
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\tv_cvt_pk_u8_f32 %0, %1, 1, %2\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "=v"(r) : "v"(x), "v"(0xaabbccddu));
    }
    out[i] = r;
}
int main() {
    // every f32 in [0, 300) would be 1.1e9 values: sample all values within +-4 ulp of every k and k + 0.5, plus a coarse sweep and specials
    const int N = 1 << 22;
    float *h = (float *)malloc(N * 4);
    int n = 0;
    for (int k = 0; k <= 257 && n < N - 64; k++)
        for (int half = 0; half < 2; half++) {
            float c = (float)k + 0.5f * half;
            unsigned b; memcpy(&b, &c, 4);
            for (int d = -4; d <= 4; d++) { unsigned bb = b + d; if (k == 0 && half == 0 && d < 0) continue; memcpy(&h[n++], &bb, 4); }
        }
    for (int i = 0; n < N - 8; i++) h[n++] = -2.0f + 300.0f * (float)i / (float)(N - n + i + 1);
    h[n++] = -1e30f; h[n++] = 1e30f; h[n++] = -0.0f; h[n++] = __builtin_inff(); h[n++] = -__builtin_inff();
    float *d; unsigned *o, *ho = (unsigned *)malloc(N * 4);
    hipMalloc(&d, N * 4); hipMalloc(&o, N * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; mode++) {
        k<<<(n + 255) / 256, 256>>>(d, o, n, mode);
        hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
        long bad_rne = 0, bad_trunc = 0, bad_rest = 0;
        for (int i = 0; i < n; i++) {
            const unsigned got = (ho[i] >> 8) & 0xff;
            const float x = h[i];
            const float cl = x < 0.f ? 0.f : (x > 255.f ? 255.f : x);
            const unsigned rne = (unsigned)nearbyintf(cl), tr = (unsigned)cl;
            if (got != rne) bad_rne++;
            if (got != tr) bad_trunc++;
            if ((ho[i] & 0xffff00ffu) != 0xaabb00ddu) bad_rest++;
        }
        printf("mode %s: %d values, differs from round-to-nearest-even on %ld, from truncation on %ld, other bytes disturbed on %ld\n", mode ? "RTZ" : "default", n, bad_rne, bad_trunc, bad_rest);
    }
    return 0;
}
