// Semantics of v_cvt_pk_u8_f32 on gfx950 (rounding, saturation) — the ingest kernel wants floor(clamp(x, 0, 255.99)).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xaabbccddu);
}
int main() {
    float h[] = {-5.f, -0.4f, 0.f, 0.3f, 0.5f, 0.7f, 1.f, 1.49f, 1.5f, 1.51f, 2.5f, 3.5f, 254.4f, 254.5f, 254.6f, 255.f, 255.4f, 255.5f, 256.f, 300.f, 1e9f, __builtin_nanf("")};
    const int n = sizeof(h) / 4;
    float *d; unsigned *o, ho[64];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%12.3f -> %08x (byte1 = %u)\n", h[i], ho[i], (ho[i] >> 8) & 0xff);
    return 0;
}
