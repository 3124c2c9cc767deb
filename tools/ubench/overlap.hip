// Do the pipes of one SIMD overlap across waves on gfx950?  Two waves per SIMD (an 8-wave workgroup per CU: waves w and w + 4 share
// a SIMD), each wave runs one role per test: V = 128 independent v_fma_f32, M = 16 v_mfma_f32_16x16x32_f16 (4 accumulators),
// L = 32 ds_read_b32 gathers, H = 32 half-rate VALU (v_perm_b32), idle.  Reports ns per iteration: if V|M takes max(V|idle, idle|M) the
// pipes overlap, if it takes the sum they do not.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/overlap.hip -o tools/ubench/build/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

__device__ __forceinline__ void role_valu(float *x, float a) {
    REP16(asm volatile("v_fma_f32 %0, %0, %8, %0\nv_fma_f32 %1, %1, %8, %1\nv_fma_f32 %2, %2, %8, %2\nv_fma_f32 %3, %3, %8, %3\nv_fma_f32 %4, %4, %8, %4\nv_fma_f32 %5, %5, %8, %5\nv_fma_f32 %6, %6, %8, %6\nv_fma_f32 %7, %7, %8, %7"
                       : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a));)
}
__device__ __forceinline__ void role_half(unsigned *u, unsigned b) {
    REP4(asm volatile("v_perm_b32 %0, %0, %8, %8\nv_perm_b32 %1, %1, %8, %8\nv_perm_b32 %2, %2, %8, %8\nv_perm_b32 %3, %3, %8, %8\nv_perm_b32 %4, %4, %8, %8\nv_perm_b32 %5, %5, %8, %8\nv_perm_b32 %6, %6, %8, %8\nv_perm_b32 %7, %7, %8, %8"
                      : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(b));)
}
__device__ __forceinline__ void role_mfma(f32x4 *acc, f16x8 a, f16x8 b) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
}
// roles: 0 idle, 1 V, 2 M, 3 L, 4 H, 5 mixed V + M in one wave (64 fma + 16 mfma interleaved by the compiler)
template <int RA, int RB>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, unsigned b) {
    __shared__ unsigned lut[1024];
    for (int i = threadIdx.x; i < 1024; i += 512) lut[i] = i * 2654435761u;
    __syncthreads();
    const int role = (threadIdx.x >> 8) ? RB : RA;  // waves 0-3 / 4-7
    float x[8];
    unsigned u[8];
    f32x4 acc[4];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 1e-3f + i; u[i] = threadIdx.x + i; }
    for (int i = 0; i < 4; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 fa, fb;
    for (int i = 0; i < 8; i++) { fa[i] = (_Float16)(threadIdx.x * 0.01f + i); fb[i] = (_Float16)(0.5f + i); }
    unsigned idx = threadIdx.x * 97u;
    for (int it = 0; it < iters; it++) {
        if (role == 1) role_valu(x, a);
        if (role == 2) role_mfma(acc, fa, fb);
        if (role == 3) {
#pragma unroll
            for (int k = 0; k < 32; k++) { idx = lut[(idx >> 3) & 1023] + k; }
        }
        if (role == 4) role_half(u, b);
        if (role == 5) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[k], 0, 0, 0);
                    asm volatile("v_fma_f32 %0, %0, %4, %0\nv_fma_f32 %1, %1, %4, %1\nv_fma_f32 %2, %2, %4, %2\nv_fma_f32 %3, %3, %4, %3" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(a));
                }
            }
        }
        if (role == 6) {  // L gathers that do not depend on each other (8 in flight)
            unsigned s = 0;
#pragma unroll
            for (int k = 0; k < 32; k++) s += lut[(idx + 37u * k) & 1023];
            idx += s;
        }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += x[i] + (float)u[i];
    for (int i = 0; i < 4; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + (float)idx;
}
template <int RA, int RB>
double run(const char *name) {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<RA, RB><<<256, 512>>>(out, 10, 1.0001f, 0x01020304u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<RA, RB><<<256, 512>>>(out, iters, 1.0001f, 0x01020304u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s %8.1f ns per iteration\n", name, ms * 1e6 / iters);
    hipFree(out);
    return ms * 1e6 / iters;
}
int main() {
    run<1, 0>("V | idle   (128 fma)");
    run<0, 2>("idle | M   (16 mfma)");
    run<1, 2>("V | M");
    run<1, 1>("V | V");
    run<2, 2>("M | M");
    run<5, 0>("VM mixed | idle (64 fma + 16 mfma)");
    run<5, 5>("VM mixed | VM mixed");
    run<3, 0>("L dependent | idle (32 gathers)");
    run<6, 0>("L independent | idle (32 gathers)");
    run<6, 6>("L indep | L indep");
    run<1, 6>("V | L indep");
    run<4, 0>("H | idle  (32 v_perm)");
    run<4, 4>("H | H");
    run<1, 4>("V | H");
    run<2, 6>("M | L indep");
    return 0;
}
