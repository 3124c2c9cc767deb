// VALU / LDS-gather / MFMA throughput of one SIMD against the number of waves on it (1..4): a workgroup of 256 * W threads per CU,
// every wave runs the same role.  Reports ns per iteration per wave and the aggregate instructions per microsecond per SIMD.
//   V: 128 independent v_fma_f32 (8 chains)   P: 128 v_perm_b32   G: 32 dependent ds_read_b32 gathers   I: 32 independent gathers
//   M: 16 v_mfma_f32_16x16x32_f16 (4 accumulators)   X: per iteration 96 fma + 24 independent gathers + 8 mfma (the ingest kernel's mix)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_occ.hip -o tools/ubench/build/valu_occ
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define FMA8 "v_fma_f32 %0, %0, %8, %0\nv_fma_f32 %1, %1, %8, %1\nv_fma_f32 %2, %2, %8, %2\nv_fma_f32 %3, %3, %8, %3\nv_fma_f32 %4, %4, %8, %4\nv_fma_f32 %5, %5, %8, %5\nv_fma_f32 %6, %6, %8, %6\nv_fma_f32 %7, %7, %8, %7\n"
#define PRM8 "v_perm_b32 %0, %0, %8, %8\nv_perm_b32 %1, %1, %8, %8\nv_perm_b32 %2, %2, %8, %8\nv_perm_b32 %3, %3, %8, %8\nv_perm_b32 %4, %4, %8, %8\nv_perm_b32 %5, %5, %8, %8\nv_perm_b32 %6, %6, %8, %8\nv_perm_b32 %7, %7, %8, %8\n"

template <int ROLE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a, unsigned b) {
    __shared__ unsigned lut[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lut[i] = i * 2654435761u;
    __syncthreads();
    float x[8];
    unsigned u[8];
    f32x4 acc[4];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 1e-3f + i; u[i] = threadIdx.x + i; }
    for (int i = 0; i < 4; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 fa, fb;
    for (int i = 0; i < 8; i++) { fa[i] = (_Float16)(threadIdx.x * 0.01f + i); fb[i] = (_Float16)(0.5f + i); }
    unsigned idx = threadIdx.x * 97u;
    for (int it = 0; it < iters; it++) {
        if (ROLE == 0) { REP16(asm volatile(FMA8 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a));) }
        if (ROLE == 1) { REP16(asm volatile(PRM8 : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(b));) }
        if (ROLE == 2) {
#pragma unroll
            for (int q = 0; q < 32; q++) idx = lut[(idx >> 3) & 1023] + q;
        }
        if (ROLE == 3) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                unsigned t[8];
#pragma unroll
                for (int e = 0; e < 8; e++) t[e] = lut[((idx >> 3) + 37 * e + q) & 1023];
#pragma unroll
                for (int e = 0; e < 8; e++) idx += t[e];
            }
        }
        if (ROLE == 4) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[q], 0, 0, 0);
        }
        if (ROLE == 5) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                unsigned t[6];
#pragma unroll
                for (int e = 0; e < 6; e++) t[e] = lut[((idx >> 3) + 37 * e + q) & 1023];
                REP4(asm volatile(FMA8 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a));)
                acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[q & 3], 0, 0, 0);
                acc[(q + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[(q + 1) & 3], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 6; e++) idx += t[e];
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += x[i] + (float)u[i];
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)idx;
}

template <int ROLE>
static void run(const char *name, int per_iter, float *out) {
    const int iters = 2000;
    for (int w = 1; w <= 4; w++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<ROLE><<<256, 256 * w>>>(out, 10, 1.0001f, 0x03020100u);
        hipEventRecord(e0);
        k<ROLE><<<256, 256 * w>>>(out, iters, 1.0001f, 0x03020100u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ns_iter = ms * 1e6 / iters;
        printf("%-2s waves/SIMD %d: %8.1f ns per iteration, %7.1f instr/us/SIMD (%d per iteration per wave), %.2f ns per instr per SIMD\n", name, w, ns_iter,
               per_iter * w / (ns_iter * 1e-3), per_iter, ns_iter / (per_iter * w));
    }
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<0>("V", 128, out);
    run<1>("P", 128, out);
    run<2>("G", 32, out);
    run<3>("I", 32, out);
    run<4>("M", 16, out);
    run<5>("X", 32 * 4 + 24 + 8, out);
    hipDeviceSynchronize();
    return 0;
}
