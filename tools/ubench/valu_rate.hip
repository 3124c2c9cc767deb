// Issue rate of the VALU instructions the ingest kernel leans on (gfx950): cycles per wave64 instruction per SIMD with W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_INNER 256
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
template <int OP>
__global__ void k(float *out, int iters, float a, unsigned b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3, dc = a;  // (64-bit register pairs for v_pk_fma_f32)
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    for (int i = 0; i < iters; i++) {
        // 16 x 8 independent instructions per iteration
#define OPS(ASM, CON) REP16(asm volatile(ASM "\n" ASM2(ASM) : CON);)
        if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %0\nv_fma_f32 %1, %1, %8, %1\nv_fma_f32 %2, %2, %8, %2\nv_fma_f32 %3, %3, %8, %3\nv_fma_f32 %4, %4, %8, %4\nv_fma_f32 %5, %5, %8, %5\nv_fma_f32 %6, %6, %8, %6\nv_fma_f32 %7, %7, %8, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));) }
        if (OP == 1) { REP16(asm volatile("v_med3_f32 %0, %0, %8, 1.0\nv_med3_f32 %1, %1, %8, 1.0\nv_med3_f32 %2, %2, %8, 1.0\nv_med3_f32 %3, %3, %8, 1.0\nv_med3_f32 %4, %4, %8, 1.0\nv_med3_f32 %5, %5, %8, 1.0\nv_med3_f32 %6, %6, %8, 1.0\nv_med3_f32 %7, %7, %8, 1.0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));) }
        if (OP == 2) { REP16(asm volatile("v_cvt_u32_f32 %0, %0\nv_cvt_u32_f32 %1, %1\nv_cvt_u32_f32 %2, %2\nv_cvt_u32_f32 %3, %3\nv_cvt_u32_f32 %4, %4\nv_cvt_u32_f32 %5, %5\nv_cvt_u32_f32 %6, %6\nv_cvt_u32_f32 %7, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));) }
        if (OP == 3) { REP16(asm volatile("v_dot4_u32_u8 %0, %0, %8, 0\nv_dot4_u32_u8 %1, %1, %8, 0\nv_dot4_u32_u8 %2, %2, %8, 0\nv_dot4_u32_u8 %3, %3, %8, 0\nv_dot4_u32_u8 %4, %4, %8, 0\nv_dot4_u32_u8 %5, %5, %8, 0\nv_dot4_u32_u8 %6, %6, %8, 0\nv_dot4_u32_u8 %7, %7, %8, 0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 4) { REP16(asm volatile("v_perm_b32 %0, %0, %8, %8\nv_perm_b32 %1, %1, %8, %8\nv_perm_b32 %2, %2, %8, %8\nv_perm_b32 %3, %3, %8, %8\nv_perm_b32 %4, %4, %8, %8\nv_perm_b32 %5, %5, %8, %8\nv_perm_b32 %6, %6, %8, %8\nv_perm_b32 %7, %7, %8, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 5) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 2, %8\nv_lshl_add_u32 %1, %1, 2, %8\nv_lshl_add_u32 %2, %2, 2, %8\nv_lshl_add_u32 %3, %3, 2, %8\nv_lshl_add_u32 %4, %4, 2, %8\nv_lshl_add_u32 %5, %5, 2, %8\nv_lshl_add_u32 %6, %6, 2, %8\nv_lshl_add_u32 %7, %7, 2, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 6) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %0\nv_cvt_f32_ubyte0 %1, %1\nv_cvt_f32_ubyte0 %2, %2\nv_cvt_f32_ubyte0 %3, %3\nv_cvt_f32_ubyte0 %4, %4\nv_cvt_f32_ubyte0 %5, %5\nv_cvt_f32_ubyte0 %6, %6\nv_cvt_f32_ubyte0 %7, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));) }
        if (OP == 7) { REP16(asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 8) { REP16(asm volatile("v_cvt_pk_u8_f32 %0, %8, 1, %0\nv_cvt_pk_u8_f32 %1, %8, 1, %1\nv_cvt_pk_u8_f32 %2, %8, 1, %2\nv_cvt_pk_u8_f32 %3, %8, 1, %3\nv_cvt_pk_u8_f32 %4, %8, 1, %4\nv_cvt_pk_u8_f32 %5, %8, 1, %5\nv_cvt_pk_u8_f32 %6, %8, 1, %6\nv_cvt_pk_u8_f32 %7, %8, 1, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(a));) }
        if (OP == 10) { REP16(asm volatile("v_pk_fma_f16 %0, %0, %8, %0\nv_pk_fma_f16 %1, %1, %8, %1\nv_pk_fma_f16 %2, %2, %8, %2\nv_pk_fma_f16 %3, %3, %8, %3\nv_pk_fma_f16 %4, %4, %8, %4\nv_pk_fma_f16 %5, %5, %8, %5\nv_pk_fma_f16 %6, %6, %8, %6\nv_pk_fma_f16 %7, %7, %8, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 11) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\nv_pk_fma_f32 %1, %1, %4, %1\nv_pk_fma_f32 %2, %2, %4, %2\nv_pk_fma_f32 %3, %3, %4, %3\nv_pk_fma_f32 %0, %0, %4, %0\nv_pk_fma_f32 %1, %1, %4, %1\nv_pk_fma_f32 %2, %2, %4, %2\nv_pk_fma_f32 %3, %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dc));) }
        if (OP == 9) { REP16(asm volatile("v_mul_f32 %0, %0, %8\nv_mul_f32 %1, %1, %8\nv_mul_f32 %2, %2, %8\nv_mul_f32 %3, %3, %8\nv_mul_f32 %4, %4, %8\nv_mul_f32 %5, %5, %8\nv_mul_f32 %6, %6, %8\nv_mul_f32 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(d0 + d1 + d2 + d3) + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
}
template <int OP>
void run(const char *name, int waves_per_simd) {
    float *out;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    const int iters = 2000, blocks = 256, threads = 4 * waves_per_simd * 64;  // one block per CU
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, threads>>>(out, 10, 1.0001f, 0x01020304u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, threads>>>(out, iters, 1.0001f, 0x01020304u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)iters * 128 * waves_per_simd;
    printf("%-18s waves/SIMD %d: %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w); run<9>("v_mul_f32", w); run<1>("v_med3_f32", w); run<2>("v_cvt_u32_f32", w); run<3>("v_dot4_u32_u8", w); run<4>("v_perm_b32", w);
        run<5>("v_lshl_add_u32", w); run<6>("v_cvt_f32_ubyte0", w); run<7>("v_add_u32", w); run<8>("v_cvt_pk_u8_f32", w);
        run<10>("v_pk_fma_f16", w); run<11>("v_pk_fma_f32", w);
    }
    return 0;
}
