// TEST INFRASTRUCTURE — not part of the product.  Minimal stand-in for <hip/hip_runtime.h> so that the kernel headers of
// smelter_amd/csrc can be compiled for the CPU by tests/emu (one thread per lane, see tests/emu/README.md).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define __HIPCC__ 1
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define hipLaunchKernelGGL(...) static_assert(false, "no launches in the emulator")

struct dim3 {
    unsigned x = 1, y = 1, z = 1;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }

extern thread_local dim3 threadIdx, blockIdx;
extern dim3 gridDim, blockDim;
void __syncthreads();

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }

typedef void *hipStream_t;
typedef void *hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
static inline hipError_t hipSetDevice(int) { return hipSuccess; }

// what the compositor kernels use beyond the above (tests/emu/emu_compose.cpp): LDS / global atomics between a workgroup's host threads,
// the 24-bit multiply, count-leading-zeros, and the scalar broadcast (callers pass wave-uniform values)
template <typename T>
static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <typename T>
static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <typename T>
static inline T atomicMax(T *p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
#define __builtin_amdgcn_readfirstlane(x) (x)
