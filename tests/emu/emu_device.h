// TEST INFRASTRUCTURE — not part of the product.  CPU restatements of the gfx950 instructions the ingest kernels use, for the
// lane emulator of tests/emu: one host thread per lane, wave-collective instructions (MFMA, DPP) meet at a per-wave barrier.
// Semantics follow the CDNA4 ISA: v_perm_b32 (byte select from {S0, S1}), v_alignbyte_b32, v_dot4_u32_u8, v_med3_f32,
// v_mfma_f32_16x16x32_f16 (A: lane l holds A[l & 15][8 (l >> 4) + e]; B: B[8 (l >> 4) + e][l & 15]; C/D: D[4 (l >> 4) + r][l & 15]).
#pragma once
#include <pthread.h>

struct EmuWave {
    pthread_barrier_t bar;
    _Float16 A[16][32], B[32][16];
    int xchg[64];
};
struct EmuBlock {
    pthread_barrier_t bar;
    EmuWave waves[16];
    unsigned char *smem;
};
extern EmuBlock *emu_blk;
extern thread_local unsigned char *emu_smem;

static inline unsigned dev_perm(unsigned hi, unsigned lo, unsigned sel) {
    const unsigned long long src = ((unsigned long long)hi << 32) | lo;
    unsigned out = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned s = (sel >> (8 * i)) & 0xffu;
        unsigned b;
        if (s <= 7) b = (unsigned)(src >> (8 * s)) & 0xffu;
        else if (s == 0x0c) b = 0;
        else if (s >= 0x0d) b = 0xff;
        else b = 0;  // (sign-replicating selectors 8..11: unused by the kernels)
        out |= b << (8 * i);
    }
    return out;
}
static inline unsigned dev_alignbyte(unsigned hi, unsigned lo, unsigned sh) {
    const unsigned long long src = ((unsigned long long)hi << 32) | lo;
    return (unsigned)(src >> (8 * (sh & 3)));
}
static inline unsigned dev_udot4(unsigned a, unsigned b, unsigned c) {
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
    return c;
}
static inline uint32_t dev_mad24(uint32_t a, uint32_t b, uint32_t c) { return (a & 0xffffffu) * (b & 0xffffffu) + c; }
static inline float dev_fmed3(float a, float b, float c) {
    const float lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
template <typename T>
static inline T dev_readfirstlane(T x) { return x; }  // (callers pass wave-uniform values)
static inline void dev_wait_vmcnt0() {}
static inline void dev_sched_barrier() {}
static inline void dev_wave_lds_sync() { pthread_barrier_wait(&emu_blk->waves[threadIdx.x >> 6].bar); }
static inline unsigned dev_lds_u32(unsigned byte_offset) { unsigned v; memcpy(&v, emu_smem + byte_offset, 4); return v; }

typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
static inline emu_f32x4 dev_mfma_16x16x32_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x4 c) {
    const int tid = (int)threadIdx.x, lane = tid & 63;
    EmuWave &W = emu_blk->waves[tid >> 6];
    for (int e = 0; e < 8; e++) {
        W.A[lane & 15][8 * (lane >> 4) + e] = a[e];
        W.B[8 * (lane >> 4) + e][lane & 15] = b[e];
    }
    pthread_barrier_wait(&W.bar);
    emu_f32x4 d;
    for (int r = 0; r < 4; r++) {
        const int i = 4 * (lane >> 4) + r, j = lane & 15;
        float s = c[r];
        for (int k = 0; k < 32; k++) s += (float)W.A[i][k] * (float)W.B[k][j];
        d[r] = s;
    }
    pthread_barrier_wait(&W.bar);
    return d;
}
static inline int dev_mov_dpp_quad_swap(int x) {  // quad_perm [1, 0, 3, 2]
    const int tid = (int)threadIdx.x, lane = tid & 63;
    EmuWave &W = emu_blk->waves[tid >> 6];
    W.xchg[lane] = x;
    pthread_barrier_wait(&W.bar);
    const int r = W.xchg[lane ^ 1];
    pthread_barrier_wait(&W.bar);
    return r;
}
