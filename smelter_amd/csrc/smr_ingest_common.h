// smr_ingest_common.h — pieces of the matrix-core ingest kernel (smr_ingest_wave.h) that its fused-conversion builds use:
// the Y'CbCr -> linear (hi, lo) f16-pair conversion of one 4x1 pixel block (planar_yuv_to_rgba.wgsl:35-58 + the sRGB decode of the
// node texture's view), the direct-output record, and thin names for the gfx950 builtins the kernels use.
//
// The builtins go through dev_* names so that tests/emu can run the very same kernel source on the CPU, one thread per lane
// (tests/emu/README.md): there SMR_EMU is defined and the names resolve to plain C++ restatements of the instructions.  The
// product build never defines SMR_EMU; there is no CPU path in the library.
#pragma once

#include "smr_convert_dev.h"
#include "smr_resample_dev.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef SMR_EMU
#include "emu_device.h"  // tests/emu: dev_* as functions, emu_smem
#else
#define dev_perm(hi, lo, sel) __builtin_amdgcn_perm((hi), (lo), (sel))
#define dev_alignbyte(hi, lo, sh) __builtin_amdgcn_alignbyte((hi), (lo), (sh))
#define dev_udot4(a, b, c) __builtin_amdgcn_udot4((a), (b), (c), false)
#define dev_fmed3(a, b, c) __builtin_amdgcn_fmed3f((a), (b), (c))
#define dev_mad24(a, b, c) ((u32)__umul24((a), (b)) + (c))  // a, b < 2^24: v_mad_u32_u24
#define dev_readfirstlane(x) __builtin_amdgcn_readfirstlane(x)
#define dev_mfma_16x16x32_f16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define dev_wait_vmcnt0() __builtin_amdgcn_s_waitcnt(0x0f70)
#define dev_mov_dpp_quad_swap(x) __builtin_amdgcn_mov_dpp((x), 0xb1, 0xf, 0xf, true) /* quad_perm [1,0,3,2] */
#define dev_sched_barrier() __builtin_amdgcn_sched_barrier(0)
// lanes of a wave exchanging data through LDS: the hardware runs a wave's LDS operations in order, so nothing is needed (the
// builtin emits no instruction, it only keeps the compiler from moving LDS accesses across it); the emulator's lanes are threads
#define dev_wave_lds_sync() __builtin_amdgcn_wave_barrier()
// a dword of the dynamic LDS segment by its byte offset (the segment starts at LDS offset 0: the launchers check that the kernels
// declare no static LDS) — the decode table is addressed by offsets formed in the vector ALU, without a base add
__device__ __forceinline__ u32 dev_lds_u32(u32 byte_offset) {
    typedef __attribute__((address_space(3))) const u32 lds_u32;
    return *(lds_u32 *)(uintptr_t)byte_offset;
}
#endif

namespace {

struct MConv {  // a job's colour constants (scalar registers)
    float ky, krv, kgu, kgv, kbu, cr, cg, cb, ylo, yhi, clo, chi;
};

constexpr int M_LUT_ENTRIES = 768;  // decode LUT indexed by the unclamped code + 256: entries below 256 / above 511 repeat the ends

// Direct output: where the compositor would only copy a resampled tile's texels into the output frame (k_classify_tiles,
// cls[tile] == the job's layer), the kernel converts its finished pixels to Y'CbCr itself — the arithmetic of k_compose_output's
// copy tiles on the same bytes — and the RGBA8 texels are not stored.
struct MDirect {
    const u8 *cls;   // class per 128x16 output tile, nullptr = off
    int tiles_x, nv; // nv: 1 = NV12 (interleaved chroma in `up`)
    SurfView yp, up, vp;
};

// Colour constants of a frame format: Y'CbCr -> 255 * R'G'B' + 1280.5 with the range expansion and the clamps of
// planar_yuv_to_rgba.wgsl:45-57 folded in; luma in u8 units clamped to [ylo, yhi], chroma in 1/16 u8 units clamped to [clo, chi].
inline MConv m_conv_constants(bool full) {
    const double ys = full ? 1.0 : 255.0 / 219.0, y0 = full ? 0.0 : 16.0;          // 255 * ye = ys * (Y - y0)
    const double cs = full ? 1.0 / 16.0 : 255.0 / (16.0 * 224.0);                  // 255 * ue = cs * (U16 - 16 * c0)
    const double c0 = full ? 0.0 : 16.0 * 16.0, half = full ? 16.0 * 127.5 : 16.0 * 112.0;  // 255 * (ue - 0.5) = cs * (U16 - c0 - half)
    MConv K;
    K.ky = (float)ys;
    K.krv = (float)(1.5748 * cs); K.kgu = (float)(-0.1873 * cs); K.kgv = (float)(-0.4681 * cs); K.kbu = (float)(1.8556 * cs);
    const double bias = 1024.0 + 256.0 + 0.5;  // see m_convert_px: fixed exponent, LUT offset, round half up
    K.cr = (float)(bias - ys * y0 - 1.5748 * cs * (c0 + half));
    K.cg = (float)(bias - ys * y0 + (0.1873 + 0.4681) * cs * (c0 + half));
    K.cb = (float)(bias - ys * y0 - 1.8556 * cs * (c0 + half));
    K.ylo = full ? 0.0f : 16.0f; K.yhi = full ? 255.0f : 235.0f;
    K.clo = full ? 0.0f : 256.0f; K.chi = full ? 4080.0f : 3840.0f;
    return K;
}

#ifdef __HIPCC__

// One 4x1 pixel block: luma dword yy, chroma neighbourhoods (4 bytes: columns 2q-1 .. 2q+2) of chroma rows p (ua, va) and p + 1
// (ub, vb), w13 / w31 = the row's bilinear weight vectors -> (hi | lo << 16) linear texels: o[c] = the four texels of channel c,
// i.e. eight consecutive K values of a v_mfma_f32_16x16x32_f16 A operand.
//   NOLUT (profiling builds only): no table gathers.
template <bool NOLUT>
__device__ __forceinline__ void m_convert_px(const MConv &J, u32 yy, u32 ua, u32 ub, u32 va, u32 vb, u32 w13, u32 w31, uint4 o[3]) {
    const u32 pu0 = dev_perm(ub, ua, 0x05040100u), pu1 = dev_perm(ub, ua, 0x06050201u), pu2 = dev_perm(ub, ua, 0x07060302u);
    const u32 pv0 = dev_perm(vb, va, 0x05040100u), pv1 = dev_perm(vb, va, 0x06050201u), pv2 = dev_perm(vb, va, 0x07060302u);
    const u32 pus[4] = {pu0, pu1, pu1, pu2}, pvs[4] = {pv0, pv1, pv1, pv2};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 wgt = (i & 1) ? w31 : w13;
        const int u16 = (int)dev_udot4(pus[i], wgt, 0u), v16 = (int)dev_udot4(pvs[i], wgt, 0u);
        const float uf = dev_fmed3((float)u16, J.clo, J.chi), vf = dev_fmed3((float)v16, J.clo, J.chi);
        const float yf = dev_fmed3((float)((yy >> (8 * i)) & 0xffu), J.ylo, J.yhi);
        const float r = __builtin_fmaf(yf, J.ky, __builtin_fmaf(vf, J.krv, J.cr));
        const float g = __builtin_fmaf(yf, J.ky, __builtin_fmaf(uf, J.kgu, __builtin_fmaf(vf, J.kgv, J.cg)));
        const float b = __builtin_fmaf(yf, J.ky, __builtin_fmaf(uf, J.kbu, J.cb));
        // u8 quantisation of the node texture + decode in one lookup: the constants carry + 1280.5, so r = 1024 + 256 + floor(255 R' + 0.5)
        // + fraction with a fixed exponent — the code sits in mantissa bits [22:13], its LDS offset is (bits >> 11) & 0xffc, and the
        // clamp to [0, 255] is folded into the table (768 entries: the matrix cannot leave [-237, 492]).  Two full-rate integer
        // ops per channel instead of clamp + convert + shift.
        u32 tr, tg, tb;
        if (NOLUT) {
            tr = __float_as_uint(r) >> 13; tg = __float_as_uint(g) >> 13; tb = __float_as_uint(b) >> 13;
        } else {
            tr = dev_lds_u32((__float_as_uint(r) >> 11) & 0xffcu);
            tg = dev_lds_u32((__float_as_uint(g) >> 11) & 0xffcu);
            tb = dev_lds_u32((__float_as_uint(b) >> 11) & 0xffcu);
        }
        if (i == 0) { o[0].x = tr; o[1].x = tg; o[2].x = tb; }
        if (i == 1) { o[0].y = tr; o[1].y = tg; o[2].y = tb; }
        if (i == 2) { o[0].z = tr; o[1].z = tg; o[2].z = tb; }
        if (i == 3) { o[0].w = tr; o[1].w = tg; o[2].w = tb; }
    }
}

// The tile's four finished RGBA8 pixels of one output row -> their luma bytes and this lane's share of the 2x2 chroma blocks:
// rgba_to_yuv.wgsl:26-54 on the bytes, operation for operation as k_compose_output's copy tiles (smr_fused_compose.h
// store_yuv_block; smr_convert_dev.h unorm_of_byte / yuv_byte): unorm -> BT.709 -> unorm8.  Chroma = the mean of a 2x2 block,
// ((a + b) + (c + d)) / 4 — bit for bit (a/2 + b/2)/2 + (c/2 + d/2)/2, and either sum commutes: the two rows of a block sit in
// neighbouring lanes (output row even / odd; the tile's output position is even).  The even row's lane finishes the block of
// columns 0-1, the odd row's that of columns 2-3.  Returns the four luma bytes; *mine / *other = (U | V << 8) of the block this
// lane finished / the one its neighbour finished.
__device__ __forceinline__ u32 m_direct_yuv(const u32 px[4], bool odd, u32 *mine, u32 *other) {
    // fast path (smr_yuv_fast.h: yuv_luma_byte / yuv_chroma_bytes, the reference sequence behind their guard flags): luma from the bytes as floats;
    // chroma from the block's byte sums — this lane's row sums of both blocks, the neighbour's row sums of the block this lane finishes
    // (exact integers: either order of the sum is the same value)
    float f[4][3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        f[i][0] = (float)(px[i] & 0xffu); f[i][1] = (float)((px[i] >> 8) & 0xffu); f[i][2] = (float)((px[i] >> 16) & 0xffu);
    }
    u32 yq = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) yq |= yuv_luma_byte(px[i], f[i][0], f[i][1], f[i][2]) << (8 * i);
    // this lane finishes block (odd ? 1 : 0) of its four columns and hands the row sums and pixels of the other block to its neighbour
    const u32 own_a = odd ? px[2] : px[0], own_b = odd ? px[3] : px[1], snd_a = odd ? px[0] : px[2], snd_b = odd ? px[1] : px[3];
    float own[3], snd[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float h0 = f[0][c] + f[1][c], h1 = f[2][c] + f[3][c];
        own[c] = odd ? h1 : h0;
        snd[c] = odd ? h0 : h1;
    }
    const float nb_r = __int_as_float(dev_mov_dpp_quad_swap(__float_as_int(snd[0])));
    const float nb_g = __int_as_float(dev_mov_dpp_quad_swap(__float_as_int(snd[1])));
    const float nb_b = __int_as_float(dev_mov_dpp_quad_swap(__float_as_int(snd[2])));
    const u32 nb_a = (u32)dev_mov_dpp_quad_swap((int)snd_a), nb_bb = (u32)dev_mov_dpp_quad_swap((int)snd_b);  // (the neighbour row's pixels: the exact branch's operands)
    // upper row first: the even lane's own pixels are the block's upper row
    *mine = odd ? yuv_chroma_bytes(nb_a, nb_bb, own_a, own_b, own[0] + nb_r, own[1] + nb_g, own[2] + nb_b)
                : yuv_chroma_bytes(own_a, own_b, nb_a, nb_bb, own[0] + nb_r, own[1] + nb_g, own[2] + nb_b);
    *other = (u32)dev_mov_dpp_quad_swap((int)*mine);
    return yq;
}

// Stores what m_direct_yuv made: luma dword at (X, Y) and this lane's chroma (planar: the even row's lane writes U, the odd row's V;
// NV12: the even row's lane writes U0 V0 U1 V1).
__device__ __forceinline__ void m_direct_store(const MDirect *Dp, int X, int Y, bool odd, u32 yq, u32 mine, u32 other) {
    *(u32 *)(Dp->yp.ptr + (size_t)Y * Dp->yp.pitch + X) = yq;
    const int cx = X >> 1, cy = Y >> 1;
    if (Dp->nv) {
        if (!odd) *(u32 *)(Dp->up.ptr + (size_t)cy * Dp->up.pitch + (size_t)cx * 2) = (mine & 0xffffu) | (other << 16);
    } else if (!odd) {
        *(u16 *)(Dp->up.ptr + (size_t)cy * Dp->up.pitch + cx) = (u16)((mine & 0xffu) | ((other & 0xffu) << 8));
    } else {
        *(u16 *)(Dp->vp.ptr + (size_t)cy * Dp->vp.pitch + cx) = (u16)(((other >> 8) & 0xffu) | (mine & 0xff00u));
    }
}

#endif  // __HIPCC__

}  // namespace
