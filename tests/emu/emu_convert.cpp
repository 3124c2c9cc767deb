// TEST INFRASTRUCTURE — not part of the product; nothing in smelter_amd/ builds, links or loads this.
//
// The 4:2:0 input converter's source (smelter_amd/csrc/smr_convert_420.h: cv420_block, one 4 x 4 pixel block per call; cv420_run, a vertical run of blocks per call) compiled for the
// CPU, so that tests/test_emu_convert.py can hold it to the oracle's planar_yuv_to_rgba / nv12_to_rgba bit for bit without a GPU.
// Same shims as emu_wave.cpp (SMR_EMU: builtins -> emu_device.h, <hip/hip_runtime.h> -> shim/); no threads are needed here: a block
// is computed by one lane and lanes do not talk to each other.
#include <vector>

#include <hip/hip_runtime.h>

thread_local dim3 threadIdx, blockIdx;
dim3 gridDim, blockDim;

#include "emu_device.h"
#include "emu_guard.h"
EmuBlock *emu_blk = nullptr;
thread_local unsigned char *emu_smem = nullptr;
void __syncthreads() {}

#include "smr_convert_420.h"

// emu_set_tight(1): the TIGHT builds (k_yuv420_to_rgba_tight: nothing behind a window's last column is requested) — and, with
// emu_set_guard(.., 1), chroma planes on the pitch such a frame has: the row's bytes rounded up to a dword, not the plain kernel's reach
static int emu_tight = 0;
extern "C" void emu_set_tight(int on) { emu_tight = on; }

namespace {

struct Plane {
    GuardBuf buf;
    SurfView view;
};
// rows on a 256-byte pitch like smr_surface_create's, padding filled with a byte no plane contains by accident; with emu_set_guard(.., 1)
// on the smallest pitch conv_420_ok (smr_convert.hip) lets through: `min_row` bytes (what the last block's window loads reach), dword-aligned
Plane make_plane(const u8 *tight, int w, int h, int bpp, u8 fill, u32 min_row = 0) {
    Plane p;
    u32 pitch = (u32)(((size_t)w * bpp + 255) & ~(size_t)255);
    if (emu_min_pitch) {
        pitch = (u32)(((size_t)w * bpp + 3) & ~(size_t)3);
        if (pitch < min_row && !emu_tight) pitch = (min_row + 3u) & ~3u;
    }
    // (a plane the library allocates ends with SMR_SURFACE_TAIL spare bytes and may be read past a row's end — conv_420_ok's rule for owned
    //  surfaces; a wrapped one, emu_min_pitch, is exactly pitch * h bytes and is only let through with a pitch that holds the reach)
    p.buf.alloc((size_t)pitch * h + (emu_min_pitch || emu_tight ? 0 : SMR_SURFACE_TAIL), fill, 4);
    for (int y = 0; y < h; y++) memcpy(p.buf.ptr + (size_t)y * pitch, tight + (size_t)y * w * bpp, (size_t)w * bpp);
    p.view.ptr = p.buf.ptr; p.view.pitch = pitch; p.view.w = w; p.view.h = h;
    return p;
}
// conv_420_ok's `need`: the last block's chroma window starts at column cw - 3; its bytes begin in the dword at ((cw - 3) [* 2]) & ~3 and
// the loads reach 8 (12) bytes from there
u32 chroma_need(int w, int nv12) {
    const u32 cw = (u32)w / 2;
    return nv12 ? ((2u * (cw - 3u)) & ~3u) + 12u : ((cw - 3u) & ~3u) + 8u;
}
u32 node_pitch(int w, int rgb12) {
    if (emu_min_pitch) return rgb12 ? (u32)((3 * w + 15) & ~15) : (u32)w * 4;
    return rgb12 ? (u32)((3 * w + 255) & ~255) : (u32)w * 4;
}

}  // namespace

// y: w x h; planar: u, v: (w / 2) x (h / 2); NV12: u = interleaved (w / 2) x (h / 2) x 2, v ignored.  out: w x h x 4 tight.
// rgb12 != 0: out = h rows of 3 w bytes (12-byte groups of four pixels: R x 4, G x 4, B x 4)
// nb = 0: one cv420_block call per block; nb >= 1: cv420_run over runs of nb block rows (what k_yuv420_to_rgba's waves execute)
extern "C" int emu_convert_420_run(const u8 *y, const u8 *u, const u8 *v, int w, int h, int nv12, int full, int rgb12, int nb, u8 *out) {
    if (w % 4 || w < 8 || h % 2 || h < 2) return -1;
    const u32 need = chroma_need(w, nv12);
    Plane py = make_plane(y, w, h, 1, 0x5a), pu = make_plane(u, w / 2, h / 2, nv12 ? 2 : 1, 0xa5, need), pv = nv12 ? Plane() : make_plane(v, w / 2, h / 2, 1, 0x3c, need);
    const u32 dpitch = node_pitch(w, rgb12);
    GuardBuf dst;
    dst.alloc((size_t)dpitch * h, 0, 16);
    ConvJob J;
    memset(&J, 0, sizeof(J));
    J.yp = py.view; J.up = pu.view; J.vp = nv12 ? pu.view : pv.view;
    J.dst.ptr = dst.ptr; J.dst.pitch = dpitch; J.dst.w = w; J.dst.h = h;
    J.full = full; J.nv = nv12; J.sx = 1; J.sy = 1; J.packed = 0; J.rgb12 = rgb12;
    float ylut[256], nlut[256];
    for (u32 b = 0; b < 256; b++) {
        ylut[b] = cv420_luma_of_byte(b, full != 0);
        nlut[b] = unorm_of_byte(b);
    }
    for (int P = 0; 4 * P < h; P += nb ? nb : 1)
        for (int g = 0; 4 * g < w; g++) {
#define EMU_CV(NVv, R12, FULLv) \
    do { \
        if (emu_tight) { if (nb) cv420_run<NVv, R12, FULLv, true>(J, g, P, nb, ylut, nlut); else cv420_block<NVv, R12, FULLv, true>(J, g, P, ylut, nlut); } \
        else { if (nb) cv420_run<NVv, R12, FULLv>(J, g, P, nb, ylut, nlut); else cv420_block<NVv, R12, FULLv>(J, g, P, ylut, nlut); } \
    } while (0)
            if (full) {
                if (nv12 && rgb12) EMU_CV(true, true, true);
                else if (nv12) EMU_CV(true, false, true);
                else if (rgb12) EMU_CV(false, true, true);
                else EMU_CV(false, false, true);
            } else {
                if (nv12 && rgb12) EMU_CV(true, true, false);
                else if (nv12) EMU_CV(true, false, false);
                else if (rgb12) EMU_CV(false, true, false);
                else EMU_CV(false, false, false);
            }
#undef EMU_CV
        }
    if (rgb12) {
        for (int r = 0; r < h; r++) memcpy(out + (size_t)r * 3 * w, dst.ptr + (size_t)r * dpitch, (size_t)3 * w);
    } else {
        memcpy(out, dst.ptr, (size_t)w * 4 * h);
    }
    return 0;
}

extern "C" int emu_convert_420(const u8 *y, const u8 *u, const u8 *v, int w, int h, int nv12, int full, int rgb12, u8 *out) {
    return emu_convert_420_run(y, u, v, w, h, nv12, full, rgb12, 0, out);
}

// What a launch of k_yuv420_to_rgba computes: `n` frames of one kind (all planar or all NV12; widths / heights / ranges / node formats per
// frame) partitioned the way the host does (cv420_plan: bands dealt to the XCDs, equal shares per wave) over `waves` WORKGROUPS, every lane of every wave emulated in turn.
// ys / us / vs / outs: n pointers; ws / hs / fulls / rgb12s: n ints.  Output layout per frame as emu_convert_420_run.
extern "C" int emu_convert_420_shares(int n, const u8 *const *ys, const u8 *const *us, const u8 *const *vs, const int *ws, const int *hs, int nv12,
                                      const int *fulls, const int *rgb12s, int waves, u8 *const *outs) {
    if (n < 1 || n > MAX_CONV_JOBS || waves < 1) return -1;
    std::vector<Plane> py(n), pu(n), pv(n);
    std::vector<GuardBuf> dst(n);
    std::vector<u32> dpitch(n);
    ConvBatch B;
    memset(&B, 0, sizeof(B));
    B.n = n;
    for (int i = 0; i < n; i++) {
        const int w = ws[i], h = hs[i];
        if (w % 4 || w < 8 || h % 2 || h < 2) return -1;
        const u32 need = chroma_need(w, nv12);
        py[i] = make_plane(ys[i], w, h, 1, 0x5a); pu[i] = make_plane(us[i], w / 2, h / 2, nv12 ? 2 : 1, 0xa5, need);
        if (!nv12) pv[i] = make_plane(vs[i], w / 2, h / 2, 1, 0x3c, need);
        dpitch[i] = node_pitch(w, rgb12s[i]);
        dst[i].alloc((size_t)dpitch[i] * h, 0, 16);
        ConvJob &J = B.j[i];
        J.yp = py[i].view; J.up = pu[i].view; J.vp = nv12 ? pu[i].view : pv[i].view;
        J.dst.ptr = dst[i].ptr; J.dst.pitch = dpitch[i]; J.dst.w = w; J.dst.h = h;
        J.full = fulls[i]; J.nv = nv12; J.sx = 1; J.sy = 1; J.packed = 0; J.rgb12 = rgb12s[i];
    }
    // `waves` = workgroups here (four waves each), as the host launches them: the partition is the host's own (cv420_plan)
    u32 grid = (u32)waves;
    const u32 bands = cv420_plan(B, n, grid * 4u);
    if (grid < (bands < 8u ? bands : 8u)) grid = bands < 8u ? bands : 8u;  // (as smr_frames_to_rgba_batch does: every XCD that owns a band runs a workgroup)
    float ylut[256], nlut[256];
    for (u32 b = 0; b < 256; b++) {
        ylut[b] = cv420_luma_of_byte(b, false);
        nlut[b] = unorm_of_byte(b);
    }
    for (u32 blk = 0; blk < grid; blk++)
        for (u32 wv = 0; wv < 4; wv++)
            for (u32 lane = 0; lane < 64; lane++) {
                if (emu_tight) {
                    if (nv12) cv420_share<true, true>(B, blk, wv, grid, lane, ylut, nlut);
                    else cv420_share<false, true>(B, blk, wv, grid, lane, ylut, nlut);
                } else if (nv12) cv420_share<true>(B, blk, wv, grid, lane, ylut, nlut);
                else cv420_share<false>(B, blk, wv, grid, lane, ylut, nlut);
            }
    for (int i = 0; i < n; i++) {
        const int w = ws[i], h = hs[i];
        if (rgb12s[i]) {
            for (int r = 0; r < h; r++) memcpy(outs[i] + (size_t)r * 3 * w, dst[i].ptr + (size_t)r * dpitch[i], (size_t)3 * w);
        } else {
            memcpy(outs[i], dst[i].ptr, (size_t)w * 4 * h);
        }
    }
    return 0;
}
