// TEST INFRASTRUCTURE — not part of the product; nothing in smelter_amd/ builds, links or loads this.
//
// Lane emulator for k_ingest_wave / k_build_wave_weights: compiles the kernel source of smelter_amd/csrc/smr_ingest_wave.h for the
// CPU (SMR_EMU: builtins -> emu_device.h, <hip/hip_runtime.h> -> shim/) and runs every workgroup with one host thread per lane.
// It exists to check the kernels' index logic (window geometry, K permutation of pass 2, ring slots, staging, piece splits) on
// machines without a GPU; arithmetic follows the instruction semantics, except that the MFMA sums its 32 products in order.
#include <pthread.h>

#include <memory>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

thread_local dim3 threadIdx, blockIdx;
dim3 gridDim, blockDim;

#include "emu_device.h"
#include "emu_guard.h"
EmuBlock *emu_blk = nullptr;
thread_local unsigned char *emu_smem = nullptr;
void __syncthreads() { pthread_barrier_wait(&emu_blk->bar); }

#include "smr_ingest_wave.h"
#include "smr_tables.h"

namespace {

// A launch: `threads` host threads — one per lane — live for the whole grid and run its workgroups one after the other (a barrier between
// two workgroups: the LDS block and the EmuBlock are the workgroup's).
template <typename F>
void run_grid(unsigned blocks, unsigned threads, size_t lds, F kernel) {
    gridDim = dim3(blocks);
    blockDim = dim3(threads);
#ifdef SMR_EMU_ASAN
    std::vector<unsigned char> smem(lds ? lds : 1);  // (the dynamic LDS block at its exact size: an overrun is a report)
#else
    std::vector<unsigned char> smem(lds + 64);
#endif
    auto blk = std::make_unique<EmuBlock>();
    pthread_barrier_t step;
    pthread_barrier_init(&blk->bar, nullptr, threads);
    pthread_barrier_init(&step, nullptr, threads);
    for (unsigned w = 0; w < (threads + 63) / 64; w++) pthread_barrier_init(&blk->waves[w].bar, nullptr, 64);
    blk->smem = smem.data();
    emu_blk = blk.get();
    std::vector<std::thread> ts;
    ts.reserve(threads);
    for (unsigned t = 0; t < threads; t++)
        ts.emplace_back([&, t] {
            threadIdx = dim3(t);
            emu_smem = blk->smem;
            for (unsigned b = 0; b < blocks; b++) {
                blockIdx = dim3(b);
                kernel();
                pthread_barrier_wait(&step);
            }
        });
    for (auto &t : ts) t.join();
    pthread_barrier_destroy(&blk->bar);
    pthread_barrier_destroy(&step);
    for (unsigned w = 0; w < (threads + 63) / 64; w++) pthread_barrier_destroy(&blk->waves[w].bar);
}

struct Plane {
    std::shared_ptr<GuardBuf> buf;  // (shared: the node-texture builds alias the three plane views)
    SurfView view;
};
// rows on the allocator's 256-byte pitch; with emu_set_guard(.., 1) a NODE texture (min_row != 0) sits on the smallest pitch
// can_fuse_wave_rgba lets through: 16-byte multiples that hold the row rounded up to four texels
Plane make_plane(const u8 *tight, int w, int h, int bpp, u32 min_row = 0) {
    Plane p;
    u32 pitch = (u32)(((size_t)w * bpp + 255) & ~(size_t)255);
    if (emu_min_pitch && min_row) pitch = (min_row + 15u) & ~15u;
    p.buf = std::make_shared<GuardBuf>();
    p.buf->alloc((size_t)pitch * h, bpp == 8 ? 0xff : 0xcd, 16);  // (row padding: arbitrary bytes; as f16 texels 0xffff is a NaN)
    for (int y = 0; y < h; y++) memcpy(p.buf->ptr + (size_t)y * pitch, tight + (size_t)y * w * bpp, (size_t)w * bpp);
    p.view.ptr = p.buf->ptr; p.view.pitch = pitch; p.view.w = w; p.view.h = h;
    return p;
}

struct Band {
    GuardBuf mem;
    void *meta;
    uint4 *frag;
    int K, nks, n_units;
    bool k01 = false;
};
Band build_band(float scale, float offset, int n_dst, int n_src, int axis) {
    Band B;
    wave_band_geometry(scale, offset, n_dst, n_src, axis, &B.K, &B.nks, &B.k01);
    const int n_tiles = (n_dst + 15) / 16;
    B.n_units = axis == 2 ? (n_tiles + 1) / 2 : n_tiles;
    const size_t meta_bytes = ((size_t)B.n_units * (axis != 3 ? sizeof(int4) : sizeof(int2)) + 15) & ~(size_t)15;
    const size_t frags = axis != 3 ? (size_t)B.n_units * 2 * B.K * 2 : (size_t)B.n_units * B.K * 2;
    B.mem.alloc(meta_bytes + frags * 64 * sizeof(uint4), 0, 16);
    u8 *base = B.mem.ptr;
    B.meta = base;
    B.frag = (uint4 *)(base + meta_bytes);
    WWBatch args;
    memset(&args, 0, sizeof(args));
    WWBuild &b = args.b[0];
    b.scale = scale; b.offset = offset; b.taps = w_taps(scale); b.n_dst = n_dst; b.n_src = n_src; b.axis = axis; b.K = B.K; b.unit0 = 0;
    b.meta = B.meta; b.frag = B.frag;
    args.n = 1;
    run_grid((unsigned)B.n_units, 64, 0, [&] { k_build_wave_weights(args); });
    return B;
}

}  // namespace

// The matrix-core kernel's one-gather sRGB encode (w_encode8, smr_ingest_wave.h) against the step function it stands for, u8 = #{i : thr[i] <= x}
// (srgb_encode8's definition, smr_internal.h), on EVERY f32 from 0 up to 1.5 (the clamp's far side included), their negatives and the specials:
// returns how many disagree (0), -9 when the tables cannot be built.
extern "C" long long emu_check_encode(void) {
    static float tables[SMR_TABLE_FLOATS];
    static u32 lut16[SMR_LUT16_WORDS];
    if (!smr_build_tables(tables, lut16)) return -9;
    const float *thr = tables + 256;
    long long bad = 0;
    u32 code = 0;  // #{i >= 1 : thr[i] <= x}, kept up to date while x walks upwards
    u32 top;
    const float top_f = 1.5f;
    memcpy(&top, &top_f, 4);
    for (u32 bits = 0; bits <= top; bits++) {
        float x;
        memcpy(&x, &bits, 4);
        while (code < 255 && thr[code + 1] <= x) code++;
        if (w_encode8(x, lut16 + 256) != code) bad++;
        if (w_encode8(-x, lut16 + 256) != 0u) bad++;
    }
    const float inf = __builtin_inff();
    if (w_encode8(inf, lut16 + 256) != 255u || w_encode8(-inf, lut16 + 256) != 0u || w_encode8(3.0e38f, lut16 + 256) != 255u) bad++;
    return bad;
}

// One job through k_ingest_wave.  Planes tightly packed (NV12: u = interleaved UV, v ignored); dst = dw x dh RGBA8, tight.
// (scale, offset) per axis as smr_resample_plan_make gives them for a two-pass, horizontal-first plan.  `pieces`: vertical pieces
// per column pair (rounded up to a multiple of the workgroup's waves); `specialised` != 0 asks for the <4, 3, 2> build when the
// job's k-step counts allow it.  Returns 0, or a negative number when the job is outside the kernel's limits:
// -1 k-steps, -2 specialised build not applicable.  info[0..3] = NKS, NKS, KV, workgroups.
extern "C" int emu_ingest_wave(const u8 *y, const u8 *u, const u8 *v, int sw, int sh, int full_range, int nv12, float scale_h, float off_h, float scale_v,
                               float off_v, u8 *dst, int dw, int dh, int pieces, int specialised, int *info) {
    static float tables[SMR_TABLE_FLOATS];
    static u32 lut16[SMR_LUT16_WORDS];
    static bool have_tables = false;
    if (!have_tables) {
        if (!smr_build_tables(tables, lut16)) return -9;
        have_tables = true;
    }
    const bool planes = nv12 == 7 || nv12 == 8;  // the plane-source builds (262144): planar (7) / NV12 (8) frame, converted exactly in the wave; class builds only
    if (planes) nv12 = nv12 == 8 ? 1 : 0;
    const bool f16 = nv12 == 3 || nv12 == 5;   // y = an RGBA16F node texture (linear light): the 8192 + 16384 build (5: with an alpha channel)
    const bool alpha = nv12 == 4 || nv12 == 5; // y = a premultiplied node texture with an alpha channel: the + 65536 builds
    const bool rgb12 = nv12 == 6;              // y = the node texture as RGB12 (12 bytes per four pixels; sw a multiple of 4): the 8192 + 131072 builds
    const bool rgba = nv12 == 2 || f16 || alpha || rgb12;   // y = the RGBA8 node texture (alpha 255), u / v ignored: the kernel's 8192 builds
    if (rgba) nv12 = 0;
    const u32 sw4 = ((u32)sw + 3u) & ~3u;  // (a node's rows hold whole groups of four texels)
    Plane py = rgb12 ? make_plane(y, 3 * sw, sh, 1, 3 * sw4) : make_plane(y, sw, sh, f16 ? 8 : rgba ? 4 : 1, rgba ? sw4 * (f16 ? 8u : 4u) : 0u);
    if (rgb12) py.view.w = sw;
    Plane pu = rgba ? py : (nv12 ? make_plane(u, sw / 2, sh / 2, 2) : make_plane(u, sw / 2, sh / 2, 1));
    Plane pv = (nv12 || rgba) ? pu : make_plane(v, sw / 2, sh / 2, 1);
    const u32 tile_pitch = emu_min_pitch ? (u32)(((size_t)dw * 4 + 15) & ~(size_t)15) : (u32)(((size_t)dw * 4 + 255) & ~(size_t)255);
    GuardBuf tile;
    tile.alloc((size_t)tile_pitch * dh, 0x5a, 16);
    const bool single = specialised == 3;  // one tile per unit (axis 4 bands): windows too wide for a pair; generic builds
    if (single) specialised = 0;
    Band bh = build_band(scale_h, off_h, dw, sw, single ? 4 : 2), bv = build_band(scale_v, off_v, dh, sh, planes ? 5 : 3);
    if (info) { info[0] = bh.nks; info[1] = bh.K; info[2] = bv.K; }
    if (bh.K > W_NKS_MAX || bv.K > W_KV_MAX) return -1;
    const bool cls432 = bh.K <= 4 && bv.K == 2, cls83 = bh.K <= 8 && bv.K == 3, cls82 = !cls432 && bh.K <= 8 && bv.K == 2;
    const bool sa = specialised == 2;  // single-axis plan: scale_v = 1, off_v = the perpendicular crop offset; the 32768 builds
    if (sa) specialised = 0;
    if (specialised && !cls432 && !cls83 && !cls82) return -2;
    if (planes && !(specialised && (cls432 || cls83 || cls82))) return -2;

    WArgs args;
    memset(&args, 0, sizeof(args));
    WJob &J = args.jobs[0];
    J.yp = py.view; J.up = pu.view; J.vp = nv12 ? pu.view : pv.view;
    if (nv12) { J.up.w = sw / 2; J.vp = J.up; }
    J.dst.ptr = tile.ptr;
    J.dst.pitch = tile_pitch; J.dst.w = dw; J.dst.h = dh;
    J.src_w = sw; J.src_h = sh;
    J.conv = m_conv_constants(full_range != 0);
    J.h_meta = (const int4 *)bh.meta; J.h_frag = bh.frag; J.NKS = bh.K; J.n_pairs = bh.n_units;
    J.n_htiles = (dw + 15) / 16;
    J.v_meta = (const int2 *)bv.meta; J.v_frag = bv.frag; J.KV = bv.K; J.n_vtiles = bv.n_units;
    int p = pieces < 1 ? 1 : pieces;
    p = (p + W_WAVES - 1) / W_WAVES * W_WAVES;
    J.pieces = p;
    J.nv12 = nv12;
    J.layer = -1;
    J.single = single ? 1 : 0;
    J.full = full_range != 0;
    J.k01 = bh.k01 ? 1 : 0;
    args.wg_prefix[0] = 0;
    args.wg_prefix[1] = J.n_pairs * (p / W_WAVES);
    args.n_jobs = 1;
    if (sa) J.perp = (int)off_v;
    const bool spec = specialised && cls432, spec83 = specialised && !cls432 && cls83, spec82 = specialised && !cls432 && !cls83 && cls82;
    const int cls_nks = spec ? 4 : 0;
    args.b_bytes = w_band_bytes(cls_nks ? cls_nks : bh.K);
    args.raw_bytes = planes ? w_fx_node_bytes(cls_nks ? cls_nks : bh.K) : (w_raw_bytes(cls_nks ? cls_nks : bh.K) + 15) & ~15;
    args.direct = nullptr;
    const size_t lds = (size_t)W_OFF_B + args.b_bytes + (planes ? W_FX_TABLE_BYTES : 0) + (size_t)W_WAVES * args.raw_bytes;
    const int total = args.wg_prefix[1];
    if (info) info[3] = total;
    const unsigned blocks = (unsigned)((total + 7) & ~7);
    if (planes) {
        if (nv12) {
            if (spec && bh.k01) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 262145 + 4096>(args, tables, lut16); });
            else if (spec) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 262144 + 4096>(args, tables, lut16); });
            else if (spec83) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 3, 262144 + 4096>(args, tables, lut16); });
            else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 2, 262144 + 4096>(args, tables, lut16); });
        } else {
            if (spec && bh.k01) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 262145>(args, tables, lut16); });
            else if (spec) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 262144>(args, tables, lut16); });
            else if (spec83) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 3, 262144>(args, tables, lut16); });
            else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 2, 262144>(args, tables, lut16); });
        }
    } else if (sa) {
        if (alpha) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 32768 + 8192 + 65536>(args, tables, lut16); });
        else if (rgba) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 32768 + 8192>(args, tables, lut16); });
        else if (nv12) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 32768 + 4096>(args, tables, lut16); });
        else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 32768>(args, tables, lut16); });
    } else if (f16) {
        if (alpha) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 8192 + 16384 + 65536>(args, tables, lut16); });
        else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 8192 + 16384>(args, tables, lut16); });
    } else if (alpha) {
        if (spec && bh.k01) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 8193 + 65536>(args, tables, lut16); });
        else if (spec) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 8192 + 65536>(args, tables, lut16); });
        else if (spec83) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 3, 8192 + 65536>(args, tables, lut16); });
        else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 8192 + 65536>(args, tables, lut16); });
    } else if (rgb12) {
        if (spec && bh.k01) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 8193 + 131072>(args, tables, lut16); });
        else if (spec) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 8192 + 131072>(args, tables, lut16); });
        else if (spec83) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 3, 8192 + 131072>(args, tables, lut16); });
        else if (spec82) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 2, 8192 + 131072>(args, tables, lut16); });
        else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 8192 + 131072>(args, tables, lut16); });
    } else if (rgba) {
        if (spec && bh.k01) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 8193>(args, tables, lut16); });
        else if (spec) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 8192>(args, tables, lut16); });
        else if (spec83) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 3, 8192>(args, tables, lut16); });
        else if (spec82) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 2, 8192>(args, tables, lut16); });
        else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 8192>(args, tables, lut16); });
#ifdef SMR_EMU_ASAN  // (the instrumented library holds the node-texture builds — what a product build launches — and the single-axis ones: half
                     //  the compile time; the plane-source builds of laboratory libraries stay with the plain emulator)
    } else {
        return -3;
    }
#else
    } else if (spec && bh.k01) {
        if (nv12) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 4097>(args, tables, lut16); });
        else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 1>(args, tables, lut16); });
    } else if (specialised && cls82 && !rgba) {
        if (nv12) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 2, 4096>(args, tables, lut16); });
        else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 2, 0>(args, tables, lut16); });
    } else if (spec83) {
        if (nv12) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 3, 4096>(args, tables, lut16); });
        else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<8, 3, 0>(args, tables, lut16); });
    } else if (spec) {
        if (nv12) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 4096>(args, tables, lut16); });
        else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<4, 2, 0>(args, tables, lut16); });
    } else {
        if (nv12) run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 4096>(args, tables, lut16); });
        else run_grid(blocks, W_THREADS, lds, [&] { k_ingest_wave<0, 0, 0>(args, tables, lut16); });
    }
#endif
    for (int yy = 0; yy < dh; yy++) memcpy(dst + (size_t)yy * dw * 4, J.dst.ptr + (size_t)yy * J.dst.pitch, (size_t)dw * 4);
    return 0;
}
