// TEST INFRASTRUCTURE — not part of the product; nothing in smelter_amd/ builds, links or loads this.
//
// Lane emulator for wave B of the hot path: k_classify_tiles + k_compose_output (smelter_amd/csrc/smr_fused_compose.h, with
// smr_layout_dev.h's fragment code) compiled for the CPU (SMR_EMU: <hip/hip_runtime.h> -> shim/, one host thread per lane, LDS =
// function-local statics shared by a workgroup's threads) and launched the way smr_render_layouts launches them: the host's own
// packing of the layout list (smr_pack_one_layout), the classification, then the compositor over the band list + every tile.
// tests/test_emu_compose.py holds the result — Y'CbCr planes, NV12 or an RGBA8 node — to the oracle's apply_layouts + output
// converters byte for byte, with every buffer guard-paged on request (emu_guard.h).
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

#include "smr_fused_compose.h"
#include "smr_tables.h"

namespace {

// A launch: `threads` host threads — one per lane — live for the whole grid and run its workgroups one after the other (a barrier between
// two workgroups: the LDS statics and the EmuBlock are the workgroup's).
template <typename F>
void run_grid(unsigned blocks, unsigned threads, F kernel) {
    gridDim = dim3(blocks);
    blockDim = dim3(threads);
    auto blk = std::make_unique<EmuBlock>();
    pthread_barrier_t step;
    pthread_barrier_init(&blk->bar, nullptr, threads);
    pthread_barrier_init(&step, nullptr, threads);
    emu_blk = blk.get();
    std::vector<std::thread> ts;
    ts.reserve(threads);
    for (unsigned t = 0; t < threads; t++)
        ts.emplace_back([&, t] {
            threadIdx = dim3(t);
            for (unsigned b = 0; b < blocks; b++) {
                blockIdx = dim3(b);
                kernel();
                pthread_barrier_wait(&step);
            }
        });
    for (auto &t : ts) t.join();
    pthread_barrier_destroy(&blk->bar);
    pthread_barrier_destroy(&step);
}

struct Surface {
    GuardBuf buf;
    SurfView view;
};
// rows on the allocator's 256-byte pitch, or (emu_set_guard(.., 1)) on the smallest pitch a surface of that width can have:
// `align`-byte multiples
void make_surface(Surface &s, const u8 *tight, int w, int h, int bpp, u32 align, u8 fill) {
    u32 pitch = (u32)(((size_t)w * bpp + 255) & ~(size_t)255);
    if (emu_min_pitch) pitch = (u32)(((size_t)w * bpp + align - 1) & ~(size_t)(align - 1));
    s.buf.alloc((size_t)pitch * h, fill, align);
    if (tight)
        for (int y = 0; y < h; y++) memcpy(s.buf.ptr + (size_t)y * pitch, tight + (size_t)y * w * bpp, (size_t)w * bpp);
    s.view.ptr = s.buf.ptr; s.view.pitch = pitch; s.view.w = w; s.view.h = h;
}
void read_back(const Surface &s, u8 *tight, int bpp) {
    for (int y = 0; y < s.view.h; y++) memcpy(tight + (size_t)y * s.view.w * bpp, s.buf.ptr + (size_t)y * s.view.pitch, (size_t)s.view.w * bpp);
}

}  // namespace

// layouts: n smr_layout records (include/smr.h; the oracle's pack_layouts writes the same POD).  Sources: n_sources premultiplied RGBA8
// textures, tight rows (src_px[i] == NULL: no texture), src_kind[i] 1 = RGBA8, 2 = RGBA8 known to be opaque (a resampled video tile).
// Output W x H (even): nv 0 -> out0 / out1 / out2 = Y, U, V planes (tight), nv 1 -> out0 = Y, out1 = interleaved UV, nv 2 -> out0 = RGBA8.
// banded: how many tiles the host's grid gives the band list — -1: the classifier's own count (a list that was read back), >= 0: that
// many (0: a prediction that missed everything: every composited tile is done by the workgroup that owns it).  slices: unused (the classifier decides a tile's bands).
// info[0..6] = tiles per class (TC_CLEAR .. TC_SELECT), info[7] = tiles on the compositing list, info[8] = workgroups of the compositor.
extern "C" int emu_compose(const smr_layout *layouts, int n, int n_sources, const u8 *const *src_px, const int *src_w, const int *src_h, const int *src_kind,
                           int W, int H, int nv, int srgb, int banded, int slices, int allow_select, u8 *out0, u8 *out1, u8 *out2, int *info) {
    if (W < 2 || H < 2 || (W & 1) || (H & 1) || n < 0 || n > MAX_LAYOUT_WORDS * 32 || (slices != 2 && slices != 4 && slices != 8) || nv < 0 || nv > 2) return -1;
    static float tables_src[SMR_TABLE_FLOATS];
    static u32 lut16[SMR_LUT16_WORDS];
    static bool have_tables = false;
    if (!have_tables) {
        if (!smr_build_tables(tables_src, lut16)) return -9;
        have_tables = true;
    }
    GuardBuf tables;
    tables.alloc(sizeof(tables_src), 0, 16);
    memcpy(tables.ptr, tables_src, sizeof(tables_src));

    std::vector<Surface> srcs((size_t)n_sources);
    std::vector<SurfView> views((size_t)n_sources);
    std::vector<int> kinds((size_t)n_sources, 0);
    for (int i = 0; i < n_sources; i++) {
        if (!src_px[i]) continue;
        make_surface(srcs[i], src_px[i], src_w[i], src_h[i], 4, 16, 0x77);
        views[i] = srcs[i].view;
        kinds[i] = src_kind[i];
    }
    // the host's packing (smr_pack_layouts without its staging ring)
    size_t total_masks = 0;
    for (int i = 0; i < n; i++) total_masks += layouts[i].masks_len > SMR_MAX_MASKS ? SMR_MAX_MASKS : layouts[i].masks_len;
    GuardBuf dl, dm;
    dl.alloc((size_t)(n ? n : 1) * sizeof(DevLayout), 0, 16);
    dm.alloc((total_masks ? total_masks : 1) * sizeof(DevMask), 0, 16);
    DevLayout *hl = (DevLayout *)dl.ptr;
    DevMask *hm = (DevMask *)dm.ptr;
    u32 mo = 0;
    for (int i = 0; i < n; i++) smr_pack_one_layout(layouts[i], views.data(), kinds.data(), (u32)n_sources, W, H, srgb != 0, tables_src + 256, hl[i], hm, mo);

    const int tiles_x = (W + B_TILE_W - 1) / B_TILE_W, tiles_y = (H + B_TILE_H - 1) / B_TILE_H, tiles = tiles_x * tiles_y;
    GuardBuf tcb, directb, listb;
    tcb.alloc((size_t)tiles * sizeof(TileClass), 0, 16);
    directb.alloc(((size_t)tiles + 15) & ~(size_t)15, 0xff, 16);
    listb.alloc(sizeof(TileList) + (size_t)tiles * B_AREA_BANDS * sizeof(TileFull), 0, 16);
    TileClass *tc = (TileClass *)tcb.ptr;
    TileList *full = (TileList *)listb.ptr;
    run_grid((unsigned)((tiles + B_CLASSIFY_TILES - 1) / B_CLASSIFY_TILES), 64 * B_CLASSIFY_TILES,
             [&] { k_classify_tiles(hl, hm, n, W, H, tiles_x, tiles, 0ull, tc, directb.ptr, full, allow_select, 0); });
    if (info) {
        for (int k = 0; k < 9; k++) info[k] = 0;
        for (int t = 0; t < tiles; t++)
            if (tc[t].kind <= TC_SELECT) info[tc[t].kind]++;
        info[7] = (int)full->count[0];
#ifdef SMR_EMU_HEAVY_DEBUG
        { int hv = 0; for (u32 k = 0; k < full->count[0]; k++) hv += (full->e[k].general & 4u) ? 1 : 0; fprintf(stderr, "listed %u heavy %d\n", full->count[0], hv); }
#endif
    }
    u32 n_banded = banded < 0 ? full->count[0] : (u32)banded;
    if (n_banded > (u32)tiles * B_AREA_BANDS) n_banded = (u32)tiles * B_AREA_BANDS;

    Surface p0, p1, p2;
    if (nv == 2) {
        make_surface(p0, nullptr, W, H, 4, 16, 0xe1);
        p1.view = p2.view = p0.view;
    } else {
        make_surface(p0, nullptr, W, H, 1, 4, 0xe1);
        if (nv == 1) {
            make_surface(p1, nullptr, W / 2, H / 2, 2, 4, 0xe1);
            p2.view = p1.view;
        } else {
            make_surface(p1, nullptr, W / 2, H / 2, 1, 2, 0xe1);
            make_surface(p2, nullptr, W / 2, H / 2, 1, 2, 0xe1);
        }
    }
    const bool big = n > B_MAX_LAYOUTS || (int)mo > B_MAX_MASKS;
    const unsigned grid = n_banded + (unsigned)((tiles + B_COPY_TILES - 1) / B_COPY_TILES);
    if (info) info[8] = (int)grid;
    const float *tab = (const float *)tables.ptr;
    const int flags = srgb ? 1 : 0;
#define EMU_COMPOSE(NVv, BIGv) \
    run_grid(grid, 256, [&] { k_compose_output<NVv, BIGv>(p0.view, p1.view, p2.view, W, H, hl, hm, n, (int)mo, flags, tab, tiles_x, tiles, tc, full, (int)n_banded, 0); })
    if (nv == 0) { if (big) EMU_COMPOSE(0, true); else EMU_COMPOSE(0, false); }
    else if (nv == 1) { if (big) EMU_COMPOSE(1, true); else EMU_COMPOSE(1, false); }
    else { if (big) EMU_COMPOSE(2, true); else EMU_COMPOSE(2, false); }
#undef EMU_COMPOSE
    if (nv == 2) read_back(p0, out0, 4);
    else {
        read_back(p0, out0, 1);
        if (nv == 1) read_back(p1, out1, 2);
        else { read_back(p1, out1, 1); read_back(p2, out2, 1); }
    }
    return 0;
}
