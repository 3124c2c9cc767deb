// smr_layout.hip — the layout compositor.
//
// Replaces LayoutShader::render (smelter-render/src/transformations/layout/shader.rs:93-167),
// ParamsBindGroups::update (layout/params.rs:169-334) and apply_layouts.wgsl:127-377.
//
// The reference clears the target and issues one full-pipeline draw per layout, each a
// read-modify-write of the RGBA8 target.  Here ALL layouts are applied in ONE launch:
// every 32x8-pixel workgroup bins the layout list against its tile (bit mask in LDS),
// then each thread walks the surviving layouts back to front with the running colour
// held in a register.  The colour is re-quantised to RGBA8 (sRGB-encoded in
// GpuOptimized mode) after every layout, exactly where the reference's render-target
// store would quantise it, so results are identical to N separate draws while the
// target is written once and never read.
#include "smr_layout_dev.h"

#include <algorithm>

namespace {

__global__ __launch_bounds__(LAYOUT_TILE_W *LAYOUT_TILE_H) void k_apply_layouts(SurfView target, const DevLayout *__restrict__ layouts,
                                                                               const DevMask *__restrict__ masks, int n,
                                                                               int srgb, const float *__restrict__ tables) {
    __shared__ u32 s_bits[MAX_LAYOUT_WORDS];
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * LAYOUT_TILE_W, ty0 = blockIdx.y * LAYOUT_TILE_H;
    bin_layouts(s_bits, layouts, n, tx0, ty0, tx0 + LAYOUT_TILE_W, ty0 + LAYOUT_TILE_H, tid, LAYOUT_TILE_W * LAYOUT_TILE_H);

    const int px = tx0 + (tid % LAYOUT_TILE_W), py = ty0 + (tid / LAYOUT_TILE_W);
    if (px >= target.w || py >= target.h) return;

    const float *dec = tables, *thr = tables + 256;
    u32 acc = 0;  // RGBA8 of the cleared target (wgpu::Color::TRANSPARENT, shader.rs:135)
    const int words = (n + 31) >> 5;
    for (int wi = 0; wi < words; wi++) {
        u32 bits = s_bits[wi];
        while (bits) {
            const int li = (wi << 5) + __builtin_ctz(bits);
            bits &= bits - 1;
            acc = composite_layout(acc, layouts[li], masks, px, py, srgb, dec, thr);
        }
    }
    *(u32 *)(target.ptr + (size_t)py * target.pitch + (size_t)px * 4) = acc;
}

}  // namespace

// Packs the POD layout list into the compact device form (rotation, quad and bounding box
// precomputed on the host in f32, as the vertex stage does per draw: apply_layouts.wgsl:127-157).
int smr_pack_layouts(smr_ctx *ctx, const smr_layout *layouts, u32 n, const SurfView *src_views, const int *src_kind,
                     u32 n_sources, int out_w, int out_h, size_t extra_bytes, PackedLayouts *out) {
    if (n > ctx->max_layouts) n = ctx->max_layouts;  // params.rs:176-182: extra layouts are skipped
    if (n > MAX_LAYOUT_WORDS * 32) n = MAX_LAYOUT_WORDS * 32;
    size_t total_masks = 0;
    for (u32 i = 0; i < n; i++) total_masks += layouts[i].masks_len > SMR_MAX_MASKS ? SMR_MAX_MASKS : layouts[i].masks_len;
    const size_t lay_bytes = ((size_t)n * sizeof(DevLayout) + 255) & ~(size_t)255;
    const size_t mask_bytes = (total_masks * sizeof(DevMask) + 255) & ~(size_t)255;
    const size_t bytes = lay_bytes + mask_bytes + extra_bytes + 256;

    // ring of pinned staging slots; a slot is reusable once the kernel that read its device copy is done
    if (ctx->layout_ring.empty()) ctx->layout_ring.resize(8);
    // (a slot whose device copy recent frames keep reusing — a node of the scene that does not move while another does — is passed over:
    //  recycling it would wait for the stream)
    ctx->pack_clock++;
    for (size_t tries = 0; tries + 1 < ctx->layout_ring.size(); tries++) {
        const LayoutSlot &c = ctx->layout_ring[ctx->layout_ring_next];
        if (!(c.busy && c.unfenced && ctx->pack_clock - c.last_reuse < 64)) break;
        ctx->layout_ring_next = (ctx->layout_ring_next + 1) % ctx->layout_ring.size();
    }
    {   // every slot is some node's resident pack (a scene of many nested layout nodes at rest under one that moves): the ring grows rather
        // than recycle one of them every frame (slots are addressed by index between calls: ctx->layout_last)
        const LayoutSlot &c = ctx->layout_ring[ctx->layout_ring_next];
        if (c.busy && c.unfenced && ctx->pack_clock - c.last_reuse < 64 && ctx->layout_ring.size() < 64) {
            ctx->layout_ring.emplace_back();
            ctx->layout_ring_next = ctx->layout_ring.size() - 1;
        }
    }
    LayoutSlot &slot = ctx->layout_ring[ctx->layout_ring_next];
    ctx->layout_ring_next = (ctx->layout_ring_next + 1) % ctx->layout_ring.size();
    if (slot.busy) {
        // (a slot whose copy later frames reused without an event of their own: those frames are on this stream)
        if (slot.unfenced) SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        else SMR_HIP(ctx, hipEventSynchronize(slot.done));
        slot.busy = false;
        slot.unfenced = false;
    }
    if (slot.bytes < bytes) {
        if (slot.host) (void)hipHostFree(slot.host);
        if (slot.dev) (void)hipFree(slot.dev);
        slot.host = nullptr;
        slot.dev = nullptr;
        slot.bytes = 0;
        size_t want = (bytes + 65535) & ~(size_t)65535;
        SMR_HIP(ctx, hipHostMalloc(&slot.host, want, hipHostMallocDefault));
        SMR_HIP(ctx, hipMalloc(&slot.dev, want));
        slot.bytes = want;
    }
    if (!slot.done) SMR_HIP(ctx, hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    // (the slot's device copy no longer matches anything; zeroed so that equal packs are equal byte for byte, padding included)
    slot.resident_bytes = 0;
    if (ctx->layout_last == (int)(&slot - ctx->layout_ring.data())) ctx->layout_last = -1;
    memset(slot.host, 0, bytes);

    DevLayout *hl = (DevLayout *)slot.host;
    DevMask *hm = (DevMask *)((u8 *)slot.host + lay_bytes);
    u32 mo = 0;
    for (u32 i = 0; i < n; i++) smr_pack_one_layout(layouts[i], src_views, src_kind, n_sources, out_w, out_h, ctx->srgb(), ctx->h_tables + 256, hl[i], hm, mo);
    out->layouts = (const DevLayout *)slot.dev;
    out->masks = (const DevMask *)((u8 *)slot.dev + lay_bytes);
    out->host_layouts = hl;
    out->host_masks = hm;
    out->n = (int)n;
    out->n_masks = (int)mo;
    out->slot = &slot;
    out->extra_host = (u8 *)slot.host + lay_bytes + mask_bytes;
    out->extra_dev = (u8 *)slot.dev + lay_bytes + mask_bytes;
    out->copy_bytes = lay_bytes + mask_bytes + extra_bytes;
    return SMR_OK;
}

// host -> device copy of the packed slot (after the caller filled the extra region)
// A pack that equals, byte for byte, one whose device copy is still resident (a scene — or one node of it: every LayoutNode of a frame
// packs its own list — that does not move between two frames: same layouts, same source surfaces, same parameter block) reuses that
// device copy: no copy is queued, the frame is its kernels only.  The staging slot just filled goes back to the ring (it is the next one
// handed out), the reused slot stays busy until this frame is done.
int smr_pack_commit(smr_ctx *ctx, PackedLayouts *p) {
    LayoutSlot *prev = nullptr;
    if (!ctx->no_pack_reuse) {
        const size_t nslots = ctx->layout_ring.size();
        for (size_t k = 0; k < nslots && !prev; k++) {  // (the pack committed last first: a scene of one node at rest)
            LayoutSlot *c = &ctx->layout_ring[((size_t)(ctx->layout_last < 0 ? 0 : ctx->layout_last) + nslots - k) % nslots];
            if (c != p->slot && c->resident_bytes == p->copy_bytes && memcmp(c->host, p->slot->host, p->copy_bytes) == 0) prev = c;
        }
    }
    if (prev) {
        const ptrdiff_t d = (const u8 *)prev->dev - (const u8 *)p->slot->dev;
        p->layouts = (const DevLayout *)((const u8 *)p->layouts + d);
        p->masks = (const DevMask *)((const u8 *)p->masks + d);
        p->extra_dev = (u8 *)p->extra_dev + d;
        ctx->layout_ring_next = (size_t)(p->slot - ctx->layout_ring.data());
        p->slot = prev;
        p->reused = true;
        prev->last_reuse = ctx->pack_clock;
        ctx->pack_reused++;
        return SMR_OK;
    }
    SMR_HIP(ctx, hipMemcpyAsync(p->slot->dev, p->slot->host, p->copy_bytes, hipMemcpyHostToDevice, ctx->stream));
    p->slot->resident_bytes = p->copy_bytes;
    ctx->layout_last = (int)(p->slot - ctx->layout_ring.data());
    return SMR_OK;
}

// After the last kernel that reads the pack was queued.  A frame that queued a copy records the slot's event; a frame that reused the previous
// frame's device copy — a scene at rest: every frame of a static scene — records nothing: an event is a marker packet on the stream, and the
// trace shows ~5 us between a frame's last kernel and the next frame's first one on the same stream because of it (profiles/r06_sensitivity.txt).
// The slot is then "unfenced": the one time it is recycled (the scene changed and the ring came round) the host waits for the stream instead.
int smr_pack_done(smr_ctx *ctx, PackedLayouts *p) {
    p->slot->busy = true;
    if (p->reused) {
        p->slot->unfenced = true;
        return SMR_OK;
    }
    SMR_HIP(ctx, hipEventRecord(p->slot->done, ctx->stream));
    p->slot->unfenced = false;
    return SMR_OK;
}

extern "C" int smr_apply_layouts(smr_ctx *ctx, smr_surface *target, const smr_layout *layouts, uint32_t n,
                                 const smr_surface *const *sources, uint32_t n_sources) {
    SMR_ENTER(ctx);
    if (!ctx || !target || (n && !layouts)) return SMR_ERR_INVALID;
    if (target->fmt != SMR_PX_RGBA8) return smr_fail(ctx, SMR_ERR_INVALID, "smr_apply_layouts: target must be RGBA8");
    std::vector<SurfView> views(n_sources ? n_sources : 1);
    std::vector<int> kinds(n_sources ? n_sources : 1, 0);
    for (u32 i = 0; i < n_sources; i++) {
        if (sources && sources[i]) {
            if (sources[i]->fmt != SMR_PX_RGBA8) return smr_fail(ctx, SMR_ERR_INVALID, "smr_apply_layouts: source %u is not RGBA8", i);
            views[i] = view_of(sources[i]);
            kinds[i] = 1;
        }
    }
    PackedLayouts packed;
    int rc = smr_pack_layouts(ctx, layouts, n, views.data(), kinds.data(), n_sources, (int)target->w, (int)target->h, 0, &packed);
    if (rc != SMR_OK) return rc;
    rc = smr_pack_commit(ctx, &packed);
    if (rc != SMR_OK) return rc;
    rc = smr_launch_apply_layouts(ctx, target, &packed);
    if (rc != SMR_OK) return rc;
    return smr_pack_done(ctx, &packed);
}

int smr_launch_apply_layouts(smr_ctx *ctx, smr_surface *target, const PackedLayouts *p) {
    StageScope scope(ctx, SMR_STAGE_LAYOUT);
    ctx->kernel_launches[SMR_KERNEL_APPLY_LAYOUTS]++;
    dim3 grid((target->w + LAYOUT_TILE_W - 1) / LAYOUT_TILE_W, (target->h + LAYOUT_TILE_H - 1) / LAYOUT_TILE_H, 1);
    hipLaunchKernelGGL(k_apply_layouts, grid, dim3(LAYOUT_TILE_W * LAYOUT_TILE_H), 0, ctx->stream, view_of(target), p->layouts,
                       p->masks, p->n, ctx->srgb() ? 1 : 0, ctx->d_tables);
    SMR_HIP(ctx, hipGetLastError());
    return SMR_OK;
}
