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
    LayoutSlot &slot = ctx->layout_ring[ctx->layout_ring_next];
    ctx->layout_ring_next = (ctx->layout_ring_next + 1) % ctx->layout_ring.size();
    if (slot.busy) {
        SMR_HIP(ctx, hipEventSynchronize(slot.done));
        slot.busy = false;
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
    const float DEG = 0.017453292519943295f;
    u32 mo = 0;
    for (u32 i = 0; i < n; i++) {
        const smr_layout &L = layouts[i];
        DevLayout &D = hl[i];
        memset(&D, 0, sizeof(D));
        D.top = L.top; D.left = L.left; D.width = L.width; D.height = L.height;
        for (int k = 0; k < 4; k++) {
            D.radius[k] = L.border_radius[k];
            D.color[k] = L.color[k];
            D.border_color[k] = L.border_color[k];
            D.crop[k] = L.crop[k];
        }
        D.type = L.type;
        D.border_width = L.border_width;
        D.blur = L.blur_radius;
        D.masks_off = mo;
        D.masks_len = L.masks_len > SMR_MAX_MASKS ? SMR_MAX_MASKS : L.masks_len;
        for (u32 m = 0; m < D.masks_len; m++) {
            const smr_mask &K = L.masks[m];
            DevMask &DM = hm[mo++];
            memset(&DM, 0, sizeof(DM));
            for (int k = 0; k < 4; k++) DM.radius[k] = K.radius[k];
            DM.top = K.top; DM.left = K.left; DM.width = K.width; DM.height = K.height;
            // solid region of smoothstep(-.5, .5, -sdf): the rect inset by .5 minus the four corner squares of side max radius
            // (+ rounding slack unless every quantity is a multiple of 1/2 below 2^15: then each f32 operation of the SDF is exact)
            const float mr = fmaxf(fmaxf(K.radius[0], K.radius[1]), fmaxf(K.radius[2], K.radius[3]));
            auto hi = [](float v) { return v * 2.0f == floorf(v * 2.0f) && fabsf(v) < 32768.0f; };
            const bool mexact = hi(K.radius[0]) && hi(K.radius[1]) && hi(K.radius[2]) && hi(K.radius[3]) && hi(K.left) && hi(K.top) &&
                                hi(K.width) && hi(K.height);
            const float mslack = mexact ? 0.0f : 0.015625f;
            DM.inset = 0.5f + mslack;
            DM.corner = (mr + mslack > DM.inset) ? mr + mslack : 0.0f;
        }
        float qleft = L.left, qtop = L.top, qw = L.width, qh = L.height;
        if (L.type == 2) {  // box shadow quad grown by blur on each side (apply_layouts.wgsl:216-229)
            qleft = L.left - L.blur_radius; qtop = L.top - L.blur_radius;
            qw = L.width + 2.0f * L.blur_radius; qh = L.height + 2.0f * L.blur_radius;
        }
        D.qw = qw; D.qh = qh;
        D.cx = qleft + qw / 2.0f; D.cy = qtop + qh / 2.0f;
        float ang = L.rotation_degrees * DEG;
        D.cs = cosf(ang); D.sn = sinf(ang);
        D.src_kind = 0;
        D.tex_w = 1; D.tex_h = 1;
        D.src_index = -1;
        if (L.type == 0 && L.source_index < n_sources && src_kind[L.source_index] != 0) {
            D.src = src_views[L.source_index];
            D.src_kind = src_kind[L.source_index];
            D.tex_w = D.src.w; D.tex_h = D.src.h;
            D.src_index = (int)L.source_index;
        }
        D.rqw = 1.0f / D.qw; D.rqh = 1.0f / D.qh;
        D.rtw = 1.0f / (float)D.tex_w; D.rth = 1.0f / (float)D.tex_h;
        // ---- classification helpers for the fused compose kernel
        D.flags = (D.cs == 1.0f && D.sn == 0.0f) ? DL_UNROTATED : 0;
        // Solid region: inside the rect inset by m >= radius on every side the SDF is <= -m, i.e. edge_distance >= m.
        // m must also reach the point where every smoothstep saturates at exactly 1:
        //   no border: smoothstep(-.5,.5,ed) -> ed >= .5;  texture border: smoothstep(bw-.5,bw+.5,ed) -> ed >= bw+.5;
        //   colour border: smoothstep(bw,bw+1,ed) -> ed >= bw+1;  shadow: smoothstep(-b/2,b/2,ed) -> ed >= b/2.
        float rmax = fmaxf(fmaxf(L.border_radius[0], L.border_radius[1]), fmaxf(L.border_radius[2], L.border_radius[3]));
        float need = 0.5f;
        if (L.type == 2) need = fmaxf(L.blur_radius / 2.0f, 0.5f);  // >= .5 keeps the region inside the half-open quad coverage
        else if (L.border_width >= 1.0f) need = L.border_width + (L.type == 0 ? 0.5f : 1.0f);
        // Outside the four corner squares (side = max radius) the SDF is the plain distance to the nearest straight edge
        // (smr_layout_dev.h, rect_solid_box), so only those squares and the `need` band along the edges are not solid.
        auto half_int = [](float v) { return v * 2.0f == floorf(v * 2.0f) && fabsf(v) < 32768.0f; };
        const bool exact = half_int(L.border_radius[0]) && half_int(L.border_radius[1]) && half_int(L.border_radius[2]) &&
                           half_int(L.border_radius[3]) && half_int(L.left) && half_int(L.top) && half_int(L.width) && half_int(L.height) &&
                           half_int(need) && half_int(qleft) && half_int(qw) && half_int(qtop) && half_int(qh);
        const float slack = exact ? 0.0f : 0.015625f;  // 1/64 px for f32 rounding in the SDF
        D.inset = need + slack;
        D.corner = (rmax + slack > D.inset) ? rmax + slack : 0.0f;
        if (L.type == 0 && D.src_kind != 0 && (D.flags & DL_UNROTATED) && L.crop[0] == 0.0f && L.crop[1] == 0.0f &&
            L.crop[2] == (float)D.tex_w && L.crop[3] == (float)D.tex_h && L.width == (float)D.tex_w && L.height == (float)D.tex_h &&
            L.left == floorf(L.left) && L.top == floorf(L.top) && fabsf(L.left) < 65536.0f && fabsf(L.top) < 65536.0f) {
            D.flags |= DL_ALIGNED;
            D.ix = (int)L.left;
            D.iy = (int)L.top;
        }
        if (L.type != 0 && L.color[3] == 1.0f) {
            D.flags |= DL_COLOR_OPAQUE;
            const float *thr = ctx->h_tables + 256;
            auto enc = [&](float x) -> u32 {
                if (ctx->srgb()) {
                    if (!(x > 0.0f)) return 0u;
                    return (u32)(std::upper_bound(thr + 1, thr + 256, x) - (thr + 1));  // #{i in 1..255 : thr[i] <= x}
                }
                x = !(x > 0.0f) ? 0.0f : (x > 1.0f ? 1.0f : x);
                return (u32)(int)(x * 255.0f + 0.5f);
            };
            D.solid_px = enc(L.color[0]) | (enc(L.color[1]) << 8) | (enc(L.color[2]) << 16) | (255u << 24);
        }
        const bool finite = std::isfinite(qleft) && std::isfinite(qtop) && std::isfinite(qw) && std::isfinite(qh) && std::isfinite(D.cs) &&
                            std::isfinite(D.sn);
        if (!(qw > 0.0f) || !(qh > 0.0f) || L.type > 2 || !finite) {
            // (a quad with a NaN / infinite corner rasterises to nothing; it must not reach the float -> int conversions below)
            D.bx0 = D.by0 = 0; D.bx1 = D.by1 = -1;  // never binned
        } else if (D.flags & DL_UNROTATED) {
            // pixel x is covered iff qleft <= x + .5 < qleft + qw (layout_covers), i.e. qleft - .5 <= x < qleft + qw - .5.
            // Tight bounds matter: a one-pixel margin makes every neighbour of a tile-aligned rect "touch" the next tile column.
            // 1/64 px of slack unless the quad is on half-integers (then the coverage arithmetic is exact in f32).
            const float s = (half_int(qleft) && half_int(qtop) && half_int(qw) && half_int(qh)) ? 0.0f : 0.015625f;
            auto clampi_h = [](float v, int lo, int hi) { return v < (float)lo ? lo : (v > (float)hi ? hi : (int)v); };
            D.bx0 = clampi_h(ceilf(qleft - 0.5f - s), 0, out_w);
            D.bx1 = clampi_h(ceilf(qleft + qw - 0.5f + s), 0, out_w);
            D.by0 = clampi_h(ceilf(qtop - 0.5f - s), 0, out_h);
            D.by1 = clampi_h(ceilf(qtop + qh - 0.5f + s), 0, out_h);
        } else {
            float ex = fabsf(D.cs) * qw / 2.0f + fabsf(D.sn) * qh / 2.0f;
            float ey = fabsf(D.sn) * qw / 2.0f + fabsf(D.cs) * qh / 2.0f;
            auto clampi_h = [](float v, int lo, int hi) { return v < (float)lo ? lo : (v > (float)hi ? hi : (int)v); };
            D.bx0 = clampi_h(floorf(D.cx - ex - 1.0f), 0, out_w);
            D.bx1 = clampi_h(ceilf(D.cx + ex + 1.0f), 0, out_w);
            D.by0 = clampi_h(floorf(D.cy - ey - 1.0f), 0, out_h);
            D.by1 = clampi_h(ceilf(D.cy + ey + 1.0f), 0, out_h);
        }
    }
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
// A pack that equals, byte for byte, the one committed last (a scene that does not move between two frames: same layouts, same
// source surfaces, same parameter block) reuses that pack's device copy: no copy is queued, the frame is its kernels only.  The
// staging slot just filled goes back to the ring (it is the next one handed out), the reused slot stays busy until this frame is done.
int smr_pack_commit(smr_ctx *ctx, PackedLayouts *p) {
    LayoutSlot *prev = ctx->layout_last >= 0 ? &ctx->layout_ring[(size_t)ctx->layout_last] : nullptr;
    if (prev && prev != p->slot && !ctx->no_pack_reuse && prev->resident_bytes == p->copy_bytes && memcmp(prev->host, p->slot->host, p->copy_bytes) == 0) {
        const ptrdiff_t d = (const u8 *)prev->dev - (const u8 *)p->slot->dev;
        p->layouts = (const DevLayout *)((const u8 *)p->layouts + d);
        p->masks = (const DevMask *)((const u8 *)p->masks + d);
        p->extra_dev = (u8 *)p->extra_dev + d;
        ctx->layout_ring_next = (size_t)(p->slot - ctx->layout_ring.data());
        p->slot = prev;
        ctx->pack_reused++;
        return SMR_OK;
    }
    SMR_HIP(ctx, hipMemcpyAsync(p->slot->dev, p->slot->host, p->copy_bytes, hipMemcpyHostToDevice, ctx->stream));
    p->slot->resident_bytes = p->copy_bytes;
    ctx->layout_last = (int)(p->slot - ctx->layout_ring.data());
    return SMR_OK;
}

int smr_pack_done(smr_ctx *ctx, PackedLayouts *p) {
    SMR_HIP(ctx, hipEventRecord(p->slot->done, ctx->stream));
    p->slot->busy = true;
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
