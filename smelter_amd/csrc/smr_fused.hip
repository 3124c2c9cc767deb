// smr_fused.hip — the per-frame hot path: LayoutNode::render + read_outputs in two kernel waves.
//
// Reference sequence per frame (smelter-render/src/state/render_loop.rs:19-230,
// transformations/layout.rs:169-278): per input a YUV->RGBA pass, per scaled child two Lanczos
// passes through an Rgba16Float intermediate, N blended layout draws, three RGBA->YUV passes —
// every one a full round trip through memory (~470 MB of traffic for 8x1080p -> 4K, against
// 37 MB of algorithmic bytes).
//
// Here:
//   wave A  k_yuv420_to_rgba (smr_convert_420.h: every frame of the call in one launch, the reference's node texture bit for bit,
//           RGB12 where only the resampler reads it) + k_ingest_wave (smr_ingest_wave.h: both Lanczos passes on the matrix cores,
//           node textures -> dst-sized RGBA8 tiles, all inputs of the frame in one launch; the f16 intermediate lives in registers).
//           Options: the conversion folded into k_ingest_wave (SMR_INGEST_LAB_FUSED: no node texture at all, not exact);
//           k_ingest_resample (smr_fused_ingest.h: every pass in f32, SMR_INGEST_VALU_F32).
//   wave B  k_compose_output (smr_fused_compose.h) — all layouts + RGBA->Y'CbCr (or an RGBA8 node target) in one launch, driven
//           by per-tile class records (k_classify_tiles) that are kept while the layout list repeats; the RGBA8 output frame
//           lives only in registers.
// Every quantisation point of the reference pipeline (u8 node texture, f16 intermediate, u8 sRGB tile, u8 render target after
// each draw) is reproduced; the compositor and the f32 ingest kernel keep the f32 operation sequence of the general kernels
// (the only substitutions are exact ones: LUTs for u8 -> f32, correctly rounded division through a reciprocal + FMAs, operations
// that cannot act on the operands at hand, layers an opaque layer overwrites) — tests/test_gpu_fused.py checks fused ==
// pass-per-launch bit for bit; the matrix-core resampler stays within 1 LSB of the f32 passes on the same node texture.
// Anything the fused kernels do not cover (single-pass plans, box pre-reduction, packed inputs, odd
// output sizes, 4:2:2 / 4:4:4 outputs) falls back to the general kernels of smr_convert / smr_resample / smr_layout per
// layout — never to the CPU.
#include "smr_fused_compose.h"
#include "smr_fused_ingest.h"
#include "smr_ingest_wave.h"

#include <cstdlib>

namespace {

bool fused_disabled(smr_ctx *ctx) {  // SMR_OPT_FUSED_KERNELS
#ifdef SMR_LAB
    if (!ctx->ablate_read) {
        const char *a = getenv("SMR_ABLATE");
        ctx->ablate = a ? atoi(a) : 0;
        ctx->ablate_read = true;
    }
#endif
    return ctx->fused_disabled == 1;
}

// every YUV-family FrameData variant converts to alpha == 1 (wgpu/format/*_to_rgba.wgsl return vec4(.., 1.0))
bool frame_is_opaque(u32 fmt) { return fmt <= SMR_FRAME_NV12; }

// surface-cache slots: smr_internal.h (one table, disjoint ranges)
constexpr size_t SLOT_TARGET = SMR_SLOT_TARGET, SLOT_INGEST_NODE = SMR_SLOT_INGEST_NODE, SLOT_NODE0 = SMR_SLOT_NODE0, SLOT_TILE0 = SMR_SLOT_TILE0,
                 SLOT_TRANSPOSED0 = SMR_SLOT_TRANSPOSED0, SLOT_TRANSPOSED_SINGLE = SMR_SLOT_TRANSPOSED_SINGLE, SLOT_REDUCED0 = SMR_SLOT_REDUCED0;

// An RGB12 node texture (12 bytes per four pixels: smr_convert_420.h) serves a layout when the frame is one k_yuv420_to_rgba takes and the plan
// is a plain two-pass, horizontal-first one within the kernel's pair windows — every input of the benchmark scenes.
bool rgb12_node_serves(smr_ctx *ctx, const smr_frame *f, const smr_resample_plan &plan, const smr_surface *tile) {
    if (!ctx->compact_nodes || !smr_conv_rgb12_ok(ctx, f)) return false;
    if (!(plan.kind == 2 && plan.levels[0] == 0 && plan.levels[1] == 0 && plan.axis[0] == 0 && plan.axis[1] == 1)) return false;
    SurfView probe;  // (the geometry test needs the node's size, not its pixels)
    probe.ptr = nullptr; probe.pitch = (3u * f->width + 255u) & ~255u; probe.w = (int)f->width; probe.h = (int)f->height;
    bool single = false;
    if (!can_fuse_wave_rgba(ctx, probe, plan, tile, 3, &single) || single) return false;
    // ... and only where it was measured to pay: the two class builds (scales around 1.5 and around 3 — configs[2] 59.6 -> 58.1 us of converter +
    // resampler per frame, configs[3] 189 -> 167 us); the generic build is slower on RGB12 (configs[1] 29.0 -> 31.7 us), profiles/r04_rgb12.txt
    int NKS, KV, unused;
    if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[0], plan.offset[0], (int)tile->w, (int)f->width, 2)) NKS = t->K;
    else wave_band_geometry(plan.scale[0], plan.offset[0], (int)tile->w, (int)f->width, 2, &NKS, &unused);
    if (const smr_ctx::MfmaTable *t = find_mfma_table(ctx, plan.scale[1], plan.offset[1], (int)tile->h, (int)f->height, 3)) KV = t->K;
    else wave_band_geometry(plan.scale[1], plan.offset[1], (int)tile->h, (int)f->height, 3, &KV, &unused);
    return (NKS <= 4 && KV == 2) || (NKS <= 8 && KV == 3) || (ctx->rgb12_cls82 && NKS <= 8 && KV == 2);
}
SurfView rgb12_view(const smr_surface *node, const smr_frame *f) {  // the node in pixels; its rows hold 3 w bytes
    SurfView v = view_of(node);
    v.w = (int)f->width;
    return v;
}

}  // namespace

extern "C" int smr_render_layouts(smr_ctx *ctx, const smr_layout *layouts, uint32_t n, const smr_source *sources,
                                  uint32_t n_sources, uint32_t out_w, uint32_t out_h, const smr_frame *out,
                                  smr_surface *out_rgba) {
    SMR_ENTER(ctx);
    if (!ctx || (n && !layouts) || (n_sources && !sources)) return SMR_ERR_INVALID;
    if (!out && !out_rgba) return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: no output given");
    if (out_w == 0 || out_h == 0) return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: empty output");
    if (out && (out->width != out_w || out->height != out_h))
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: output frame is %ux%u, expected %ux%u", out->width, out->height, out_w, out_h);
    if (out_rgba && (out_rgba->fmt != SMR_PX_RGBA8 || out_rgba->w != out_w || out_rgba->h != out_h))
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: out_rgba must be RGBA8 %ux%u", out_w, out_h);
    if (out)
        if (int rc = smr_validate_frame(ctx, out, "smr_render_layouts (output)")) return rc;
    if (n > ctx->max_layouts) n = ctx->max_layouts;
    if (n > SMR_SLOT_MAX_LAYOUTS) n = (uint32_t)SMR_SLOT_MAX_LAYOUTS;  // (the packer draws no more than MAX_LAYOUT_WORDS * 32 = 1024 anyway; every per-layout scratch slot stays in its range)
    if (n_sources > SMR_SLOT_MAX_SOURCES) return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: too many sources");
    const bool fused = !fused_disabled(ctx);

    // ---- sources: node views; frames get a node surface only if some layout needs one
    std::vector<SurfView> views(n_sources + n + 1);
    std::vector<int> kinds(n_sources + n + 1, 0);  // 0 none, 1 RGBA8, 2 RGBA8 known opaque
    std::vector<int> src_w(n_sources + 1, 0), src_h(n_sources + 1, 0);
    std::vector<u8> node_ready(n_sources + 1, 0);
    for (u32 i = 0; i < n_sources; i++) {
        const smr_source &s = sources[i];
        if ((s.kind == SMR_SOURCE_SURFACE || s.kind == SMR_SOURCE_OPAQUE_SURFACE) && s.surface) {
            if (s.surface->fmt != SMR_PX_RGBA8) return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: source %u is not RGBA8", i);
            views[i] = view_of(s.surface);
            kinds[i] = s.kind == SMR_SOURCE_OPAQUE_SURFACE ? 2 : 1;
            src_w[i] = (int)s.surface->w;
            src_h[i] = (int)s.surface->h;
            node_ready[i] = 1;
        } else if (s.kind == SMR_SOURCE_FRAME && s.frame && s.frame->planes[0]) {
            if (int rc = smr_validate_frame(ctx, s.frame, "smr_render_layouts (source frame)")) return rc;
            kinds[i] = frame_is_opaque(s.frame->format) ? 2 : 1;  // view filled lazily by ensure_node
            src_w[i] = (int)s.frame->width;
            src_h[i] = (int)s.frame->height;
        } else if (s.kind != SMR_SOURCE_NONE && s.kind > SMR_SOURCE_OPAQUE_SURFACE) {
            return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: bad source kind %u", s.kind);
        }
    }
    // (the conversions are queued and go out together — one launch for the 4:2:0 frames of a call, smr_frames_to_rgba_batch — before
    //  the first kernel that reads a node texture: flush_nodes)
    std::vector<const smr_frame *> conv_in;
    std::vector<smr_surface *> conv_node;
    std::vector<u8> conv_rgb12;
    std::vector<SurfView> cviews(n_sources + 1);   // RGB12 node textures, for the matrix-core resampler only
    std::vector<u8> cnode_ready(n_sources + 1, 0);
    auto ensure_rgb12_node = [&](u32 i) -> int {
        if (cnode_ready[i]) return SMR_OK;
        const smr_frame *f = sources[i].frame;
        smr_surface *node = smr_cached_surface(ctx, SMR_SLOT_NODE_RGB12_0 + i, 3 * f->width, f->height, SMR_PX_R8);
        if (!node) return SMR_ERR_OOM;
        conv_in.push_back(f);
        conv_node.push_back(node);
        conv_rgb12.push_back(1);
        cviews[i] = rgb12_view(node, f);
        cnode_ready[i] = 1;
        return SMR_OK;
    };
    auto ensure_node = [&](u32 i) -> int {
        if (node_ready[i]) return SMR_OK;
        const smr_frame *f = sources[i].frame;
        smr_surface *node = smr_cached_surface(ctx, SLOT_NODE0 + i, f->width, f->height, SMR_PX_RGBA8);
        if (!node) return SMR_ERR_OOM;
        conv_in.push_back(f);
        conv_node.push_back(node);
        conv_rgb12.push_back(0);
        views[i] = view_of(node);
        node_ready[i] = 1;
        return SMR_OK;
    };
    auto flush_nodes = [&]() -> int {
        if (conv_in.empty()) return SMR_OK;
        int rc = smr_frames_to_rgba_batch(ctx, conv_in.data(), conv_node.data(), (u32)conv_in.size(), conv_rgb12.data());
        conv_in.clear();
        conv_node.clear();
        conv_rgb12.clear();
        return rc;
    };

    // ---- resample_scaled_children (layout.rs:238-278): per texture layout decide direct / general / fused
    std::vector<smr_layout> eff(layouts, layouts + n);
    std::vector<IngestJob> jobs;
    std::vector<WJob> wjobs, wjobs_rgb12, wjobs_rgba, wjobs_rgba_alpha, wjobs_f16, wjobs_f16_alpha, wjobs_sa, wjobs_sa_rgba, wjobs_sa_rgba_alpha;
    std::vector<WJob> wjobs_planes[2][3];  // plane-source jobs by (NV12, class): the frame's planes, converted in the kernel (the default route of 4:2:0 frames)
    std::vector<u32> wjob_layout, wjob_rgb12_layout, wjob_rgba_layout;  // layout index of each job (direct output)
    std::vector<MTransposeBack> transposed;
    ctx->weight_call++;
    u32 next_view = n_sources;
    for (u32 li = 0; li < n; li++) {
        smr_layout &L = eff[li];
        if (L.type != 0) continue;
        const u32 si = L.source_index;
        if (si >= n_sources || kinds[si] == 0) { L.source_index = SMR_NO_SOURCE; continue; }
        const bool is_frame = sources[si].kind == SMR_SOURCE_FRAME;
        bool resampled = false;
        if (ctx->srgb()) {  // CpuOptimized has no resampler (layout/layout_renderer.rs:22-27)
            const float rw = roundf(L.width), rh = roundf(L.height);
            const u32 dw = rw >= 1.0f ? (u32)rw : 1u, dh = rh >= 1.0f ? (u32)rh : 1u;
            smr_resample_plan plan;
            int kind = smr_resample_plan_make((u32)src_w[si], (u32)src_h[si], L.crop, dw, dh, &plan);
            if (kind < 0) return smr_fail(ctx, kind, "smr_render_layouts: degenerate resample plan for layout %u", li);
            if (kind > 0) {
                if (dw > 16384 || dh > 16384) return smr_fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: layout %u is too large", li);
                smr_surface *tile = smr_cached_surface(ctx, SLOT_TILE0 + li, dw, dh, SMR_PX_RGBA8);
                if (!tile) return SMR_ERR_OOM;
                bool on_mfma = false;
                if (fused && is_frame && can_fuse_wave(ctx, sources[si].frame, plan, tile)) {  // the wave-autonomous matrix-core kernel
                    WJob J;
                    int rc = make_wave_job(ctx, sources[si].frame, plan, tile, &J);
                    if (rc != SMR_OK) return rc;
                    wjobs.push_back(J); wjob_layout.push_back(li);
                    on_mfma = true;
                }
                if (!on_mfma && fused && is_frame) {  // a vertical-first plan: the same kernel on the transposed frame
                    WJob J;
                    MTransposeBack back;
                    int rc = make_wave_job_transposed(ctx, sources[si].frame, plan, tile, SLOT_TRANSPOSED0 + 4 * (size_t)li, &J, &on_mfma, &back);
                    if (rc != SMR_OK) return rc;
                    if (on_mfma) { wjobs.push_back(J); wjob_layout.push_back(li); transposed.push_back(back); }
                }
                // a single-axis plan (only one of width / height changes) of an opaque source: the one pass on the matrix cores, its f32
                // sums encoded directly (k_ingest_wave's 32768 builds); a height-only plan runs on the transposed frame / node
                if (!on_mfma && fused && kinds[si] != 0 && plan.kind == 1 && plan.levels[0] == 0 && plan.levels[1] == 0) {
                    const smr_resample_plan p2 = single_axis_as_two_pass(plan);
                    const int perp = plan.perp_offset[0];
                    WJob J;
                    MTransposeBack back;
                    bool rgba_job = false, took = false;
                    if (is_frame && can_fuse_wave(ctx, sources[si].frame, p2, tile)) {
                        int rc = make_wave_job(ctx, sources[si].frame, p2, tile, &J);
                        if (rc != SMR_OK) return rc;
                        took = true;
                    }
                    if (!took && is_frame) {
                        int rc = make_wave_job_transposed(ctx, sources[si].frame, p2, tile, SLOT_TRANSPOSED0 + 4 * (size_t)li, &J, &took, &back);
                        if (rc != SMR_OK) return rc;
                        if (took) transposed.push_back(back);
                    }
                    if (!took) {  // the RGBA route
                        SurfView probe = views[si];
                        if (is_frame && !node_ready[si]) { probe.ptr = nullptr; probe.pitch = ((u32)src_w[si] * 4u + 255u) & ~255u; probe.w = src_w[si]; probe.h = src_h[si]; }
                        const bool h_only = p2.axis[0] == 0;
                        if (!h_only || can_fuse_wave_rgba(ctx, probe, p2, tile)) {
                            if (is_frame) {
                                int rc = ensure_node(si);
                                if (rc != SMR_OK) return rc;
                            }
                            if (h_only) {
                                if (can_fuse_wave_rgba(ctx, views[si], p2, tile)) {
                                    int rc = make_wave_job_rgba(ctx, views[si], p2, tile, &J);
                                    if (rc != SMR_OK) return rc;
                                    took = true;
                                }
                            } else {
                                int rc = flush_nodes();
                                if (rc != SMR_OK) return rc;
                                rc = make_wave_job_rgba_transposed(ctx, views[si], p2, tile, SLOT_TRANSPOSED0 + 4 * (size_t)li, &J, &took, &back);
                                if (rc != SMR_OK) return rc;
                                if (took) transposed.push_back(back);
                            }
                            rgba_job = took;
                        }
                    }
                    if (took) {
                        J.perp = perp;
                        (rgba_job ? (kinds[si] == 2 ? wjobs_sa_rgba : wjobs_sa_rgba_alpha) : wjobs_sa).push_back(J);
                        on_mfma = true;
                    }
                }
                // an opaque source the fused conversion does not read (4:2:2, 4:4:4, packed YUV, BGRA / ARGB frames after the exact
                // converter; opaque surfaces): the same matrix-core kernel on its RGBA8 node texture
                if (!on_mfma && fused && kinds[si] != 0) {
                    // (kinds 1: the node has an alpha channel — the four-channel builds)
                    std::vector<WJob> &rgba_jobs = kinds[si] == 2 ? wjobs_rgba : wjobs_rgba_alpha;
                    std::vector<WJob> &f16_jobs = kinds[si] == 2 ? wjobs_f16 : wjobs_f16_alpha;
                    int pcls = -1;
                    if (is_frame && kinds[si] == 2 && !node_ready[si] && !ctx->direct_output && can_fuse_planes(ctx, sources[si].frame, plan, tile, &pcls)) {
                        // the default route of a 4:2:0 frame: the matrix-core kernel on the frame's planes, exact conversion in the wave (no node texture)
                        WJob J;
                        int rc = make_wave_job_planes(ctx, sources[si].frame, plan, tile, &J);
                        if (rc != SMR_OK) return rc;
                        wjobs_planes[J.nv12][pcls].push_back(J);
                        on_mfma = true;
                    }
                    if (!on_mfma && is_frame && kinds[si] == 2 && rgb12_node_serves(ctx, sources[si].frame, plan, tile)) {
                        // SMR_OPT_PLANE_SOURCE off / direct output: exact converter -> RGB12 node -> the matrix-core kernel
                        int rc = ensure_rgb12_node(si);
                        if (rc != SMR_OK) return rc;
                        WJob J;
                        rc = make_wave_job_rgba(ctx, cviews[si], plan, tile, &J, false, true);
                        if (rc != SMR_OK) return rc;
                        wjobs_rgb12.push_back(J); wjob_rgb12_layout.push_back(li);
                        on_mfma = true;
                    }
                    if (!on_mfma && is_frame && !node_ready[si]) {
                        // (only convert when the kernel will take the job: the geometry test needs the node's size, not its pixels)
                        SurfView probe;
                        probe.ptr = nullptr; probe.pitch = ((u32)src_w[si] * 4u + 255u) & ~255u; probe.w = src_w[si]; probe.h = src_h[si];
                        bool probe_single = false;
                        if (can_fuse_wave_rgba(ctx, probe, plan, tile, 4, &probe_single)) {
                            int rc = ensure_node(si);
                            if (rc != SMR_OK) return rc;
                        }
                    }
                    bool single = false;
                    if (!on_mfma && node_ready[si] && can_fuse_wave_rgba(ctx, views[si], plan, tile, 4, &single)) {
                        WJob J;
                        int rc = make_wave_job_rgba(ctx, views[si], plan, tile, &J, single);
                        if (rc != SMR_OK) return rc;
                        rgba_jobs.push_back(J);
                        if (kinds[si] == 2) wjob_rgba_layout.push_back(li);
                        on_mfma = true;
                    }
                    if (!on_mfma && plan.kind == 2 && (plan.levels[0] != 0 || plan.levels[1] != 0) && plan.axis[0] == 1 && plan.axis[1] == 0) {
                        // the same for a vertical-first residual: the box-reduced node transposed in, the tile transposed back
                        smr_surface *reduced = smr_cached_surface(ctx, SLOT_REDUCED0 + li, (u32)plan.reduced_w, (u32)plan.reduced_h, SMR_PX_RGBA16F);
                        smr_surface *reduced_t = smr_cached_surface(ctx, SLOT_TRANSPOSED0 + 4 * (size_t)li, (u32)plan.reduced_h, (u32)plan.reduced_w, SMR_PX_RGBA16F);
                        smr_surface *tile_t = smr_cached_surface(ctx, SLOT_TRANSPOSED0 + 4 * (size_t)li + 3, tile->h, tile->w, SMR_PX_RGBA8);
                        if (!reduced || !reduced_t || !tile_t) return SMR_ERR_OOM;
                        smr_resample_plan pt = plan;
                        pt.axis[0] = 0; pt.axis[1] = 1;
                        bool single = false;
                        if (can_fuse_wave_rgba(ctx, view_of(reduced_t), pt, tile_t, 8, &single)) {
                            if (is_frame) {
                                int rc = ensure_node(si);
                                if (rc != SMR_OK) return rc;
                            }
                            smr_surface node;  // non-owning alias of the node view
                            node.ptr = views[si].ptr; node.pitch = views[si].pitch; node.w = (u32)views[si].w; node.h = (u32)views[si].h;
                            node.fmt = SMR_PX_RGBA8;
                            int rc = flush_nodes();
                            if (rc != SMR_OK) return rc;
                            rc = smr_downsample(ctx, &node, 1u << plan.levels[0], 1u << plan.levels[1], reduced);
                            if (rc != SMR_OK) return rc;
                            rc = launch_transpose<uint2>(ctx, reduced, reduced_t);
                            if (rc != SMR_OK) return rc;
                            WJob J;
                            rc = make_wave_job_rgba(ctx, view_of(reduced_t), pt, tile_t, &J, single);
                            if (rc != SMR_OK) return rc;
                            f16_jobs.push_back(J);
                            MTransposeBack back;
                            back.tile_t = tile_t; back.tile = tile;
                            transposed.push_back(back);
                            on_mfma = true;
                        }
                    }
                    if (!on_mfma && plan.kind == 2 && (plan.levels[0] != 0 || plan.levels[1] != 0) && plan.axis[0] == 0 && plan.axis[1] == 1) {
                        // box-pre-reduced plan (shrink factors from 4): downsample.wgsl's pass as it is, then the residual Lanczos
                        // (scales below 2) on the matrix cores, reading the RGBA16F texels as they are
                        smr_surface *reduced = smr_cached_surface(ctx, SLOT_REDUCED0 + li, (u32)plan.reduced_w, (u32)plan.reduced_h, SMR_PX_RGBA16F);
                        if (!reduced) return SMR_ERR_OOM;
                        bool single = false;
                        if (can_fuse_wave_rgba(ctx, view_of(reduced), plan, tile, 8, &single)) {
                            if (is_frame) {
                                int rc = ensure_node(si);
                                if (rc != SMR_OK) return rc;
                            }
                            smr_surface node;  // non-owning alias of the node view
                            node.ptr = views[si].ptr; node.pitch = views[si].pitch; node.w = (u32)views[si].w; node.h = (u32)views[si].h;
                            node.fmt = SMR_PX_RGBA8;
                            int rc = flush_nodes();
                            if (rc != SMR_OK) return rc;
                            rc = smr_downsample(ctx, &node, 1u << plan.levels[0], 1u << plan.levels[1], reduced);
                            if (rc != SMR_OK) return rc;
                            WJob J;
                            rc = make_wave_job_rgba(ctx, view_of(reduced), plan, tile, &J, single);
                            if (rc != SMR_OK) return rc;
                            f16_jobs.push_back(J);
                            on_mfma = true;
                        }
                    }
                    if (!on_mfma && plan.kind == 2 && plan.levels[0] == 0 && plan.levels[1] == 0 && plan.axis[0] == 1) {  // vertical-first
                        if (is_frame) {
                            int rc = ensure_node(si);
                            if (rc != SMR_OK) return rc;
                        }
                        WJob J;
                        MTransposeBack back;
                        int rc = flush_nodes();
                        if (rc != SMR_OK) return rc;
                        rc = make_wave_job_rgba_transposed(ctx, views[si], plan, tile, SLOT_TRANSPOSED0 + 4 * (size_t)li, &J, &on_mfma, &back);
                        if (rc != SMR_OK) return rc;
                        if (on_mfma) {
                            rgba_jobs.push_back(J);
                            if (kinds[si] == 2) wjob_rgba_layout.push_back(li);  // (never direct: the kernel writes the transposed tile)
                            transposed.push_back(back);
                        }
                    }
                }
                if (on_mfma) {
                } else if (fused && is_frame && can_fuse_ingest(sources[si].frame, plan)) {
                    IngestJob J;
                    int rc = make_ingest_job(ctx, sources[si].frame, plan, tile, &J);
                    if (rc != SMR_OK) return rc;
                    jobs.push_back(J);
                } else {
                    if (is_frame) {
                        int rc = ensure_node(si);
                        if (rc != SMR_OK) return rc;
                    }
                    smr_surface node;  // non-owning alias of the node view for the general resampler
                    node.ptr = views[si].ptr; node.pitch = views[si].pitch; node.w = (u32)views[si].w; node.h = (u32)views[si].h;
                    node.fmt = SMR_PX_RGBA8;
                    int rc = flush_nodes();
                    if (rc != SMR_OK) return rc;
                    rc = smr_resample(ctx, &node, L.crop, tile);
                    if (rc < 0) return rc;
                }
                // ResampledChild::output_crop (resampler.rs:292-299)
                L.crop[0] = 0.0f; L.crop[1] = 0.0f; L.crop[2] = (float)dw; L.crop[3] = (float)dh;
                views[next_view] = view_of(tile);
                // a resampled opaque source stays opaque: alpha = (sum of w) / (sum of w) == 1 exactly in both passes
                kinds[next_view] = kinds[si];
                L.source_index = next_view++;
                resampled = true;
            }
        }
        if (!resampled && is_frame) {
            int rc = ensure_node(si);
            if (rc != SMR_OK) return rc;
        }
    }
    if (int rc = flush_nodes()) return rc;  // (the queued conversions: everything below may read a node texture)

    // ---- parameters -> device (one pinned staging slot, one copy)
    const u32 b_tiles_x = (out_w + B_TILE_W - 1) / B_TILE_W, b_tiles_y = (out_h + B_TILE_H - 1) / B_TILE_H;
    const u32 b_tiles = b_tiles_x * b_tiles_y;
    const size_t order_bytes = 0;
    PackedLayouts packed;
    int rc = smr_pack_layouts(ctx, eff.data(), n, views.data(), kinds.data(), next_view, (int)out_w, (int)out_h, order_bytes + sizeof(MDirect), &packed);
    if (rc != SMR_OK) return rc;
    const u32 n_first = compose_predict(packed, (int)b_tiles_x, (int)b_tiles_y, ctx->compose_bitmap);
    const bool fits_b = fused && (out_w % 2 == 0) && (out_h % 2 == 0) && packed.n <= MAX_LAYOUT_WORDS * 32;
    const bool big_list = packed.n > B_MAX_LAYOUTS || packed.n_masks > B_MAX_MASKS;  // read in place instead of from an LDS copy
    const bool fuse_yuv = fits_b && out && !out_rgba && (out->format == SMR_FRAME_PLANAR_YUV420 || out->format == SMR_FRAME_NV12) &&
                          out->planes[0] && out->planes[1] && (out->format == SMR_FRAME_NV12 || out->planes[2]);
    // an RGBA8 target: a node's texture (out_rgba), or — for the output formats wave B does not write itself (4:2:2, 4:4:4, full-range
    // 4:2:0, RGBA) — a scratch surface the output converter then reads (smr_rgba_to_frame): the same compositor kernel either way
    smr_surface *rgba_target = out_rgba;
    if (!rgba_target && fits_b && !fuse_yuv) {
        rgba_target = smr_cached_surface(ctx, SLOT_TARGET, out_w, out_h, SMR_PX_RGBA8);
        if (!rgba_target) return SMR_ERR_OOM;
    }
    const bool fuse_rgba = fits_b && !fuse_yuv && rgba_target && (((uintptr_t)rgba_target->ptr) % 16 == 0) && (rgba_target->pitch % 16 == 0);
    const bool fuse_out = fuse_yuv || fuse_rgba;

    // ---- tile classes (k_classify_tiles): kept while a layout list repeats (a scene at rest; a few lists per context, so a
    //      nested node and the root do not evict each other), recomputed when it changes.
    //      With SMR_OPT_DIRECT_OUTPUT the tiles the compositor would only copy from an input resampled in this call are written
    //      as Y'CbCr by wave A itself (MDirect) and skipped by wave B.
    MDirect direct;
    memset(&direct, 0, sizeof(direct));
    bool classify_now = false;
    unsigned long long direct_mask = 0;
    TileClassMap *cm = nullptr;
    if (fuse_out) {
#ifndef SMR_ABLATION_BUILDS
        if (fuse_yuv && ctx->direct_output && ctx->ablate == 0) {
            auto mark = [&](std::vector<WJob> &js, const std::vector<u32> &lis) {
                for (size_t j = 0; j < js.size() && j < lis.size(); j++) {
                    const u32 li = lis[j];
                    const DevLayout &D = packed.host_layouts[li];
                    if (li < 64 && (D.flags & DL_ALIGNED) && (D.flags & DL_UNROTATED) && D.src_kind == 2 && D.src.ptr == js[j].dst.ptr && D.ix % 4 == 0 &&
                        D.iy % 2 == 0 && !js[j].single) {
                        direct_mask |= 1ull << li;
                        js[j].layer = (int)li; js[j].ox = D.ix; js[j].oy = D.iy;
                    }
                }
            };
            mark(wjobs, wjob_layout);
            mark(wjobs_rgb12, wjob_rgb12_layout);
            mark(wjobs_rgba, wjob_rgba_layout);
        }
#endif
        // key: everything the classification reads
        const size_t lb = (size_t)packed.n * sizeof(DevLayout), mb = (size_t)packed.n_masks * sizeof(DevMask);
        std::vector<u8> &key = ctx->class_key_scratch;
        key.resize(lb + mb + 24);
        memcpy(key.data(), packed.host_layouts, lb);
        if (mb) memcpy(key.data() + lb, packed.host_masks, mb);
        const u32 tail[4] = {out_w, out_h, (u32)packed.n, (u32)packed.n_masks};
        memcpy(key.data() + lb + mb, tail, 16);
        memcpy(key.data() + lb + mb + 16, &direct_mask, 8);
        if (ctx->class_maps.empty()) ctx->class_maps.resize(4);
        ctx->class_clock++;
        TileClassMap *lru = &ctx->class_maps[0];
        for (TileClassMap &m : ctx->class_maps) {
            if (m.ready && m.key == key) { cm = &m; break; }
            if (m.last_use < lru->last_use) lru = &m;
        }
        if (!cm) {
            cm = lru;
            if (cm->n < b_tiles) {
                if (cm->d_class) (void)hipFree(cm->d_class);
                if (cm->d_direct) (void)hipFree(cm->d_direct);
                if (cm->d_list) (void)hipFree(cm->d_list);
                cm->d_class = nullptr; cm->d_direct = nullptr; cm->d_list = nullptr; cm->n = 0;
                SMR_HIP(ctx, hipMalloc((void **)&cm->d_class, (size_t)b_tiles * sizeof(TileClass)));
                SMR_HIP(ctx, hipMalloc((void **)&cm->d_direct, b_tiles));
                SMR_HIP(ctx, hipMalloc((void **)&cm->d_list, sizeof(TileList) + (size_t)b_tiles * B_AREA_BANDS * sizeof(TileFull)));  // (an entry per band)
                SMR_HIP(ctx, hipMemsetAsync(cm->d_list, 0, sizeof(u32) * B_LIST_COUNTERS, ctx->stream));
                cm->counter = 0;
                cm->n = b_tiles;
            }
            if (!cm->h_count) {
                SMR_HIP(ctx, hipHostMalloc((void **)&cm->h_count, 64, hipHostMallocDefault));
                SMR_HIP(ctx, hipEventCreateWithFlags(&cm->count_ev, hipEventDisableTiming));
            }
            classify_now = true;  // (after the layout list is on the device)
            cm->key = key;
            cm->ready = true;
            cm->count_known = false;
            cm->class_serial++;
        }
        cm->last_use = ctx->class_clock;
        if (ctx->debug_ingest)
            fprintf(stderr, "[smr] tile classes: %s; direct output: layer mask %llx of %zu resampled tiles\n", classify_now ? "classifying" : "cached",
                    direct_mask, wjobs.size() + wjobs_rgb12.size() + wjobs_rgba.size());
        if (direct_mask) {
            direct.cls = cm->d_direct;
            direct.tiles_x = (int)b_tiles_x;
            direct.nv = out->format == SMR_FRAME_NV12 ? 1 : 0;
            direct.yp = view_of(out->planes[0]); direct.up = view_of(out->planes[1]);
            direct.vp = direct.nv ? direct.up : view_of(out->planes[2]);
        }
    }
    memcpy((u8 *)packed.extra_host + order_bytes, &direct, sizeof(direct));
    rc = smr_pack_commit(ctx, &packed);  // (may move the pack's device pointers onto the previous frame's identical copy)
    if (rc != SMR_OK) return rc;
    const MDirect *direct_dev = direct.cls ? (const MDirect *)((const u8 *)packed.extra_dev + order_bytes) : nullptr;
    if (classify_now) {
        if (++cm->counter == B_LIST_COUNTERS) {  // (the ring of list counters: smr_fused_compose.h)
            SMR_HIP(ctx, hipMemsetAsync(cm->d_list, 0, sizeof(u32) * B_LIST_COUNTERS, ctx->stream));
            cm->counter = 0;
        }
        hipLaunchKernelGGL(k_classify_tiles, dim3((b_tiles + B_CLASSIFY_TILES - 1) / B_CLASSIFY_TILES), dim3(64 * B_CLASSIFY_TILES), 0, ctx->stream,
                           packed.layouts, packed.masks, packed.n, (int)out_w, (int)out_h, (int)b_tiles_x, (int)b_tiles, direct_mask, (TileClass *)cm->d_class, cm->d_direct, (TileList *)cm->d_list,
                           ctx->compose_select ? 1 : 0, cm->counter);
        SMR_HIP(ctx, hipGetLastError());
    }
    if (fuse_out) {
        // the length of the list of tiles that need compositing comes back to the host for the frames that reuse these classes
        // (never waited for: one copy in flight at a time, valid only if the map was not reclassified since it was issued)
        if (cm->count_pending && hipEventQuery(cm->count_ev) == hipSuccess) {
            cm->count_pending = false;
            cm->count_known = cm->count_serial == cm->class_serial;
        }
        // (asked for by the first frame that REUSES the classes: a scene in motion classifies every frame and would pay a copy and a marker
        //  packet per frame for lengths nobody reads)
        if (!classify_now && !cm->count_known && !cm->count_pending) {
            SMR_HIP(ctx, hipMemcpyAsync(cm->h_count, (const u32 *)cm->d_list + cm->counter, 4, hipMemcpyDeviceToHost, ctx->stream));
            SMR_HIP(ctx, hipEventRecord(cm->count_ev, ctx->stream));
            cm->count_pending = true;
            cm->count_serial = cm->class_serial;
        }
    }

    // ---- wave A (job descriptors ride in the kernel arguments)
    if (!wjobs.empty()) {
        rc = launch_wave(ctx, wjobs, direct_dev);
        if (rc != SMR_OK) return rc;
    }
    // node-texture jobs go out class by class: launch_wave picks ONE build for a launch, and a scene that mixes classes (a main tile at 1.5x
    // beside thumbnails at 3x) would otherwise run all of them on the generic build (the RGB12 generic build is slower than RGBA8's)
    auto by_class = [&](std::vector<WJob> &jobs, bool rgb12) -> int {
        std::vector<WJob> group[5];
        for (const WJob &J : jobs) {
            const int k = (J.NKS <= 4 && J.KV == 2) ? (J.k01 ? 1 : 0) : (J.NKS <= 8 && J.KV == 3) ? 2 : (J.NKS <= 8 && J.KV == 2) ? 3 : 4;
            group[k].push_back(J);
        }
        for (auto &g : group)
            if (!g.empty())
                if (int r = launch_wave(ctx, g, direct_dev, true, false, false, false, rgb12)) return r;
        return SMR_OK;
    };
    for (auto &by_nv : wjobs_planes)
        for (auto &g : by_nv)
            if (!g.empty())
                if (int r = launch_wave(ctx, g, nullptr, false, false, false, false, false, true)) return r;
    if (!wjobs_rgb12.empty()) {
        rc = by_class(wjobs_rgb12, true);
        if (rc != SMR_OK) return rc;
    }
    if (!wjobs_rgba.empty()) {
        rc = by_class(wjobs_rgba, false);
        if (rc != SMR_OK) return rc;
    }
    if (!wjobs_rgba_alpha.empty()) {
        rc = launch_wave(ctx, wjobs_rgba_alpha, nullptr, true, false, false, true);
        if (rc != SMR_OK) return rc;
    }
    if (!wjobs_f16.empty()) {
        rc = launch_wave(ctx, wjobs_f16, nullptr, true, true);
        if (rc != SMR_OK) return rc;
    }
    if (!wjobs_f16_alpha.empty()) {
        rc = launch_wave(ctx, wjobs_f16_alpha, nullptr, true, true, false, true);
        if (rc != SMR_OK) return rc;
    }
    if (!wjobs_sa_rgba_alpha.empty()) {
        rc = launch_wave(ctx, wjobs_sa_rgba_alpha, nullptr, true, false, true, true);
        if (rc != SMR_OK) return rc;
    }
    if (!wjobs_sa.empty()) {
        rc = launch_wave(ctx, wjobs_sa, nullptr, false, false, true);
        if (rc != SMR_OK) return rc;
    }
    if (!wjobs_sa_rgba.empty()) {
        rc = launch_wave(ctx, wjobs_sa_rgba, nullptr, true, false, true);
        if (rc != SMR_OK) return rc;
    }
    for (const MTransposeBack &b : transposed) {
        rc = launch_transpose<u32>(ctx, b.tile_t, b.tile);
        if (rc != SMR_OK) return rc;
    }
    if (!jobs.empty()) {
        rc = launch_ingest(ctx, jobs);
        if (rc != SMR_OK) return rc;
    }

    // ---- wave B (or the general compositor + output converters)
    if (fuse_out) {
        StageScope scope(ctx, SMR_STAGE_FUSED_COMPOSE);
        ctx->kernel_launches[SMR_KERNEL_COMPOSE_OUTPUT]++;
        // 1-D grid: bands of the tiles that need the (latency-bound) general path first — as many as the class list holds when its
        // length is known, as many as the host's own prediction says otherwise — then every tile in order
        // (list entries are bands: at most B_AREA_BANDS per predicted tile)
        u32 n_banded = cm->count_known ? *cm->h_count : n_first * (u32)B_AREA_BANDS;
        if (n_banded > b_tiles * (u32)B_AREA_BANDS) n_banded = b_tiles * (u32)B_AREA_BANDS;
        if (ctx->debug_ingest)
            fprintf(stderr, "k_compose_output: %u tiles, %u bands on the list (%s; predicted tiles %u, last read back %u)\n", b_tiles, n_banded, cm->count_known ? "known" : "predicted", n_first, *cm->h_count);
        dim3 grid(n_banded + (b_tiles + B_COPY_TILES - 1) / B_COPY_TILES, 1, 1);
        const TileList *full = (const TileList *)cm->d_list;
        const TileClass *tc = (const TileClass *)cm->d_class;
        const int flags = (ctx->srgb() ? 1 : 0) | ((ctx->ablate >> 8) << 8);
        SurfView p0, p1, p2;
        int nv;
        if (fuse_rgba) {
            p0 = p1 = p2 = view_of(rgba_target);
            nv = 2;
        } else {
            p0 = view_of(out->planes[0]); p1 = view_of(out->planes[1]);
            nv = out->format == SMR_FRAME_NV12 ? 1 : 0;
            p2 = nv ? p1 : view_of(out->planes[2]);
        }
        typedef void (*ComposeKernel)(SurfView, SurfView, SurfView, int, int, const DevLayout *, const DevMask *, int, int, int, const float *, int, int,
                                      const TileClass *, const TileList *, int, int);
        static const ComposeKernel kernels[3][2] = {{k_compose_output<0, false>, k_compose_output<0, true>},
                                                    {k_compose_output<1, false>, k_compose_output<1, true>},
                                                    {k_compose_output<2, false>, k_compose_output<2, true>}};
        hipLaunchKernelGGL(kernels[nv][big_list ? 1 : 0], grid, dim3(256), 0, ctx->stream, p0, p1, p2, (int)out_w, (int)out_h, packed.layouts, packed.masks,
                           packed.n, packed.n_masks, flags, ctx->d_tables, (int)b_tiles_x, (int)b_tiles, tc, full, (int)n_banded, cm->counter);
        SMR_HIP(ctx, hipGetLastError());
        rc = smr_pack_done(ctx, &packed);
        if (rc != SMR_OK) return rc;
        if (fuse_rgba && out) return smr_rgba_to_frame(ctx, rgba_target, out);
        return SMR_OK;
    }

    smr_surface *target = out_rgba ? out_rgba : smr_cached_surface(ctx, SLOT_TARGET, out_w, out_h, SMR_PX_RGBA8);
    if (!target) return SMR_ERR_OOM;
    rc = smr_launch_apply_layouts(ctx, target, &packed);
    if (rc != SMR_OK) return rc;
    rc = smr_pack_done(ctx, &packed);
    if (rc != SMR_OK) return rc;
    if (out) return smr_rgba_to_frame(ctx, target, out);
    return SMR_OK;
}

// InputTexture::convert_to_node_texture + ResampledChild::render for one input: the exact converter into a node texture and the
// matrix-core kernel on it (the default), the fused-conversion kernel (SMR_INGEST_LAB_FUSED), the f32 kernel (SMR_INGEST_VALU_F32),
// otherwise convert + general resample.  This is the per-shard step of the multi-GPU path: each GPU turns its inputs into dst-sized tiles.
//
// `new_call`: open a weight-cache call of its own.  The batch entry point passes false: the bands it already handed to jobs that are
// not launched yet carry the batch's call id and must keep their eviction protection while this input builds its own.
static int ingest_resample_one(smr_ctx *ctx, const smr_frame *in, const float crop[4], smr_surface *dst, bool new_call) {
    if (!ctx || !in || !crop || !dst) return SMR_ERR_INVALID;
    if (dst->fmt != SMR_PX_RGBA8) return smr_fail(ctx, SMR_ERR_INVALID, "smr_ingest_resample: dst must be RGBA8");
    if (int rc = smr_validate_frame(ctx, in, "smr_ingest_resample")) return rc;
    if (!ctx->srgb()) return smr_fail(ctx, SMR_ERR_INVALID, "smr_ingest_resample: CpuOptimized mode has no resampler");
    smr_resample_plan plan;
    int kind = smr_resample_plan_make(in->width, in->height, crop, dst->w, dst->h, &plan);
    if (kind < 0) return smr_fail(ctx, kind, "smr_ingest_resample: degenerate plan");
    if (kind == 0) return 0;
    if (new_call) ctx->weight_call++;
    if (!fused_disabled(ctx) && can_fuse_wave(ctx, in, plan, dst)) {  // (fused conversion: opt-in)
        std::vector<WJob> wjobs(1);
        int rc = make_wave_job(ctx, in, plan, dst, &wjobs[0]);
        if (rc != SMR_OK) return rc;
        rc = launch_wave(ctx, wjobs);
        return rc == SMR_OK ? kind : rc;
    }
    if (!fused_disabled(ctx)) {  // ... a vertical-first plan: the same kernel on the transposed frame
        std::vector<WJob> wjobs(1);
        MTransposeBack back;
        bool ok = false;
        int rc = make_wave_job_transposed(ctx, in, plan, dst, SLOT_TRANSPOSED_SINGLE, &wjobs[0], &ok, &back);
        if (rc != SMR_OK) return rc;
        if (ok) {
            rc = launch_wave(ctx, wjobs);
            if (rc == SMR_OK) rc = launch_transpose<u32>(ctx, back.tile_t, back.tile);
            return rc == SMR_OK ? kind : rc;
        }
    }
    // the exact converter into the node texture, then the matrix-core kernel on it (every Y'CbCr format; two-pass plans within the kernel's
    // windows, either pass order)
    if (!fused_disabled(ctx) && can_fuse_planes(ctx, in, plan, dst)) {  // the default route: the frame's planes, converted in the kernel
        std::vector<WJob> wjobs(1);
        int rc = make_wave_job_planes(ctx, in, plan, dst, &wjobs[0]);
        if (rc != SMR_OK) return rc;
        rc = launch_wave(ctx, wjobs, nullptr, false, false, false, false, false, true);
        return rc == SMR_OK ? kind : rc;
    }
    if (!fused_disabled(ctx) && ctx->ingest_impl != SMR_INGEST_VALU_F32 && rgb12_node_serves(ctx, in, plan, dst)) {
        smr_surface *cnode = smr_cached_surface(ctx, SMR_SLOT_INGEST_NODE_RGB12, 3 * in->width, in->height, SMR_PX_R8);
        if (!cnode) return SMR_ERR_OOM;
        const u8 one = 1;
        int rc = smr_frames_to_rgba_batch(ctx, &in, &cnode, 1, &one);
        if (rc != SMR_OK) return rc;
        std::vector<WJob> wjobs(1);
        rc = make_wave_job_rgba(ctx, rgb12_view(cnode, in), plan, dst, &wjobs[0], false, true);
        if (rc != SMR_OK) return rc;
        rc = launch_wave(ctx, wjobs, nullptr, true, false, false, false, true);
        return rc == SMR_OK ? kind : rc;
    }
    if (!fused_disabled(ctx) && in->format <= SMR_FRAME_NV12 && ctx->ingest_impl != SMR_INGEST_VALU_F32) {
        smr_surface *node = smr_cached_surface(ctx, SLOT_INGEST_NODE, in->width, in->height, SMR_PX_RGBA8);
        if (!node) return SMR_ERR_OOM;
        bool single = false;
        if (can_fuse_wave_rgba(ctx, view_of(node), plan, dst, 4, &single)) {
            int rc = smr_frame_to_rgba(ctx, in, node);
            if (rc != SMR_OK) return rc;
            std::vector<WJob> wjobs(1);
            rc = make_wave_job_rgba(ctx, view_of(node), plan, dst, &wjobs[0], single);
            if (rc != SMR_OK) return rc;
            rc = launch_wave(ctx, wjobs, nullptr, true);
            return rc == SMR_OK ? kind : rc;
        }
        if (plan.kind == 2 && plan.levels[0] == 0 && plan.levels[1] == 0 && plan.axis[0] == 1) {  // vertical-first: the node transposed in, the tile back
            int rc = smr_frame_to_rgba(ctx, in, node);
            if (rc != SMR_OK) return rc;
            std::vector<WJob> wjobs(1);
            MTransposeBack back;
            bool ok = false;
            rc = make_wave_job_rgba_transposed(ctx, view_of(node), plan, dst, SLOT_TRANSPOSED_SINGLE, &wjobs[0], &ok, &back);
            if (rc != SMR_OK) return rc;
            if (ok) {
                rc = launch_wave(ctx, wjobs, nullptr, true);
                if (rc == SMR_OK) rc = launch_transpose<u32>(ctx, back.tile_t, back.tile);
                return rc == SMR_OK ? kind : rc;
            }
            return smr_resample(ctx, node, crop, dst);  // (the node is converted already)
        }
    }
    if (!fused_disabled(ctx) && can_fuse_ingest(in, plan)) {  // (the f32 kernel: bit-identical to the pass kernels, whatever the option)
        std::vector<IngestJob> jobs(1);
        int rc = make_ingest_job(ctx, in, plan, dst, &jobs[0]);
        if (rc != SMR_OK) return rc;
        rc = launch_ingest(ctx, jobs);
        return rc == SMR_OK ? kind : rc;
    }
    smr_surface *node = smr_cached_surface(ctx, SLOT_INGEST_NODE, in->width, in->height, SMR_PX_RGBA8);
    if (!node) return SMR_ERR_OOM;
    int rc = smr_frame_to_rgba(ctx, in, node);
    if (rc != SMR_OK) return rc;
    return smr_resample(ctx, node, crop, dst);
}

extern "C" int smr_ingest_resample(smr_ctx *ctx, const smr_frame *in, const float crop[4], smr_surface *dst) {
    SMR_ENTER(ctx);
    return ingest_resample_one(ctx, in, crop, dst, true);
}

// The same for all inputs of a shard at once: the inputs the matrix-core kernel takes share ONE conversion launch (a node texture per
// input) and ONE resampling launch (their rows are balanced over the workgroups together); the others go one by one.  kinds[i] receives
// the plan kind of input i (0 = direct: dst untouched).
extern "C" int smr_ingest_resample_batch(smr_ctx *ctx, const smr_frame *const *in, const float *crops, smr_surface *const *dst, uint32_t n,
                                         int *kinds) {
    SMR_ENTER(ctx);
    if (!ctx || (n && (!in || !crops || !dst))) return SMR_ERR_INVALID;
    if (!ctx->srgb()) return smr_fail(ctx, SMR_ERR_INVALID, "smr_ingest_resample_batch: CpuOptimized mode has no resampler");
    if (n > 1024) return smr_fail(ctx, SMR_ERR_INVALID, "smr_ingest_resample_batch: too many inputs");
    std::vector<IngestJob> jobs;
    std::vector<WJob> wjobs, wjobs_rgba, wjobs_rgb12, wjobs_planes[2][3];
    std::vector<const smr_frame *> conv_in;
    std::vector<smr_surface *> conv_node;
    std::vector<u8> conv_rgb12;
    ++ctx->weight_call;
    for (uint32_t i = 0; i < n; i++) {
        if (!in[i] || !dst[i] || dst[i]->fmt != SMR_PX_RGBA8) return smr_fail(ctx, SMR_ERR_INVALID, "smr_ingest_resample_batch: bad input %u", i);
        if (int rc = smr_validate_frame(ctx, in[i], "smr_ingest_resample_batch")) return rc;
        const float *crop = crops + 4 * i;
        smr_resample_plan plan;
        int kind = smr_resample_plan_make(in[i]->width, in[i]->height, crop, dst[i]->w, dst[i]->h, &plan);
        if (kind < 0) return smr_fail(ctx, kind, "smr_ingest_resample_batch: degenerate plan for input %u", i);
        if (kinds) kinds[i] = kind;
        if (kind == 0) continue;
        const bool fused = !fused_disabled(ctx);
        if (fused && can_fuse_wave(ctx, in[i], plan, dst[i])) {  // (fused conversion: opt-in)
            WJob J;
            int rc = make_wave_job(ctx, in[i], plan, dst[i], &J);
            if (rc != SMR_OK) return rc;
            wjobs.push_back(J);
            continue;
        }
        int pcls = -1;
        if (fused && !fused_conversion(ctx) && can_fuse_planes(ctx, in[i], plan, dst[i], &pcls)) {  // the default route: no node texture
            WJob J;
            int rc = make_wave_job_planes(ctx, in[i], plan, dst[i], &J);
            if (rc != SMR_OK) return rc;
            wjobs_planes[J.nv12][pcls].push_back(J);
            continue;
        }
        if (fused && !fused_conversion(ctx) && ctx->ingest_impl != SMR_INGEST_VALU_F32 && rgb12_node_serves(ctx, in[i], plan, dst[i])) {
            smr_surface *cnode = smr_cached_surface(ctx, SMR_SLOT_NODE_RGB12_0 + i, 3 * in[i]->width, in[i]->height, SMR_PX_R8);
            if (!cnode) return SMR_ERR_OOM;
            WJob J;
            int rc = make_wave_job_rgba(ctx, rgb12_view(cnode, in[i]), plan, dst[i], &J, false, true);
            if (rc != SMR_OK) return rc;
            conv_in.push_back(in[i]);
            conv_node.push_back(cnode);
            conv_rgb12.push_back(1);
            wjobs_rgb12.push_back(J);
            continue;
        }
        if (fused && !fused_conversion(ctx) && ctx->ingest_impl != SMR_INGEST_VALU_F32 && in[i]->format <= SMR_FRAME_NV12) {
            SurfView probe;  // (the geometry test needs the node's size, not its pixels)
            probe.ptr = nullptr; probe.pitch = (in[i]->width * 4u + 255u) & ~255u; probe.w = (int)in[i]->width; probe.h = (int)in[i]->height;
            bool single = false;
            if (can_fuse_wave_rgba(ctx, probe, plan, dst[i], 4, &single)) {
                smr_surface *node = smr_cached_surface(ctx, SLOT_NODE0 + i, in[i]->width, in[i]->height, SMR_PX_RGBA8);
                if (!node) return SMR_ERR_OOM;
                if (can_fuse_wave_rgba(ctx, view_of(node), plan, dst[i], 4, &single)) {
                    WJob J;
                    int rc = make_wave_job_rgba(ctx, view_of(node), plan, dst[i], &J, single);
                    if (rc != SMR_OK) return rc;
                    conv_in.push_back(in[i]);
                    conv_node.push_back(node);
                    conv_rgb12.push_back(0);
                    wjobs_rgba.push_back(J);
                    continue;
                }
            }
        }
        if (fused && can_fuse_ingest(in[i], plan)) {
            IngestJob J;
            int rc = make_ingest_job(ctx, in[i], plan, dst[i], &J);
            if (rc != SMR_OK) return rc;
            jobs.push_back(J);
        } else {
            int rc = ingest_resample_one(ctx, in[i], crop, dst[i], false);  // (inside this call: pending jobs' bands stay protected)
            if (rc < 0) return rc;
        }
    }
    if (!conv_in.empty()) {
        int rc = smr_frames_to_rgba_batch(ctx, conv_in.data(), conv_node.data(), (u32)conv_in.size(), conv_rgb12.data());
        if (rc != SMR_OK) return rc;
        if (!wjobs_rgb12.empty()) rc = launch_wave(ctx, wjobs_rgb12, nullptr, true, false, false, false, true);
        if (rc != SMR_OK) return rc;
        if (!wjobs_rgba.empty()) rc = launch_wave(ctx, wjobs_rgba, nullptr, true);
        if (rc != SMR_OK) return rc;
    }
    for (auto &by_nv : wjobs_planes)
        for (auto &g : by_nv)
            if (!g.empty())
                if (int rc = launch_wave(ctx, g, nullptr, false, false, false, false, false, true)) return rc;
    if (!wjobs.empty()) {
        int rc = launch_wave(ctx, wjobs);
        if (rc != SMR_OK) return rc;
    }
    if (!jobs.empty()) {
        int rc = launch_ingest(ctx, jobs);
        if (rc != SMR_OK) return rc;
    }
    return SMR_OK;
}
