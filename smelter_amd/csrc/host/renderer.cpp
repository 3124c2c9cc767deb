// renderer.cpp — the reference's `Renderer` (smelter-render/src/state.rs:96-252) over the scene engine and the smr_* kernels:
// register inputs / images / built-in shaders, update_scene(output, resolution, format, JSON), render(FrameSet) -> FrameSet.
// Per frame (InnerRenderer::render, state.rs:220-252): populate_inputs (staleness filter, render_loop.rs:19-42), a depth-first
// walk of every output's render graph (render_graph.rs, node.rs:22-181) — input refs, images, text, shader nodes, nested
// layout nodes into RGBA8 node surfaces — and read_outputs (render_loop.rs:59-230) fused into the root layout node's launch.
// SURVEY.md §8 a14 (+ a2, a4); no GPU code here, only calls into the C ABI of this library.
#include <cstring>
#include <deque>
#include <map>
#include <set>

#include "scene.h"

using namespace smr_host;

namespace {

struct ImageRes {
    smr_surface *surface = nullptr;  // premultiplied RGBA8 node texture at the image's own resolution
    uint32_t w = 0, h = 0;
    bool opaque = false;             // every pixel's alpha is 255 (seen at registration; premultiplication keeps the alpha byte)
};

struct Source {
    uint32_t kind = SMR_SOURCE_NONE;
    const smr_surface *surface = nullptr;
    const smr_frame *frame = nullptr;
    uint32_t w = 0, h = 0;
    bool opaque = false;  // every texel's alpha is 255 (the renderer writes its node surfaces itself, so it may know): the compositor then copies
                          // or samples the layer where it would otherwise blend it (SMR_SOURCE_OPAQUE_SURFACE) — the same bytes, fewer of them touched
};

// every YUV-family FrameData variant converts to alpha == 1 (wgpu/format/*_to_rgba.wgsl return vec4(.., 1.0)); BGRA / ARGB / RGBA carry their own
bool frame_opaque(const smr_frame *f) { return f && f->format <= SMR_FRAME_NV12; }

// Does a LayoutNode's output have alpha 255 everywhere?  Yes when one of its layers is opaque — a colour with alpha 1, a texture whose source
// is opaque: bilinear weights are 8-bit fractions and sum to 1 exactly, so an opaque texture samples alpha exactly 1 —, has no rotation, rounded
// corner, border or mask, lies on half-integer coordinates (the coverage arithmetic is then exact) and covers every pixel centre by at least half
// a pixel: its fragment's alpha is exactly 1 there, and whatever is blended above leaves src.a + 1 * (1 - src.a), which stores 255.
// Conservative: `false` only costs the compositor its shortcut.
bool layouts_leave_an_opaque_surface(const std::vector<smr_layout> &layouts, const std::vector<Source> &kids, uint32_t w, uint32_t h) {
    auto half_int = [](float v) { return v * 2.0f == std::floor(v * 2.0f) && std::fabs(v) < 32768.0f; };
    for (const smr_layout &L : layouts) {
        if (L.rotation_degrees != 0.0f || L.border_width != 0.0f || L.masks_len != 0) continue;
        if (L.border_radius[0] != 0.0f || L.border_radius[1] != 0.0f || L.border_radius[2] != 0.0f || L.border_radius[3] != 0.0f) continue;
        if (L.type == 1) {
            if (L.color[3] != 1.0f) continue;
        } else if (L.type == 0) {
            if (L.source_index >= kids.size()) continue;
            const Source &k = kids[L.source_index];
            const bool src_opaque = (k.kind == SMR_SOURCE_FRAME && frame_opaque(k.frame)) || (k.kind == SMR_SOURCE_SURFACE && k.opaque);
            if (!src_opaque) continue;
        } else {
            continue;
        }
        if (!(half_int(L.left) && half_int(L.top) && half_int(L.width) && half_int(L.height))) continue;
        if (L.left <= 0.0f && L.top <= 0.0f && L.left + L.width >= (float)w && L.top + L.height >= (float)h) return true;
    }
    return false;
}

// What one frame in flight owns of an output: its two alternating output frames and the per-node intermediate surfaces.
// (One lane = the reference's renderer; with more lanes consecutive frames are enqueued on different contexts / HIP streams
// and overlap on the device, while the scene — and with it every transition — stays one state advanced by one pts sequence.)
struct OutputLane {
    smr_frame frames[2];
    bool have_frames = false;
    int flip = 0;
    std::vector<smr_surface *> node_surface;   // per graph node, (re)allocated on size change (NodeTexture::ensure_size)
    std::vector<smr_surface *> scaled_image;   // per graph node: an Image node whose size differs from the image's
};

struct Output {
    Scene scene;
    uint32_t w = 0, h = 0, format = SMR_FRAME_PLANAR_YUV420;
    std::deque<OutputLane> lanes;              // one per renderer lane, grown on demand (a deque: callers hold pointers to `frames`)
    OutputLane *l = nullptr;                   // the lane of the frame being rendered
    std::vector<smr_surface *> text_surface;   // per graph node: the rendered glyph run of a Text node (once per scene update; read-only per frame)
};

}  // namespace

struct smr_renderer {
    smr_ctx *ctx = nullptr;           // the context GPU work is issued on: lane_ctx[0] outside smr_renderer_render, the frame's lane inside
    std::vector<smr_ctx *> lane_ctx;  // [0] = the context given to smr_renderer_create, then smr_renderer_add_lane's
    uint64_t frame_no = 0;
    int64_t timeout_ns = 500000000;  // stream_fallback_timeout
    std::set<std::string> inputs;
    std::map<std::string, ImageRes> images;
    std::map<std::string, uint32_t> shaders;  // shader_id -> smr_builtin_shader_id
    smr_text_measure_fn measure = nullptr;    // the caller's text shaper (fitted Text nodes)
    void *measure_user = nullptr;
    smr_fontbook *fontbook = nullptr;         // TextRendererCtx: with a book, Text nodes are measured and drawn here (not owned)
    std::map<std::string, Output> outputs;
    std::string err;
    std::vector<smr_layout> layouts;  // scratch
};

namespace {

int fail(smr_renderer *r, int code, const std::string &msg) {
    if (r) r->err = msg;
    return code;
}
int gpu(smr_renderer *r, int rc, const char *what) {
    if (rc >= 0) return rc;
    return fail(r, rc, std::string(what) + ": " + smr_last_error(r->ctx));
}

void free_surfaces(smr_renderer *r, std::vector<smr_surface *> &v) {
    for (smr_surface *s : v)
        if (s) smr_surface_destroy(r->ctx, s);
    v.clear();
}
void sync_lanes(smr_renderer *r) {
    for (smr_ctx *c : r->lane_ctx) smr_sync(c);
}
void free_lane_frames(smr_renderer *r, OutputLane &l) {
    if (!l.have_frames) return;
    smr_frame_destroy(r->ctx, &l.frames[0]);
    smr_frame_destroy(r->ctx, &l.frames[1]);
    l.have_frames = false;
}
// per-node surfaces belong to one render graph: dropped (and re-created on demand) when the graph changes
void free_node_surfaces(smr_renderer *r, Output &o) {
    const size_t n = o.scene.nodes().size();
    for (OutputLane &l : o.lanes) {
        free_surfaces(r, l.node_surface);
        free_surfaces(r, l.scaled_image);
        l.node_surface.assign(n, nullptr);
        l.scaled_image.assign(n, nullptr);
    }
    free_surfaces(r, o.text_surface);
    o.text_surface.assign(n, nullptr);
}
void free_output(smr_renderer *r, Output &o) {
    free_node_surfaces(r, o);
    for (OutputLane &l : o.lanes) free_lane_frames(r, l);
}
// the lane's share of an output, created the first time the lane renders it
int enter_lane(smr_renderer *r, Output &o, size_t lane) {
    if (o.lanes.size() <= lane) o.lanes.resize(lane + 1);
    OutputLane &l = o.lanes[lane];
    o.l = &l;
    const size_t n = o.scene.nodes().size();
    if (l.node_surface.size() != n) { l.node_surface.assign(n, nullptr); l.scaled_image.assign(n, nullptr); }
    if (!l.have_frames) {
        int rc = gpu(r, smr_frame_create(r->ctx, o.format, o.w, o.h, &l.frames[0]), "output frame");
        if (rc < 0) return rc;
        rc = gpu(r, smr_frame_create(r->ctx, o.format, o.w, o.h, &l.frames[1]), "output frame");
        if (rc < 0) {  // (both or neither: the next attempt starts from nothing)
            smr_frame_destroy(r->ctx, &l.frames[0]);
            return rc;
        }
        l.have_frames = true;
    }
    return 0;
}

// NodeTexture::ensure_size (state/node_texture.rs:22-42)
int ensure_surface(smr_renderer *r, smr_surface *&slot, uint32_t w, uint32_t h) {
    smr_surface_info info;
    if (slot && smr_surface_info_get(slot, &info) == 0 && info.width == w && info.height == h) return 0;
    if (slot) { smr_surface_destroy(r->ctx, slot); slot = nullptr; }
    return gpu(r, smr_surface_create(r->ctx, w, h, SMR_PX_RGBA8, &slot), "node surface");
}

struct FrameSetView {
    const smr_input_frame *frames;
    uint32_t n;
    int64_t pts_ns;
};

// RenderNode::render (state/node.rs) for one node of one output; returns what its parent samples
// ShaderParam::to_bytes (shader/node.rs:95-112) over the API form {"type": f32|u32|i32|list|struct, "value": ...}
static void flatten_shader_param(const Json &j, std::vector<uint8_t> &out) {
    const Json *ty = j.get("type"), *v = j.get("value");
    if (!ty || !v) return;
    auto put = [&](const void *p) { const uint8_t *b = (const uint8_t *)p; out.insert(out.end(), b, b + 4); };
    if (ty->str == "f32") { float f = (float)v->num; put(&f); }
    else if (ty->str == "u32") { uint32_t u = as_u32(v->num); put(&u); }
    else if (ty->str == "i32") { int32_t n = as_i32(v->num); put(&n); }
    else if (v->kind == Json::Array) for (const Json &e : v->arr) flatten_shader_param(e, out);
}

int render_node(smr_renderer *r, Output &o, int idx, const FrameSetView &fs, Source &out) {
    const GraphNode &g = o.scene.nodes()[idx];
    const Stateful &c = *g.component;
    out = Source();
    switch (g.kind) {
    case Kind::InputStream: {
        // populate_inputs: a registered input with a fresh enough frame, else the node has no texture
        if (!r->inputs.count(c.ref_id)) return 0;
        for (uint32_t i = 0; i < fs.n; i++) {
            const smr_input_frame &f = fs.frames[i];
            if (!f.input_id || !f.frame || c.ref_id != f.input_id) continue;
            const int64_t oldest = fs.pts_ns > r->timeout_ns ? fs.pts_ns - r->timeout_ns : 0;  // Duration::saturating_sub
            if (oldest > f.pts_ns) return 0;
            out.kind = SMR_SOURCE_FRAME; out.frame = f.frame; out.w = f.frame->width; out.h = f.frame->height;
            return 0;
        }
        return 0;
    }
    case Kind::Image: {
        auto it = r->images.find(c.ref_id);
        if (it == r->images.end()) return 0;
        const uint32_t w = as_u32(c.leaf_size.width), h = as_u32(c.leaf_size.height);
        if (w == it->second.w && h == it->second.h) {
            out.kind = SMR_SOURCE_SURFACE; out.surface = it->second.surface; out.w = w; out.h = h; out.opaque = it->second.opaque;
            return 0;
        }
        if (w == 0 || h == 0) return 0;
        // the image pass draws the asset into the node's own resolution (bitmap_image.rs:65-88): bilinear
        smr_surface *&dst = o.l->scaled_image[idx];
        int rc = ensure_surface(r, dst, w, h);
        if (rc < 0) return rc;
        rc = gpu(r, smr_rescale_bilinear(r->ctx, it->second.surface, dst), "image node");
        if (rc < 0) return rc;
        out.kind = SMR_SOURCE_SURFACE; out.surface = dst; out.w = w; out.h = h;
        return 0;
    }
    case Kind::Text: {
        if (o.text_surface[idx]) {
            out.kind = SMR_SOURCE_SURFACE; out.surface = o.text_surface[idx];
            out.w = as_u32(c.leaf_size.width); out.h = as_u32(c.leaf_size.height);
        }
        return 0;
    }
    default: break;
    }
    // nodes with children: render them first (depth first, node.rs:167-181)
    std::vector<Source> kids(g.children.size());
    for (size_t k = 0; k < g.children.size(); k++) {
        int rc = render_node(r, o, g.children[k], fs, kids[k]);
        if (rc < 0) return rc;
    }
    if (g.kind == Kind::Shader) {
        auto it = r->shaders.find(c.ref_id);
        if (it == r->shaders.end()) return fail(r, -1, "Shader \"" + c.ref_id + "\" does not exist. You have to register it first before using it in the scene definition.");
        const uint32_t w = as_u32(c.leaf_size.width), h = as_u32(c.leaf_size.height);
        if (w == 0 || h == 0) return 0;
        int rc = ensure_surface(r, o.l->node_surface[idx], w, h);
        if (rc < 0) return rc;
        // shader nodes sample RGBA node textures: raw input frames are converted first (InputTexture::convert_to_node_texture)
        std::vector<const smr_surface *> srcs;
        std::vector<bool> src_opaque;
        for (size_t k = 0; k < kids.size(); k++) {
            if (kids[k].kind == SMR_SOURCE_FRAME) {
                smr_surface *&t = o.l->node_surface[g.children[k]];
                rc = ensure_surface(r, t, kids[k].w, kids[k].h);
                if (rc < 0) return rc;
                rc = gpu(r, smr_frame_to_rgba(r->ctx, kids[k].frame, t), "input node texture");
                if (rc < 0) return rc;
                srcs.push_back(t);
                src_opaque.push_back(frame_opaque(kids[k].frame));
            } else if (kids[k].kind == SMR_SOURCE_SURFACE) {
                srcs.push_back(kids[k].surface);
                src_opaque.push_back(kids[k].opaque);
            }
        }
        bool out_opaque = false;
        // the @group(1) uniform: ShaderParam::to_bytes (shader/node.rs:95-112), the values in order, little endian, no padding
        std::vector<uint8_t> params;
        flatten_shader_param(c.shader_param, params);
        if (it->second == SMR_SHADER_GAUSSIAN_BLUR) {
            if (srcs.empty()) return 0;  // nothing to sample: the node stays empty
            // the blur kernel maps texel to texel: a source of another size is first brought to the node's resolution
            const smr_surface *src0 = srcs[0];
            smr_surface_info si;
            smr_surface_info_get(src0, &si);
            if (si.width != w || si.height != h) {
                smr_surface *&t = o.l->scaled_image[idx];
                rc = ensure_surface(r, t, w, h);
                if (rc < 0) return rc;
                rc = gpu(r, smr_rescale_bilinear(r->ctx, src0, t), "shader source");
                if (rc < 0) return rc;
                srcs[0] = t;
            } else {
                // The blur of an opaque texture is opaque: both passes accumulate alpha as sum += 1 * w in the very order they accumulate the
                // weights' own sum, so alpha = sum / sum == 1 exactly and the store writes 255 (k_gauss_axis, smr_misc.hip).  (Not claimed
                // through the rescale above: its weights are not 8-bit fractions.)
                out_opaque = src_opaque[0];
            }
            if (params.size() < sizeof(smr_gaussian_blur_params)) params.assign(sizeof(smr_gaussian_blur_params), 0);
        }
        rc = gpu(r, smr_builtin_shader(r->ctx, it->second, params.data(), params.size(), srcs.data(), (uint32_t)srcs.size(), o.l->node_surface[idx],
                                       (float)((double)fs.pts_ns / 1e9)),
                 "shader node");
        if (rc < 0) return rc;
        out.kind = SMR_SOURCE_SURFACE; out.surface = o.l->node_surface[idx]; out.w = w; out.h = h; out.opaque = out_opaque;
        return 0;
    }
    // layout node (nested): LayoutNode::render into an RGBA8 node surface
    std::vector<std::optional<Size>> res(kids.size());
    std::vector<smr_source> srcs(kids.size());
    for (size_t k = 0; k < kids.size(); k++) {
        if (kids[k].kind != SMR_SOURCE_NONE) res[k] = Size{(float)kids[k].w, (float)kids[k].h};
        srcs[k].kind = (kids[k].kind == SMR_SOURCE_SURFACE && kids[k].opaque) ? (uint32_t)SMR_SOURCE_OPAQUE_SURFACE : kids[k].kind;
        srcs[k].surface = kids[k].surface; srcs[k].frame = kids[k].frame;
    }
    uint32_t w = 0, h = 0;
    std::string err;
    if (!o.scene.node_layouts(idx, fs.pts_ns, res, smr_ctx_mode(r->ctx) == SMR_MODE_GPU_OPTIMIZED, r->layouts, w, h, err)) return fail(r, -1, err);
    if (w == 0 || h == 0) return 0;
    const bool node_opaque = layouts_leave_an_opaque_surface(r->layouts, kids, w, h);
    int rc = ensure_surface(r, o.l->node_surface[idx], w, h);
    if (rc < 0) return rc;
    rc = gpu(r, smr_render_layouts(r->ctx, r->layouts.data(), (uint32_t)r->layouts.size(), srcs.data(), (uint32_t)srcs.size(), w, h, nullptr,
                                   o.l->node_surface[idx]),
             "layout node");
    if (rc < 0) return rc;
    out.kind = SMR_SOURCE_SURFACE; out.surface = o.l->node_surface[idx]; out.w = w; out.h = h; out.opaque = node_opaque;
    return 0;
}

int render_output(smr_renderer *r, Output &o, const FrameSetView &fs, const smr_frame **result) {
    o.l->flip ^= 1;
    smr_frame *target = &o.l->frames[o.l->flip];
    *result = target;
    if (o.scene.nodes().empty()) return gpu(r, smr_frame_fill_black(r->ctx, target), "empty output");
    const GraphNode &root = o.scene.nodes()[0];
    if (root.component->is_layout()) {
        std::vector<Source> kids(root.children.size());
        for (size_t k = 0; k < root.children.size(); k++) {
            int rc = render_node(r, o, root.children[k], fs, kids[k]);
            if (rc < 0) return rc;
        }
        std::vector<std::optional<Size>> res(kids.size());
        std::vector<smr_source> srcs(kids.size());
        for (size_t k = 0; k < kids.size(); k++) {
            if (kids[k].kind != SMR_SOURCE_NONE) res[k] = Size{(float)kids[k].w, (float)kids[k].h};
            srcs[k].kind = (kids[k].kind == SMR_SOURCE_SURFACE && kids[k].opaque) ? (uint32_t)SMR_SOURCE_OPAQUE_SURFACE : kids[k].kind;
            srcs[k].surface = kids[k].surface; srcs[k].frame = kids[k].frame;
        }
        uint32_t w = 0, h = 0;
        std::string err;
        if (!o.scene.node_layouts(0, fs.pts_ns, res, smr_ctx_mode(r->ctx) == SMR_MODE_GPU_OPTIMIZED, r->layouts, w, h, err)) return fail(r, -1, err);
        if (w == o.w && h == o.h && o.format != SMR_FRAME_RGBA) {
            // LayoutNode::render + read_outputs in one go: the root's RGBA target never exists
            return gpu(r, smr_render_layouts(r->ctx, r->layouts.data(), (uint32_t)r->layouts.size(), srcs.data(), (uint32_t)srcs.size(), w, h,
                                             target, nullptr),
                       "root layout node");
        }
        // a root whose own width/height differ from the output resolution: render the node, then convert like any other root
        if (w == 0 || h == 0) return gpu(r, smr_frame_fill_black(r->ctx, target), "empty output");
        int rc = ensure_surface(r, o.l->node_surface[0], w, h);
        if (rc < 0) return rc;
        rc = gpu(r, smr_render_layouts(r->ctx, r->layouts.data(), (uint32_t)r->layouts.size(), srcs.data(), (uint32_t)srcs.size(), w, h, nullptr,
                                       o.l->node_surface[0]),
                 "root layout node");
        if (rc < 0) return rc;
        smr_surface *&t = o.l->scaled_image[0];
        rc = ensure_surface(r, t, o.w, o.h);
        if (rc < 0) return rc;
        rc = gpu(r, smr_rescale_bilinear(r->ctx, o.l->node_surface[0], t), "root rescale");
        if (rc < 0) return rc;
        return gpu(r, smr_rgba_to_frame(r->ctx, t, target), "output conversion");
    }
    Source s;
    int rc = render_node(r, o, 0, fs, s);
    if (rc < 0) return rc;
    if (s.kind == SMR_SOURCE_NONE) return gpu(r, smr_frame_fill_black(r->ctx, target), "empty output");  // render_loop.rs:127-139
    const smr_surface *rgba = s.surface;
    if (s.kind == SMR_SOURCE_FRAME) {
        rc = ensure_surface(r, o.l->node_surface[0], s.w, s.h);
        if (rc < 0) return rc;
        rc = gpu(r, smr_frame_to_rgba(r->ctx, s.frame, o.l->node_surface[0]), "input node texture");
        if (rc < 0) return rc;
        rgba = o.l->node_surface[0];
    }
    if (s.w != o.w || s.h != o.h) {  // the output converters sample the root texture over the whole output (rgba_to_yuv.rs:74-117)
        smr_surface *&t = o.l->scaled_image[0];
        rc = ensure_surface(r, t, o.w, o.h);
        if (rc < 0) return rc;
        rc = gpu(r, smr_rescale_bilinear(r->ctx, rgba, t), "root rescale");
        if (rc < 0) return rc;
        rgba = t;
    }
    return gpu(r, smr_rgba_to_frame(r->ctx, rgba, target), "output conversion");
}

}  // namespace

extern "C" {

SMR_API int smr_renderer_create(smr_ctx *ctx, int64_t stream_fallback_timeout_ns, smr_renderer **out) {
    if (!ctx || !out) return -1;
    smr_renderer *r = new smr_renderer();
    r->ctx = ctx;
    r->lane_ctx.push_back(ctx);
    if (stream_fallback_timeout_ns >= 0) r->timeout_ns = stream_fallback_timeout_ns;
    r->shaders["gaussian_blur"] = SMR_SHADER_GAUSSIAN_BLUR;
    *out = r;
    return 0;
}

SMR_API void smr_renderer_destroy(smr_renderer *r) {
    if (!r) return;
    sync_lanes(r);
    for (auto &kv : r->outputs) free_output(r, kv.second);
    for (auto &kv : r->images)
        if (kv.second.surface) smr_surface_destroy(r->ctx, kv.second.surface);
    delete r;
}

SMR_API const char *smr_renderer_last_error(const smr_renderer *r) { return r ? r->err.c_str() : "null renderer"; }

SMR_API int smr_renderer_register_input(smr_renderer *r, const char *input_id) {
    if (!r || !input_id) return fail(r, -1, "smr_renderer_register_input: null argument");
    r->inputs.insert(input_id);
    return 0;
}
SMR_API int smr_renderer_unregister_input(smr_renderer *r, const char *input_id) {
    if (!r || !input_id) return fail(r, -1, "smr_renderer_unregister_input: null argument");
    r->inputs.erase(input_id);
    return 0;
}

SMR_API int smr_renderer_register_image(smr_renderer *r, const char *image_id, const uint8_t *rgba, uint32_t width, uint32_t height) {
    if (!r || !image_id || !rgba || !width || !height) return fail(r, -1, "smr_renderer_register_image: null argument");
    if (r->images.count(image_id)) return fail(r, -1, std::string("Failed to register an image. Image \"") + image_id + "\" is already registered.");
    // BitmapAsset: straight-alpha pixels uploaded once, premultiplied into the node texture format (bitmap_image.rs:20-88)
    smr_surface *raw = nullptr, *pm = nullptr;
    int rc = gpu(r, smr_surface_create(r->ctx, width, height, SMR_PX_RGBA8, &raw), "image upload");
    if (rc < 0) return rc;
    rc = gpu(r, smr_surface_upload(r->ctx, raw, rgba, (size_t)width * 4), "image upload");
    if (rc >= 0) rc = gpu(r, smr_surface_create(r->ctx, width, height, SMR_PX_RGBA8, &pm), "image upload");
    if (rc >= 0) rc = gpu(r, smr_add_premultiplied_alpha(r->ctx, raw, pm), "image premultiply");
    if (rc >= 0) rc = gpu(r, smr_sync(r->ctx), "image upload");
    smr_surface_destroy(r->ctx, raw);
    if (rc < 0) {
        if (pm) smr_surface_destroy(r->ctx, pm);
        return rc;
    }
    ImageRes res;
    res.surface = pm; res.w = width; res.h = height;
    res.opaque = true;
    for (size_t i = 0, n = (size_t)width * height; i < n && res.opaque; i++) res.opaque = rgba[4 * i + 3] == 255;
    r->images[image_id] = res;
    for (auto &kv : r->outputs) kv.second.scene.register_image(image_id, (float)width, (float)height);
    return 0;
}

SMR_API int smr_renderer_set_text_measurer(smr_renderer *r, smr_text_measure_fn fn, void *user) {
    if (!r) return -1;
    r->measure = fn;
    r->measure_user = user;
    for (auto &kv : r->outputs) kv.second.scene.set_text_measurer(fn, user);
    return 0;
}

SMR_API int smr_renderer_set_fontbook(smr_renderer *r, smr_fontbook *book) {
    if (!r) return -1;
    r->fontbook = book;
    r->measure = book ? smr_fontbook_measure : nullptr;
    r->measure_user = book;
    for (auto &kv : r->outputs) kv.second.scene.set_text_measurer(r->measure, r->measure_user);
    return 0;
}

SMR_API int smr_renderer_register_shader(smr_renderer *r, const char *shader_id, uint32_t builtin_id) {
    if (!r || !shader_id) return fail(r, -1, "smr_renderer_register_shader: null argument");
    if (builtin_id > SMR_SHADER_SILLY) return fail(r, -1, "smr_renderer_register_shader: unknown built-in shader (user WGSL is not supported)");
    r->shaders[shader_id] = builtin_id;
    return 0;
}

// TextRendererNode::render for one node (text_renderer.rs:72-167): clear to the background colour, blit the glyph run — once per update
static int set_text_run(smr_renderer *r, Output &o, int node, const float bg[4], const smr_glyph *glyphs, uint32_t n, const uint8_t *atlas,
                        uint32_t atlas_w, uint32_t atlas_h) {
    const Stateful &c = *o.scene.nodes()[node].component;
    const uint32_t w = as_u32(c.leaf_size.width), h = as_u32(c.leaf_size.height);
    if (!w || !h) return 0;  // (a zero-sized text node is the 1x1 transparent texture: text_renderer.rs:77-85)
    int rc = ensure_surface(r, o.text_surface[node], w, h);
    if (rc < 0) return rc;
    rc = gpu(r, smr_blit_glyphs(r->ctx, o.text_surface[node], bg, glyphs, n, atlas, atlas_w, atlas_h), "text node");
    // every lane's stream reads the run from its next frame on
    if (rc >= 0 && r->lane_ctx.size() > 1) rc = gpu(r, smr_sync(r->ctx), "text node");
    return rc;
}

// With a font book the renderer is its own TextRendererCtx: every Text node of the new scene is laid out at its resolved size
// (layout_text's final set_size: the node's width, text_renderer.rs:333-343), rasterised and drawn — in TextComponent's colour
// (glyphon::Color::rgba of the straight bytes, text_renderer.rs:176-177) over convert_to_shader_color(background_color) (:62, 371-374).
static int draw_text_nodes(smr_renderer *r, const std::string &output_id, Output &o) {
    (void)output_id;
    const auto &nodes = o.scene.nodes();
    for (int i = 0; i < (int)nodes.size(); i++) {
        if (nodes[i].kind != Kind::Text) continue;
        const Stateful &c = *nodes[i].component;
        const Stateful::TextSpec &t = c.text_spec;
        smr_text_params p;
        memset(&p, 0, sizeof(p));
        p.text = c.text.c_str(); p.font_family = t.family.c_str(); p.style = t.style.c_str(); p.weight = t.weight.c_str(); p.wrap = t.wrap.c_str();
        p.align = t.align.c_str();
        p.font_size = t.font_size; p.line_height = t.line_height;
        p.max_width = c.leaf_size.width; p.max_height = c.leaf_size.height;
        const float color[4] = {t.color.r / 255.0f, t.color.g / 255.0f, t.color.b / 255.0f, t.color.a / 255.0f};
        float bg[4];
        convert_to_shader_color(t.background, smr_ctx_mode(r->ctx) == SMR_MODE_GPU_OPTIMIZED, bg);
        smr_text_run run;
        if (smr_fontbook_rasterise(r->fontbook, &p, as_u32(c.leaf_size.width), as_u32(c.leaf_size.height), color, &run) != 0)
            return fail(r, -1, std::string("text node: ") + smr_fontbook_last_error(r->fontbook));
        const int rc = set_text_run(r, o, i, bg, run.glyphs, run.n_glyphs, run.atlas, run.atlas_w, run.atlas_h);
        if (rc < 0) return rc;
    }
    return 0;
}

SMR_API int smr_renderer_update_scene(smr_renderer *r, const char *output_id, uint32_t width, uint32_t height, uint32_t output_format,
                                      const char *scene_json) {
    if (!r || !output_id || !scene_json || !width || !height) return fail(r, -1, "smr_renderer_update_scene: null argument");
    if (output_format != SMR_FRAME_PLANAR_YUV420 && output_format != SMR_FRAME_PLANAR_YUV422 && output_format != SMR_FRAME_PLANAR_YUV444 &&
        output_format != SMR_FRAME_NV12 && output_format != SMR_FRAME_RGBA)
        return fail(r, -1, "smr_renderer_update_scene: output format must be planar YUV 4:2:0 / 4:2:2 / 4:4:4, NV12 or RGBA");
    Output &o = r->outputs[output_id];
    for (auto &kv : r->images) o.scene.register_image(kv.first, (float)kv.second.w, (float)kv.second.h);
    o.scene.set_text_measurer(r->measure, r->measure_user);
    std::string err;
    {
        // Shader ids are resolved while the new scene is built (ShaderComponent::stateful_component, shader_component.rs:44-52):
        // an unknown id is SceneError::ShaderNotFound and the previous scene stays.  So: convert the definition without
        // touching the active scene (Scene::parse), look the ids up, and only then update.
        std::string converted;
        if (!o.scene.parse(scene_json, converted, err)) {
            if (o.w == 0) r->outputs.erase(output_id);  // a failed first update leaves no output behind
            return fail(r, -1, err);
        }
        Json tree;
        JsonParser jp(converted);
        std::string missing, text_err;
        if (jp.parse(tree, err)) {
            std::vector<const Json *> stack{&tree};
            while (!stack.empty() && missing.empty() && text_err.empty()) {
                const Json *j = stack.back();
                stack.pop_back();
                for (const Json &a : j->arr) stack.push_back(&a);
                for (const auto &kv : j->obj) stack.push_back(&kv.second);
                const Json *type = j->get("type"), *sid = j->get("shader_id");
                if (type && sid && type->str == "Shader" && !r->shaders.count(sid->str)) missing = sid->str;
                // The renderer draws its Text nodes itself (smr_renderer_set_fontbook) once the new scene is in place; what can make that fail for
                // reasons the scene carries — an empty font book, text that is not UTF-8, sizes no run can have — is found HERE, while the previous
                // scene is still the active one (the reference fails an update as a whole: state.rs:177-189).  smr_fontbook_measure runs
                // the rasteriser's own checks (text.cpp: measure / rasterise share sane_sizes, the font match and the decoder).
                if (type && type->str == "Text" && r->fontbook) {
                    auto str = [&](const char *k) { const Json *v = j->get(k); return v ? v->str : std::string(); };
                    auto num = [&](const char *k, double d) { const Json *v = j->get(k); return v ? v->num : d; };
                    const std::string text = str("text"), family = str("font_family"), style = str("style"), weight = str("weight"), wrap = str("wrap");
                    smr_text_params p;
                    memset(&p, 0, sizeof(p));
                    p.text = text.c_str(); p.font_family = family.c_str(); p.style = style.c_str(); p.weight = weight.c_str(); p.wrap = wrap.c_str();
                    p.align = "Left";
                    p.font_size = (float)num("font_size", 0.0); p.line_height = (float)num("line_height", num("font_size", 0.0));
                    p.max_width = 7682.0f; p.max_height = 4320.0f;
                    float widest = 0.0f;
                    uint32_t lines = 0;
                    if (smr_fontbook_measure(r->fontbook, &p, &widest, &lines) != 0) text_err = std::string("text node: ") + smr_fontbook_last_error(r->fontbook);
                }
            }
        }
        if (!missing.empty()) {
            if (o.w == 0) r->outputs.erase(output_id);
            return fail(r, -1, "Shader \"" + missing + "\" does not exist. You have to register it first before using it in the scene definition.");
        }
        if (!text_err.empty()) {
            if (o.w == 0) r->outputs.erase(output_id);
            return fail(r, -1, text_err);
        }
    }
    if (!o.scene.update(scene_json, width, height, err)) {
        if (o.w == 0) r->outputs.erase(output_id);  // a failed first update leaves no output behind
        return fail(r, -1, err);
    }
    // frames in flight read the surfaces dropped below
    sync_lanes(r);
    if (o.w != width || o.h != height || o.format != output_format)
        for (OutputLane &l : o.lanes) free_lane_frames(r, l);  // re-created at the new geometry when the lane next renders
    o.w = width; o.h = height; o.format = output_format;
    // node indices belong to the new graph: per-node surfaces are re-created on demand, text runs must be supplied again
    free_node_surfaces(r, o);
    int rc = enter_lane(r, o, 0);  // the output's first frames exist after a successful update, as before
    // (what is left to fail below is a device allocation — the frames, a text node's surface: SMR_ERR_OOM with the NEW scene active and the
    //  nodes that could not be drawn transparent; include/smr.h says so.  Everything the scene itself can get wrong was refused above.)
    if (rc >= 0 && r->fontbook) rc = draw_text_nodes(r, output_id, o);
    return rc;
}

SMR_API int smr_renderer_unregister_output(smr_renderer *r, const char *output_id) {
    if (!r || !output_id) return fail(r, -1, "smr_renderer_unregister_output: null argument");
    auto it = r->outputs.find(output_id);
    if (it == r->outputs.end()) return 0;
    sync_lanes(r);
    free_output(r, it->second);
    r->outputs.erase(it);
    return 0;
}

SMR_API int smr_renderer_node_count(const smr_renderer *r, const char *output_id) {
    if (!r || !output_id) return -1;
    auto it = r->outputs.find(output_id);
    return it == r->outputs.end() ? -1 : (int)it->second.scene.nodes().size();
}

SMR_API int smr_renderer_node_info(smr_renderer *r, const char *output_id, int node, smr_scene_node *out) {
    if (!r || !output_id || !out) return fail(r, -1, "smr_renderer_node_info: null argument");
    auto it = r->outputs.find(output_id);
    if (it == r->outputs.end()) return fail(r, -1, "smr_renderer_node_info: unknown output");
    const auto &nodes = it->second.scene.nodes();
    if (node < 0 || node >= (int)nodes.size()) return fail(r, -1, "smr_renderer_node_info: node index out of range");
    const GraphNode &g = nodes[node];
    const Stateful &c = *g.component;
    memset(out, 0, sizeof(*out));
    out->kind = g.kind == Kind::InputStream ? SMR_NODE_INPUT_STREAM : g.kind == Kind::Text ? SMR_NODE_TEXT : g.kind == Kind::Image ? SMR_NODE_IMAGE
              : g.kind == Kind::Shader ? SMR_NODE_SHADER : SMR_NODE_LAYOUT;
    out->parent = g.parent;
    out->n_children = (uint32_t)g.children.size();
    const Size sz = g.has_forced_size ? g.forced_size : c.leaf_size;
    out->width = as_u32(sz.width); out->height = as_u32(sz.height);
    out->ref_id = c.ref_id.c_str(); out->id = c.id.c_str();
    out->payload = c.kind == Kind::Text ? c.text.c_str() : "";
    return 0;
}

SMR_API int smr_renderer_set_text(smr_renderer *r, const char *output_id, int node, const float bg[4], const smr_glyph *glyphs, uint32_t n,
                                  const uint8_t *atlas, uint32_t atlas_w, uint32_t atlas_h) {
    if (!r || !output_id || !bg) return fail(r, -1, "smr_renderer_set_text: null argument");
    auto it = r->outputs.find(output_id);
    if (it == r->outputs.end()) return fail(r, -1, "smr_renderer_set_text: unknown output");
    Output &o = it->second;
    if (node < 0 || node >= (int)o.scene.nodes().size() || o.scene.nodes()[node].kind != Kind::Text)
        return fail(r, -1, "smr_renderer_set_text: not a text node");
    return set_text_run(r, o, node, bg, glyphs, n, atlas, atlas_w, atlas_h);
}

SMR_API int smr_renderer_render(smr_renderer *r, int64_t pts_ns, const smr_input_frame *inputs, uint32_t n_inputs, smr_output_frame *outputs,
                                uint32_t cap, uint32_t *n_outputs) {
    if (!r || (n_inputs && !inputs) || !n_outputs) return fail(r, -1, "smr_renderer_render: null argument");
    FrameSetView fs{inputs, n_inputs, pts_ns};
    // consecutive frames rotate through the lanes: this frame's GPU work goes to its lane's context (stream, scratch) and
    // into the lane's own output frames and node surfaces; the scene state is shared and advances with pts as ever
    const size_t lane = (size_t)(r->frame_no++ % r->lane_ctx.size());
    smr_ctx *base = r->ctx;
    r->ctx = r->lane_ctx[lane];
    uint32_t k = 0;
    int rc = 0;
    for (auto &kv : r->outputs) {
        const smr_frame *frame = nullptr;
        rc = enter_lane(r, kv.second, lane);
        if (rc >= 0) rc = render_output(r, kv.second, fs, &frame);
        if (rc < 0) break;
        if (k < cap && outputs) { outputs[k].output_id = kv.first.c_str(); outputs[k].frame = frame; outputs[k].ctx = r->ctx; }
        k++;
    }
    r->ctx = base;
    if (rc < 0) return rc;
    *n_outputs = k;
    return 0;
}

SMR_API int smr_renderer_add_lane(smr_renderer *r, smr_ctx *ctx) {
    if (!r || !ctx) return fail(r, -1, "smr_renderer_add_lane: null argument");
    for (smr_ctx *c : r->lane_ctx)
        if (c == ctx) return fail(r, -1, "smr_renderer_add_lane: this context is a lane already");
    if (smr_ctx_mode(ctx) != smr_ctx_mode(r->ctx)) return fail(r, -1, "smr_renderer_add_lane: the lane's context has another rendering mode");
    r->lane_ctx.push_back(ctx);
    // frames in flight: every lane's kernels run beside another lane's from now on
    for (smr_ctx *c : r->lane_ctx) (void)smr_ctx_set_option(c, SMR_OPT_SHARED_DEVICE, 1);
    return 0;
}

SMR_API int smr_renderer_sync(smr_renderer *r) {
    if (!r) return -1;
    int rc = 0;
    for (smr_ctx *c : r->lane_ctx) {
        const int e = smr_sync(c);
        if (e < 0 && rc >= 0) rc = gpu(r, e, "sync");
    }
    return rc;
}

}  // extern "C"
