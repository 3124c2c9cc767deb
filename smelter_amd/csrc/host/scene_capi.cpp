// scene_capi.cpp — extern "C" surface of the host scene engine (include/smr.h, "a5/a6").
#include <cstring>

#include "scene.h"

using namespace smr_host;

struct smr_scene {
    Scene scene;
    std::string err;
    std::string parsed;  // storage behind smr_scene_parse's out_json
};

static int set_err(smr_scene *s, const std::string &msg) {
    if (s) s->err = msg;
    return -1;  // SMR_ERR_VALIDATION
}

extern "C" {

SMR_API int smr_scene_create(smr_scene **out) {
    if (!out) return -1;
    *out = new smr_scene();
    return 0;
}
SMR_API void smr_scene_destroy(smr_scene *scene) { delete scene; }
SMR_API const char *smr_scene_last_error(const smr_scene *scene) { return scene ? scene->err.c_str() : "null scene"; }

SMR_API int smr_scene_register_image(smr_scene *scene, const char *image_id, uint32_t width, uint32_t height) {
    if (!scene || !image_id) return set_err(scene, "smr_scene_register_image: null argument");
    scene->scene.register_image(image_id, (float)width, (float)height);
    return 0;
}

SMR_API int smr_scene_set_text_measurer(smr_scene *scene, smr_text_measure_fn fn, void *user) {
    if (!scene) return -1;
    scene->scene.set_text_measurer(fn, user);
    return 0;
}

SMR_API int smr_scene_update(smr_scene *scene, const char *scene_json, uint32_t out_width, uint32_t out_height) {
    if (!scene || !scene_json) return set_err(scene, "smr_scene_update: null argument");
    std::string err;
    if (!scene->scene.update(scene_json, out_width, out_height, err)) return set_err(scene, err);
    return 0;
}

SMR_API int smr_scene_parse(smr_scene *scene, const char *scene_json, const char **out_json) {
    if (!scene || !scene_json || !out_json) return set_err(scene, "smr_scene_parse: null argument");
    std::string err;
    if (!scene->scene.parse(scene_json, scene->parsed, err)) return set_err(scene, err);
    *out_json = scene->parsed.c_str();
    return 0;
}

SMR_API int smr_scene_node_count(const smr_scene *scene) { return scene ? (int)scene->scene.nodes().size() : -1; }

SMR_API int smr_scene_node_info(const smr_scene *scene_c, int node, smr_scene_node *out) {
    smr_scene *scene = const_cast<smr_scene *>(scene_c);
    if (!scene || !out) return set_err(scene, "smr_scene_node_info: null argument");
    const auto &nodes = scene->scene.nodes();
    if (node < 0 || node >= (int)nodes.size()) return set_err(scene, "smr_scene_node_info: node index out of range");
    const GraphNode &g = nodes[node];
    const Stateful &c = *g.component;
    memset(out, 0, sizeof(*out));
    switch (g.kind) {
    case Kind::InputStream: out->kind = SMR_NODE_INPUT_STREAM; break;
    case Kind::Text: out->kind = SMR_NODE_TEXT; break;
    case Kind::Image: out->kind = SMR_NODE_IMAGE; break;
    case Kind::Shader: out->kind = SMR_NODE_SHADER; break;
    default: out->kind = SMR_NODE_LAYOUT; break;
    }
    out->parent = g.parent;
    out->n_children = (uint32_t)g.children.size();
    Size sz = g.has_forced_size ? g.forced_size : c.leaf_size;
    out->width = as_u32(sz.width);
    out->height = as_u32(sz.height);
    out->ref_id = c.ref_id.c_str();  // owned by the component tree: valid until the next successful update
    out->id = c.id.c_str();
    out->payload = c.kind == Kind::Text ? c.text.c_str() : "";
    return 0;
}

SMR_API int smr_scene_node_children(const smr_scene *scene_c, int node, int32_t *out, uint32_t cap) {
    smr_scene *scene = const_cast<smr_scene *>(scene_c);
    if (!scene) return -1;
    const auto &nodes = scene->scene.nodes();
    if (node < 0 || node >= (int)nodes.size()) return set_err(scene, "smr_scene_node_children: node index out of range");
    const auto &ch = nodes[node].children;
    for (size_t i = 0; i < ch.size() && i < cap && out; i++) out[i] = ch[i];
    return (int)ch.size();
}

SMR_API int smr_scene_node_layouts(smr_scene *scene, int node, int64_t pts_ns, const uint32_t *child_wh, uint32_t n_children, uint32_t mode,
                                   smr_layout *out, uint32_t cap, uint32_t *n_out, uint32_t *out_width, uint32_t *out_height) {
    if (!scene || !n_out || (n_children && !child_wh) || (cap && !out)) return set_err(scene, "smr_scene_node_layouts: null argument");
    std::vector<std::optional<Size>> res(n_children);
    for (uint32_t i = 0; i < n_children; i++)
        if (child_wh[2 * i] != SMR_NO_RESOLUTION) res[i] = Size{(float)child_wh[2 * i], (float)child_wh[2 * i + 1]};
    std::vector<smr_layout> layouts;
    uint32_t w = 0, h = 0;
    std::string err;
    if (!scene->scene.node_layouts(node, pts_ns, res, mode == SMR_MODE_GPU_OPTIMIZED, layouts, w, h, err)) return set_err(scene, err);
    *n_out = (uint32_t)layouts.size();
    for (size_t i = 0; i < layouts.size() && i < cap; i++) out[i] = layouts[i];
    if (out_width) *out_width = w;
    if (out_height) *out_height = h;
    return 0;
}

SMR_API double smr_cubic_bezier_easing(double progress, double x1, double y1, double x2, double y2) {
    return cubic_bezier_easing(progress, x1, y1, x2, y2);
}
SMR_API double smr_bounce_easing(double progress) { return bounce_easing(progress); }
SMR_API int smr_parse_color(const char *text, uint8_t rgba[4]) {
    if (!text || !rgba) return -1;
    RGBA c;
    std::string err;
    if (!parse_color(text, c, err)) return -1;
    rgba[0] = c.r; rgba[1] = c.g; rgba[2] = c.b; rgba[3] = c.a;
    return 0;
}

}  // extern "C"
