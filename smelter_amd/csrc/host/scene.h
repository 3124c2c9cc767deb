// scene.h — host-side scene engine: Component tree -> stateful tree (transitions) -> NestedLayout -> flattened
// RenderLayout list (smr_layout[]) per frame.  C++ restatement of smelter-render/src/scene/* and
// transformations/layout/flatten.rs; all layout arithmetic is f32 in the reference's order,
// transition timing is f64 like the reference.  No GPU code here.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "json.h"
#include "smr.h"

namespace smr_host {

// Rust's `as usize` / `as u32` / `as i32` on a float — saturating, NaN -> 0: what every `Resolution { width: x as usize, .. }` and every
// ShaderParam conversion of the reference does.  (A plain C++ cast of an out-of-range float is undefined behaviour.)
inline size_t as_usize(double v) { return !(v > 0.0) ? 0 : v >= 18446744073709551615.0 ? SIZE_MAX : (size_t)v; }
inline uint32_t as_u32(double v) { return !(v > 0.0) ? 0u : v >= 4294967295.0 ? UINT32_MAX : (uint32_t)v; }
inline int32_t as_i32(double v) { return v != v ? 0 : v <= -2147483648.0 ? INT32_MIN : v >= 2147483647.0 ? INT32_MAX : (int32_t)v; }

struct RGBA { uint8_t r = 0, g = 0, b = 0, a = 0; bool operator==(const RGBA &o) const { return r == o.r && g == o.g && b == o.b && a == o.a; } };
struct Size { float width = 0, height = 0; };
struct BorderRadius {
    float tl = 0, tr = 0, br = 0, bl = 0;
    bool operator==(const BorderRadius &o) const { return tl == o.tl && tr == o.tr && br == o.br && bl == o.bl; }
};
struct BoxShadow {
    float offset_x = 0, offset_y = 0, blur_radius = 0;
    RGBA color;
    bool operator==(const BoxShadow &o) const { return offset_x == o.offset_x && offset_y == o.offset_y && blur_radius == o.blur_radius && color == o.color; }
};
struct Padding {
    float top = 0, right = 0, bottom = 0, left = 0;
    float horizontal() const { return left + right; }
    float vertical() const { return top + bottom; }
    bool operator==(const Padding &o) const { return top == o.top && right == o.right && bottom == o.bottom && left == o.left; }
};
using OptF = std::optional<float>;

// scene/types.rs:62-84
struct Position {
    bool absolute = false;
    OptF width, height;
    // absolute only
    bool from_bottom = false, from_right = false;  // VerticalPosition::BottomOffset / HorizontalPosition::RightOffset
    float v_offset = 0, h_offset = 0;
    float rotation_degrees = 0;
    bool operator==(const Position &o) const;
    Position with_border(float bw) const;
    Position with_padding(const Padding &p) const;
};

enum class InterpKind { Linear, Bounce, CubicBezier };
struct Interpolation { InterpKind kind = InterpKind::Linear; double x1 = 0, y1 = 0, x2 = 0, y2 = 0; };
struct Transition { int64_t duration_ns = 0; Interpolation interp; bool should_interrupt = false; };

double cubic_bezier_easing(double progress, double x1, double y1, double x2, double y2);  // scene/transition/cubic_bezier.rs
double bounce_easing(double t);                                                           // scene/transition/bounce.rs

// scene/transition.rs:17-101
struct TransitionState {
    double offset_progress = 0, offset_state = 0;
    int64_t start_pts_ns = 0, duration_ns = 0;
    Interpolation interp;
    double state(int64_t pts_ns) const;
    bool is_finished(int64_t pts_ns) const { return start_pts_ns + duration_ns <= pts_ns; }
    static std::optional<TransitionState> make(const std::optional<Transition> &current, const std::optional<TransitionState> &previous,
                                               bool props_changed, bool interrupt_previous, int64_t last_pts_ns);
};

enum class Kind { InputStream, View, Rescaler, Tiles, Text, Image, Shader };
enum class HAlign { Left, Right, Justified, Center };
enum class VAlign { Top, Center, Bottom, Justified };

struct ViewParam {
    std::string id; bool has_id = false;
    bool column = false;
    Position position;
    int overflow = 1;  // 0 visible, 1 hidden, 2 fit
    RGBA background_color;
    BorderRadius border_radius;
    float border_width = 0;
    RGBA border_color;
    std::vector<BoxShadow> box_shadow;
    Padding padding;
    bool operator==(const ViewParam &o) const;
};
struct RescalerParam {
    std::string id; bool has_id = false;
    Position position;
    bool fill = false;
    HAlign horizontal_align = HAlign::Center;
    VAlign vertical_align = VAlign::Center;
    BorderRadius border_radius;
    float border_width = 0;
    RGBA border_color;
    std::vector<BoxShadow> box_shadow;
    bool operator==(const RescalerParam &o) const;
};
struct TilesParam {
    std::string id; bool has_id = false;
    OptF width, height;
    RGBA background_color;
    uint32_t ar_w = 16, ar_h = 9;
    float margin = 0, padding = 0;
    HAlign horizontal_align = HAlign::Center;
    VAlign vertical_align = VAlign::Center;
    bool operator==(const TilesParam &o) const;
};
struct Tile {
    bool present = false;  // Option<Tile>
    bool id_is_component = false;
    std::string id_str; size_t id_index = 0;
    float top = 0, left = 0, width = 0, height = 0;
    bool same_id(const Tile &o) const { return id_is_component == o.id_is_component && (id_is_component ? id_str == o.id_str : id_index == o.id_index); }
};

struct NestedLayout;

// StatefulComponent (scene.rs) — one node of the stateful tree
struct Stateful {
    Kind kind = Kind::View;
    std::string id; bool has_id = false;
    // leaves
    std::string ref_id;          // input_id / image_id / shader_id
    Size leaf_size;              // InputStream: filled by update_state; Text/Image/Shader: intrinsic
    std::string text;            // Text payload (passed through to the caller)
    struct TextSpec {            // TextComponent as TextRendererCtx::layout_text / TextRendererNode take it (text_renderer.rs:174-233, 282-346)
        float font_size = 0, line_height = 0;
        RGBA color{255, 255, 255, 255}, background{0, 0, 0, 0};
        std::string family, style, weight, wrap, align;
    } text_spec;
    Json shader_param;
    Json desc;                   // the converted component (scene::Component) as canonical JSON, see Scene::parse
    // layouts
    ViewParam view_end; std::optional<ViewParam> view_start;
    RescalerParam resc_end; std::optional<RescalerParam> resc_start;
    TilesParam tiles;
    std::optional<std::pair<std::vector<Tile>, Size>> tiles_start, tiles_last_layout;
    std::optional<TransitionState> transition;
    std::vector<std::unique_ptr<Stateful>> children;

    bool is_layout() const { return kind == Kind::View || kind == Kind::Rescaler || kind == Kind::Tiles; }
    ViewParam view(int64_t pts) const;
    RescalerParam rescaler(int64_t pts) const;
    Position position(int64_t pts) const;
    OptF width(int64_t pts) const;
    OptF height(int64_t pts) const;
    NestedLayout layout(Size size, int64_t pts);
    void node_children(std::vector<Stateful *> &out);
    size_t node_children_count();
    void update_state(const std::vector<std::optional<Size>> &input_resolutions, size_t begin);
    std::unique_ptr<Stateful> clone() const;
};

// transformations/layout.rs:37-158
struct Crop { float top, left, width, height; };
struct MaskL { BorderRadius radius; float top, left, width, height; };
// The parent masks of one layout (at most SMR_MAX_MASKS reach the kernels; real scenes nest two or three deep).  The flatten step
// runs for every output and frame on the renderer thread, and with a heap vector here most of its time went to the allocator:
// the first few masks live inside the record.
class MaskList {
  public:
    static constexpr size_t INLINE = 2;
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    MaskL *begin() { return data(); }
    MaskL *end() { return data() + n_; }
    const MaskL *begin() const { return data(); }
    const MaskL *end() const { return data() + n_; }
    MaskL &operator[](size_t i) { return data()[i]; }
    const MaskL &operator[](size_t i) const { return data()[i]; }
    void clear() { n_ = 0; heap_.clear(); }
    void reserve(size_t) {}
    void push_back(const MaskL &m) {
        if (n_ < INLINE) { inl_[n_++] = m; return; }
        if (n_ == INLINE) heap_.assign(inl_, inl_ + INLINE);
        heap_.push_back(m);
        n_++;
    }
    void resize(size_t k) {  // (shrinks only)
        if (k >= n_) return;
        if (n_ > INLINE && k <= INLINE) {
            for (size_t i = 0; i < k; i++) inl_[i] = heap_[i];
            heap_.clear();
        } else if (n_ > INLINE) {
            heap_.resize(k);
        }
        n_ = k;
    }

  private:
    MaskL *data() { return n_ > INLINE ? heap_.data() : inl_; }
    const MaskL *data() const { return n_ > INLINE ? heap_.data() : inl_; }
    MaskL inl_[INLINE];
    size_t n_ = 0;
    std::vector<MaskL> heap_;  // all the masks once there are more than INLINE
};
struct RenderLayout {
    float top, left, width, height, rotation_degrees;
    BorderRadius border_radius;
    MaskList masks;
    int content = 1;  // 0 ChildNode, 1 Color, 2 BoxShadow
    RGBA color, border_color;
    float border_width = 0;
    size_t index = 0;
    Crop crop{0, 0, 0, 0};
    float blur_radius = 0;
};
struct NestedLayout {
    float top = 0, left = 0, width = 0, height = 0, rotation_degrees = 0, scale_x = 1, scale_y = 1;
    std::optional<Crop> crop;
    std::optional<MaskL> mask;
    int content = 2;  // 0 ChildNode, 1 Color, 2 None
    RGBA content_color;
    size_t content_index = 0;
    Size content_size;
    float border_width = 0;
    RGBA border_color;
    BorderRadius border_radius;
    std::vector<BoxShadow> box_shadow;
    std::vector<NestedLayout> children;
    size_t child_nodes_count = 0;
    std::vector<RenderLayout> flatten(const std::vector<std::optional<Size>> &input_resolutions, uint32_t out_w, uint32_t out_h) const;
};

// Render-graph node (scene_state.rs IntermediateNode / Node): layout and shader nodes are internal, the rest leaves.
struct GraphNode {
    Kind kind;               // View/Rescaler/Tiles => layout node; Shader; InputStream; Text; Image
    Stateful *component;     // layout root / leaf component (owned by the scene tree)
    int parent = -1;
    std::vector<int> children;
    Size forced_size; bool has_forced_size = false;
    // node_layouts' result while nothing under the node is in transition: the list does not depend on pts then, and a scene at rest
    // is the common case (the layout + flatten maths is ~8 us of the renderer thread per frame, in front of the first launch)
    struct LayoutCache {
        bool valid = false, srgb = false;
        int64_t pts_ns = 0;
        uint32_t w = 0, h = 0;
        std::vector<std::optional<Size>> resolutions;
        std::vector<smr_layout> layouts;
    } cache;
};

class Scene {
  public:
    // Renderer::update_scene (state.rs:177-189, scene/scene_state.rs:74-127)
    bool update(const std::string &json, uint32_t out_w, uint32_t out_h, std::string &err);
    // The smelter-api conversion alone (TryFrom<Component> for scene::Component, smelter-api/src/video/component_into.rs):
    // validates and returns the converted component tree as canonical JSON; the active scene is untouched.
    bool parse(const std::string &json, std::string &out, std::string &err) const;
    void register_image(const std::string &image_id, float w, float h) { images_[image_id] = Size{w, h}; }
    // the caller's text shaper (include/smr.h smr_text_measure_fn): sizes Text nodes without explicit width / height
    void set_text_measurer(smr_text_measure_fn fn, void *user) { measure_ = fn; measure_user_ = user; }
    const std::vector<GraphNode> &nodes() const { return nodes_; }
    // LayoutProvider::layouts + NestedLayout::flatten for one layout node (transformations/layout.rs:176-184);
    // also advances the render clock (SceneState::register_render_event).
    bool node_layouts(int node, int64_t pts_ns, const std::vector<std::optional<Size>> &child_resolutions, bool srgb,
                      std::vector<smr_layout> &out, uint32_t &w, uint32_t &h, std::string &err);
    uint32_t out_w() const { return out_w_; }
    uint32_t out_h() const { return out_h_; }

  private:
    std::unique_ptr<Stateful> root_;
    std::vector<GraphNode> nodes_;
    std::map<std::string, Size> images_;
    smr_text_measure_fn measure_ = nullptr;
    void *measure_user_ = nullptr;
    std::map<std::string, Size> input_resolutions_;  // SceneState::input_resolutions — from the last render
    int64_t last_pts_ns_ = 0;
    uint32_t out_w_ = 0, out_h_ = 0;
};

void convert_to_shader_color(RGBA c, bool srgb, float out[4]);  // wgpu/utils.rs:51-81
bool parse_color(const std::string &s, RGBA &out, std::string &err);  // smelter-api/src/video/color.rs

}  // namespace smr_host
