// scene.cpp — see scene.h.  Every function cites the reference source it restates (paths relative to smelter-render/src
// unless noted).  f32 arithmetic is written operation by operation in the reference's order (this file is compiled with
// -ffp-contract=off), so the flattened layout list is bit-identical to oracle/scene.py.
#include "scene.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <deque>

namespace smr_host {

// ------------------------------------------------------------------------------------------------ small helpers
// Rust's f32::min / f32::max return the other operand when one is NaN (IEEE minNum / maxNum) and f32::clamp keeps a NaN;
// std::min / std::max do neither, and empty inputs (0x0 -> 0/0) do reach this arithmetic.
static inline float rmin(float a, float b) { return fminf(a, b); }
static inline float rmax(float a, float b) { return fmaxf(a, b); }
static inline double rmin(double a, double b) { return fmin(a, b); }
static inline double rmax(double a, double b) { return fmax(a, b); }
static inline size_t rmin(size_t a, size_t b) { return a < b ? a : b; }
static inline double rclamp(double x, double lo, double hi) { return x < lo ? lo : x > hi ? hi : x; }
static inline float rclamp(float x, float lo, float hi) { return x < lo ? lo : x > hi ? hi : x; }

static double secs(int64_t ns) {  // Duration::as_secs_f64
    return (double)(ns / 1000000000LL) + (double)(ns % 1000000000LL) / 1e9;
}
static float interp_f32(float a, float b, double s) { return (float)((double)a + (((double)b - (double)a) * s)); }  // types/interpolation.rs:13-36
static OptF interp_opt(const OptF &a, const OptF &b, double s) {
    if (a && b) return interp_f32(*a, *b, s);
    return b;
}
static BorderRadius interp(const BorderRadius &a, const BorderRadius &b, double s) {
    return {interp_f32(a.tl, b.tl, s), interp_f32(a.tr, b.tr, s), interp_f32(a.br, b.br, s), interp_f32(a.bl, b.bl, s)};
}
static Padding interp(const Padding &a, const Padding &b, double s) {
    return {interp_f32(a.top, b.top, s), interp_f32(a.right, b.right, s), interp_f32(a.bottom, b.bottom, s), interp_f32(a.left, b.left, s)};
}
static std::vector<BoxShadow> interp(const std::vector<BoxShadow> &a, const std::vector<BoxShadow> &b, double s) {
    // components/interpolation.rs:66-91
    std::vector<BoxShadow> out;
    size_t n = rmin(a.size(), b.size());
    for (size_t i = 0; i < n; i++)
        out.push_back({interp_f32(a[i].offset_x, b[i].offset_x, s), interp_f32(a[i].offset_y, b[i].offset_y, s),
                       interp_f32(a[i].blur_radius, b[i].blur_radius, s), b[i].color});
    for (size_t i = n; i < b.size(); i++) out.push_back(b[i]);
    return out;
}
static Position interp(const Position &a, const Position &b, double s) {
    // components/interpolation.rs:8-64, types/interpolation.rs:62-90
    if (!a.absolute && !b.absolute) {
        Position p = b;
        p.width = interp_opt(a.width, b.width, s);
        p.height = interp_opt(a.height, b.height, s);
        return p;
    }
    if (a.absolute && b.absolute) {
        Position p = b;
        p.width = interp_opt(a.width, b.width, s);
        p.height = interp_opt(a.height, b.height, s);
        if (a.from_right == b.from_right) p.h_offset = interp_f32(a.h_offset, b.h_offset, s);
        if (a.from_bottom == b.from_bottom) p.v_offset = interp_f32(a.v_offset, b.v_offset, s);
        p.rotation_degrees = interp_f32(a.rotation_degrees, b.rotation_degrees, s);
        return p;
    }
    return b;
}

bool Position::operator==(const Position &o) const {
    if (absolute != o.absolute || width != o.width || height != o.height) return false;
    if (!absolute) return true;
    return from_bottom == o.from_bottom && from_right == o.from_right && v_offset == o.v_offset && h_offset == o.h_offset &&
           rotation_degrees == o.rotation_degrees;
}
Position Position::with_border(float bw) const {  // components/position.rs:5-28
    Position p = *this;
    if (p.width) p.width = *p.width + 2.0f * bw;
    if (p.height) p.height = *p.height + 2.0f * bw;
    return p;
}
Position Position::with_padding(const Padding &pd) const {  // components/position.rs:30-53
    Position p = *this;
    if (p.width) p.width = *p.width + pd.horizontal();
    if (p.height) p.height = *p.height + pd.vertical();
    return p;
}
bool ViewParam::operator==(const ViewParam &o) const {
    return has_id == o.has_id && id == o.id && column == o.column && position == o.position && overflow == o.overflow &&
           background_color == o.background_color && border_radius == o.border_radius && border_width == o.border_width &&
           border_color == o.border_color && box_shadow == o.box_shadow && padding == o.padding;
}
bool RescalerParam::operator==(const RescalerParam &o) const {
    return has_id == o.has_id && id == o.id && position == o.position && fill == o.fill && horizontal_align == o.horizontal_align &&
           vertical_align == o.vertical_align && border_radius == o.border_radius && border_width == o.border_width &&
           border_color == o.border_color && box_shadow == o.box_shadow;
}
bool TilesParam::operator==(const TilesParam &o) const {
    return has_id == o.has_id && id == o.id && width == o.width && height == o.height && background_color == o.background_color &&
           ar_w == o.ar_w && ar_h == o.ar_h && margin == o.margin && padding == o.padding && horizontal_align == o.horizontal_align &&
           vertical_align == o.vertical_align;
}

// ------------------------------------------------------------------------------------------------ easing / transitions
double bounce_easing(double t) {  // scene/transition/bounce.rs:1-14
    const double n1 = 7.5625, d1 = 2.75;
    if (t < (1.0 / d1)) return n1 * t * t;
    if (t < (2.0 / d1)) return n1 * (t - 1.5 / d1) * (t - 1.5 / d1) + 0.75;
    if (t < (2.5 / d1)) return n1 * (t - 2.25 / d1) * (t - 2.25 / d1) + 0.9375;
    return n1 * (t - 2.625 / d1) * (t - 2.625 / d1) + 0.984375;
}

namespace {
const double EPS = 1e-7;  // ALLOWED_FLOATING_ERROR
bool close_to(double a, double b) { return std::fabs(a - b) < EPS; }
double clamp_root(double v) {  // clamp_valid_root_in_unit_range
    if (v < 0.0) return v >= -EPS ? 0.0 : NAN;
    if (v > 1.0) return v <= 1.0 + EPS ? 1.0 : NAN;
    return v;
}
double cubic_bezier(double t, double p1, double p2) {
    double a = 1.0 / 3.0 + (p1 - p2), b = p2 - 2.0 * p1, c = p1;
    return 3.0 * ((a * t + b) * t + c) * t;
}
double find_first_cubic_root(double p0, double p1, double p2, double p3) {  // cubic_bezier.rs:33-104
    double a = 3.0 * (p0 - 2.0 * p1 + p2), b = 3.0 * (p1 - p0), c = p0, d = -p0 + 3.0 * (p1 - p2) + p3;
    if (close_to(d, 0.0)) {
        if (close_to(a, 0.0)) {
            if (close_to(b, 0.0)) return NAN;
            return clamp_root(-c / b);
        }
        double q = std::sqrt(b * b - 4.0 * a * c), a2 = 2.0 * a;
        double root = clamp_root((q - b) / a2);
        if (!std::isnan(root)) return root;
        return clamp_root((-b - q) / a2);
    }
    a = a / d; b = b / d; c = c / d;
    double o3 = (3.0 * b - a * a) / 9.0;
    double q2 = (2.0 * (a * a * a) - 9.0 * a * b + 27.0 * c) / 54.0;
    double a3 = a / 3.0;
    double disc = q2 * q2 + o3 * o3 * o3;
    if (disc < 0.0) {
        double mp33 = -(o3 * o3 * o3), r = std::sqrt(mp33);
        double cos_phi = rclamp(-q2 / r, -1.0, 1.0), phi = std::acos(cos_phi), t1 = 2.0 * std::cbrt(r);
        double root = clamp_root(t1 * std::cos(phi / 3.0) - a3);
        if (!std::isnan(root)) return root;
        root = clamp_root(t1 * std::cos((phi + 2.0 * M_PI) / 3.0) - a3);
        if (!std::isnan(root)) return root;
        return clamp_root(t1 * std::cos((phi + 4.0 * M_PI) / 3.0) - a3);
    }
    if (disc == 0.0) {
        double u1 = -std::cbrt(q2);
        double root = clamp_root(2.0 * u1 - a3);
        if (!std::isnan(root)) return root;
        return clamp_root(-u1 - a3);
    }
    double sd = std::sqrt(disc), u1 = std::cbrt(-q2 + sd), v1 = std::cbrt(q2 + sd);
    return clamp_root(u1 - v1 - a3);
}
double interp_state(const Interpolation &k, double t) {  // transition.rs:107-117
    switch (k.kind) {
    case InterpKind::Linear: return t;
    case InterpKind::Bounce: return bounce_easing(t);
    default: return cubic_bezier_easing(t, k.x1, k.y1, k.x2, k.y2);
    }
}
}  // namespace

double cubic_bezier_easing(double progress, double x1, double y1, double x2, double y2) {  // cubic_bezier.rs:5-23
    if (close_to(progress, 0.0)) return 0.0;
    if (close_to(progress, 1.0)) return 1.0;
    double t = find_first_cubic_root(-progress, x1 - progress, x2 - progress, 1.0 - progress);
    if (std::isnan(t)) return 1.0;
    return rclamp(cubic_bezier(t, y1, y2), 0.0, 1.0);
}

double TransitionState::state(int64_t pts_ns) const {  // transition.rs:88-101
    double progress = (secs(pts_ns) - secs(start_pts_ns)) / secs(duration_ns);
    progress = offset_progress + progress * (1.0 - offset_progress);
    progress = rclamp(progress, 0.0, 1.0);  // a zero-length transition at its own start pts yields NaN here, as in the reference
    double st = interp_state(interp, progress);
    return (st - offset_state) / (1.0 - offset_state);
}

std::optional<TransitionState> TransitionState::make(const std::optional<Transition> &current, const std::optional<TransitionState> &previous,
                                                     bool props_changed, bool interrupt_previous, int64_t last_pts) {
    // transition.rs:39-86
    auto from_options = [&](const Transition &t) {
        TransitionState s;
        s.start_pts_ns = last_pts;
        s.duration_ns = t.duration_ns;
        s.interp = t.interp;
        return s;
    };
    if (previous && !previous->is_finished(last_pts)) {
        if (props_changed && interrupt_previous) {
            if (current) return from_options(*current);
            return std::nullopt;
        }
        int64_t remaining = std::max<int64_t>(0, previous->start_pts_ns + previous->duration_ns - last_pts);
        double progress_offset = 1.0 - (secs(remaining) / secs(previous->duration_ns));
        TransitionState s;
        s.offset_progress = progress_offset;
        s.offset_state = interp_state(previous->interp, progress_offset);
        s.start_pts_ns = last_pts;
        s.duration_ns = remaining;
        s.interp = current ? current->interp : previous->interp;
        return s;
    }
    if (props_changed && current) return from_options(*current);
    return std::nullopt;
}

// ------------------------------------------------------------------------------------------------ stateful components
ViewParam Stateful::view(int64_t pts) const {  // view_component.rs:49-55, view_component/interpolation.rs
    if (!transition || !view_start) return view_end;
    double s = transition->state(pts);
    ViewParam v = view_end;
    v.position = interp(view_start->position, view_end.position, s);
    v.border_radius = interp(view_start->border_radius, view_end.border_radius, s);
    v.border_width = interp_f32(view_start->border_width, view_end.border_width, s);
    v.box_shadow = interp(view_start->box_shadow, view_end.box_shadow, s);
    v.padding = interp(view_start->padding, view_end.padding, s);
    return v;
}
RescalerParam Stateful::rescaler(int64_t pts) const {  // rescaler_component.rs:57-63, rescaler_component/interpolation.rs
    if (!transition || !resc_start) return resc_end;
    double s = transition->state(pts);
    RescalerParam r = resc_end;
    r.position = interp(resc_start->position, resc_end.position, s);
    r.border_radius = interp(resc_start->border_radius, resc_end.border_radius, s);
    r.border_width = interp_f32(resc_start->border_width, resc_end.border_width, s);
    r.box_shadow = interp(resc_start->box_shadow, resc_end.box_shadow, s);
    return r;
}
Position Stateful::position(int64_t pts) const {
    switch (kind) {
    case Kind::View: { ViewParam v = view(pts); return v.position.with_border(v.border_width).with_padding(v.padding); }  // view_component.rs:66-71
    case Kind::Rescaler: { RescalerParam r = rescaler(pts); return r.position.with_border(r.border_width); }              // rescaler_component.rs:74-77
    case Kind::Tiles: { Position p; p.width = tiles.width; p.height = tiles.height; return p; }                            // tiles_component.rs:77-82
    default: { Position p; p.width = leaf_size.width; p.height = leaf_size.height; return p; }
    }
}
OptF Stateful::width(int64_t pts) const { return position(pts).width; }    // scene.rs StatefulComponent::width
OptF Stateful::height(int64_t pts) const { return position(pts).height; }

void Stateful::node_children(std::vector<Stateful *> &out) {  // scene/layout.rs:92-101
    for (auto &c : children) {
        if (c->is_layout()) c->node_children(out);
        else out.push_back(c.get());
    }
}
size_t Stateful::node_children_count() {  // (counted, not collected: this runs for every layout component of every frame)
    size_t n = 0;
    for (auto &c : children) n += c->is_layout() ? c->node_children_count() : 1;
    return n;
}
void Stateful::update_state(const std::vector<std::optional<Size>> &res, size_t begin) {  // scene/layout.rs:103-137
    size_t off = begin;
    for (auto &c : children) {
        if (c->kind == Kind::InputStream) {
            c->leaf_size = (off < res.size() && res[off]) ? *res[off] : Size{0.0f, 0.0f};
            off += 1;
        } else if (c->is_layout()) {
            size_t k = c->node_children_count();
            c->update_state(res, off);
            off += k;
        } else {
            off += 1;
        }
    }
}
std::unique_ptr<Stateful> Stateful::clone() const {
    auto c = std::make_unique<Stateful>();
    c->kind = kind; c->id = id; c->has_id = has_id; c->ref_id = ref_id; c->leaf_size = leaf_size; c->text = text; c->text_spec = text_spec;
    c->shader_param = shader_param; c->view_end = view_end; c->view_start = view_start; c->resc_end = resc_end;
    c->resc_start = resc_start; c->tiles = tiles; c->tiles_start = tiles_start; c->tiles_last_layout = tiles_last_layout;
    c->transition = transition;
    for (auto &ch : children) c->children.push_back(ch->clone());
    return c;
}

// ------------------------------------------------------------------------------------------------ layouts
static BorderRadius clip_to_size(const BorderRadius &r, Size s) {  // scene/types.rs:109-117
    float mx = rmax(0.0f, rmin(s.width, s.height) / 2.0f);
    auto cl = [&](float v) { return rclamp(v, 0.0f, mx); };
    return {cl(r.tl), cl(r.tr), cl(r.br), cl(r.bl)};
}
static BorderRadius radius_add(const BorderRadius &r, float d) {  // scene/types.rs:143-160 (Add / Sub clamp at 0)
    return {rmax(r.tl + d, 0.0f), rmax(r.tr + d, 0.0f), rmax(r.br + d, 0.0f), rmax(r.bl + d, 0.0f)};
}
static BorderRadius radius_mul(const BorderRadius &r, float k) { return {r.tl * k, r.tr * k, r.br * k, r.bl * k}; }

static void set_layout_content(NestedLayout &n, const Stateful &c, size_t index) {  // scene/layout.rs:139-162
    if (c.is_layout()) { n.content = 2; return; }
    n.content = 0;
    n.content_index = index;
    n.content_size = c.leaf_size;
}
static NestedLayout placeholder(size_t count) {  // transformations/layout.rs:280-304
    NestedLayout n;
    n.child_nodes_count = count;
    return n;
}
static NestedLayout wrap_layout_child(Stateful &ch, float top, float left, float w, float h, float rot, int64_t pts) {
    NestedLayout inner = ch.layout(Size{w, h}, pts);
    NestedLayout n;
    n.top = top; n.left = left; n.width = w; n.height = h; n.rotation_degrees = rot;
    n.child_nodes_count = inner.child_nodes_count;
    n.children.push_back(std::move(inner));
    return n;
}
static NestedLayout absolute_child(Stateful &ch, const Position &pos, Size parent, int64_t pts) {  // scene/layout.rs:164-239
    float w = pos.width ? *pos.width : parent.width;
    float h = pos.height ? *pos.height : parent.height;
    float top = pos.from_bottom ? parent.height - pos.v_offset - h : pos.v_offset;
    float left = pos.from_right ? parent.width - pos.h_offset - w : pos.h_offset;
    if (ch.is_layout()) return wrap_layout_child(ch, top, left, w, h, pos.rotation_degrees, pts);
    NestedLayout n;
    n.top = top; n.left = left; n.width = w; n.height = h; n.rotation_degrees = pos.rotation_degrees;
    set_layout_content(n, ch, 0);
    n.child_nodes_count = 1;
    return n;
}

static NestedLayout view_layout(const ViewParam &v, Size size, std::vector<std::unique_ptr<Stateful>> &children, int64_t pts) {
    // view_component/layout.rs:31-285
    const float bw = v.border_width;
    Size content{rmax(size.width - 2.0f * bw, 0.0f), rmax(size.height - 2.0f * bw, 0.0f)};
    BorderRadius radius = clip_to_size(v.border_radius, size);
    auto is_static = [&](Stateful &c) { return !(c.is_layout() && c.position(pts).absolute); };
    auto main_size = [&](Stateful &c) { return v.column ? c.height(pts) : c.width(pts); };
    auto sum_static = [&]() {
        float acc = 0.0f;
        for (auto &c : children)
            if (is_static(*c)) { OptF m = main_size(*c); acc = acc + (m ? *m : 0.0f); }
        return acc;
    };
    // static_child_size (:205-231)
    float max_size = v.column ? content.height - v.padding.vertical() : content.width - v.padding.horizontal();
    size_t unknown = 0;
    for (auto &c : children)
        if (is_static(*c) && !main_size(*c)) unknown++;
    float static_child_size = unknown == 0 ? 0.0f : rmax(0.0f, (max_size - sum_static()) / (float)unknown);
    std::optional<MaskL> mask;
    float scale = 1.0f;
    if (v.overflow != 0) mask = MaskL{radius_add(radius, -bw), bw, bw, content.width, content.height};
    if (v.overflow == 2) {  // scale_factor_for_overflow_fit (:233-262)
        float sum_size = rmax(sum_static(), 0.000000001f);
        float mx = v.column ? content.height : content.width, alt = v.column ? content.width : content.height;
        float max_alt = 0.0f; bool any = false;
        for (auto &c : children) {
            if (!is_static(*c)) continue;
            OptF a = v.column ? c->width(pts) : c->height(pts);
            float av = a ? *a : 0.0f;
            if (!any || av > max_alt) max_alt = av;
            any = true;
        }
        max_alt = rmax(any ? max_alt : 0.0f, 0.000000001f);
        scale = rmin(1.0f, rmin(mx / sum_size, alt / max_alt));
    }
    float static_offset = bw / scale;
    const float parent_bw = bw / scale;
    NestedLayout out;
    out.children.reserve(children.size());
    for (auto &cp : children) {
        Stateful &c = *cp;
        Position pos = c.position(pts);
        if (c.is_layout() && pos.absolute) { out.children.push_back(absolute_child(c, pos, size, pts)); continue; }
        OptF cw = c.width(pts), chh = c.height(pts);
        float top, left, w, h;
        if (!v.column) {
            w = cw ? *cw : static_child_size;
            h = chh ? *chh : content.height - v.padding.vertical();
            top = parent_bw + v.padding.top;
            left = static_offset + v.padding.left;
            static_offset += w;
        } else {
            h = chh ? *chh : static_child_size;
            w = cw ? *cw : content.width - v.padding.horizontal();
            top = static_offset + v.padding.top;
            left = parent_bw + v.padding.left;
            static_offset += h;
        }
        if (c.is_layout()) {
            out.children.push_back(wrap_layout_child(c, top, left, w, h, 0.0f, pts));
        } else {
            NestedLayout n;
            n.top = top; n.left = left; n.width = w; n.height = h;
            set_layout_content(n, c, 0);
            n.child_nodes_count = 1;
            out.children.push_back(std::move(n));
        }
    }
    out.width = size.width; out.height = size.height;
    out.scale_x = scale; out.scale_y = scale;
    out.mask = mask;
    out.content = 1; out.content_color = v.background_color;
    for (auto &k : out.children) out.child_nodes_count += k.child_nodes_count;
    out.border_width = bw; out.border_color = v.border_color; out.border_radius = radius; out.box_shadow = v.box_shadow;
    return out;
}

static NestedLayout rescaler_layout(const RescalerParam &r, Size size, Stateful &child, int64_t pts) {
    // rescaler_component/layout.rs:14-162
    const float bw = r.border_width;
    Size content{rmax(size.width - (2.0f * bw), 0.0f), rmax(size.height - (2.0f * bw), 0.0f)};
    OptF kw = child.width(pts), kh = child.height(pts);
    BorderRadius radius = clip_to_size(r.border_radius, size);
    float scale;
    if (!kw && !kh) scale = 1.0f;
    else if (!kw) scale = content.height / *kh;
    else if (!kh) scale = content.width / *kw;
    else scale = r.fill ? rmax(content.width / *kw, content.height / *kh) : rmin(content.width / *kw, content.height / *kh);
    NestedLayout inner;
    if (child.is_layout()) {
        NestedLayout cl = child.layout(Size{kw ? *kw : content.width / scale, kh ? *kh : content.height / scale}, pts);
        inner.content = 2;
        inner.child_nodes_count = cl.child_nodes_count;
        inner.children.push_back(std::move(cl));
    } else {
        set_layout_content(inner, child, 0);
        inner.child_nodes_count = 1;
    }
    float top = 0.0f, left = 0.0f;
    if (kh) {
        if (r.vertical_align == VAlign::Bottom) top = content.height - (*kh * scale);
        else if (r.vertical_align != VAlign::Top) top = (content.height - (*kh * scale)) / 2.0f;
    }
    if (kw) {
        if (r.horizontal_align == HAlign::Right) left = content.width - (*kw * scale);
        else if (r.horizontal_align != HAlign::Left) left = (content.width - (*kw * scale)) / (2.0f);
    }
    inner.top = top + bw; inner.left = left + bw;
    inner.width = kw ? *kw * scale : content.width;
    inner.height = kh ? *kh * scale : content.height;
    inner.scale_x = scale; inner.scale_y = scale;
    NestedLayout out;
    out.width = content.width + (bw * 2.0f); out.height = content.height + (bw * 2.0f);
    out.mask = MaskL{radius_add(radius, -bw), bw, bw, content.width, content.height};
    out.content = 2;
    out.child_nodes_count = inner.child_nodes_count;
    out.children.push_back(std::move(inner));
    out.border_width = bw; out.border_color = r.border_color; out.border_radius = radius; out.box_shadow = r.box_shadow;
    return out;
}

// tiles_component/tiles.rs:29-166
static Size tile_size(const TilesParam &t, uint32_t rows, uint32_t cols, Size ls) {
    float x_padding = (float)cols * 2.0f * t.padding, y_padding = (float)rows * 2.0f * t.padding;
    float x_margin = ((float)cols + 1.0f) * t.margin, y_margin = ((float)rows + 1.0f) * t.margin;
    float xs = rmax(ls.width - x_padding - x_margin, 0.0f) / (float)cols / (float)t.ar_w;
    float ys = rmax(ls.height - y_padding - y_margin, 0.0f) / (float)rows / (float)t.ar_h;
    float s = xs < ys ? xs : ys;
    return {(float)t.ar_w * s, (float)t.ar_h * s};
}
static std::vector<Tile> tiles_end(const TilesParam &t, Size size, std::vector<std::unique_ptr<Stateful>> &children) {
    const uint32_t count = (uint32_t)children.size();
    std::vector<Tile> out;
    if (count == 0) return out;
    uint32_t best_rows = 1, best_cols = count;
    float best_w = 0.0f;
    for (uint32_t rows = 1; rows <= count; rows++) {
        uint32_t cols = (count + rows - 1) / rows;
        float w = tile_size(t, rows, cols, size).width;
        if (w > best_w) { best_rows = rows; best_cols = cols; best_w = w; }
    }
    const uint32_t rows = best_rows, cols = best_cols;
    Size ts = tile_size(t, rows, cols, size);
    float add_y = size.height - (ts.height + 2.0f * t.padding) * (float)rows - (t.margin * ((float)rows + 1.0f));
    float add_top = 0.0f, just_y = 0.0f;
    switch (t.vertical_align) {
    case VAlign::Top: break;
    case VAlign::Center: add_top = add_y / 2.0f; break;
    case VAlign::Bottom: add_top = add_y; break;
    case VAlign::Justified: just_y = add_y / ((float)rows + 1.0f); break;
    }
    float top = add_top + just_y + t.padding + t.margin;
    size_t unnamed = 0;
    for (uint32_t row = 0; row < rows; row++) {
        uint32_t in_row = row < rows - 1 ? cols : count - ((rows - 1) * cols);
        float add_x = size.width - (ts.width + 2.0f * t.padding) * (float)in_row - (t.margin * ((float)in_row + 1.0f));
        float add_left = 0.0f, just_x = 0.0f;
        switch (t.horizontal_align) {
        case HAlign::Left: break;
        case HAlign::Right: add_left = add_x; break;
        case HAlign::Justified: just_x = add_x / (float)(in_row + 1); break;
        case HAlign::Center: add_left = add_x / 2.0f; break;
        }
        float left = add_left + just_x + t.margin + t.padding;
        for (uint32_t c = 0; c < in_row; c++) {
            Tile tile;
            tile.present = true;
            tile.top = top; tile.left = left; tile.width = ts.width; tile.height = ts.height;
            Stateful &child = *children[out.size()];
            if (child.has_id) { tile.id_is_component = true; tile.id_str = child.id; }
            else { tile.id_is_component = false; tile.id_index = unnamed++; }
            out.push_back(tile);
            left += ts.width + t.margin + t.padding * 2.0f + just_x;
        }
        top += ts.height + t.margin + t.padding * 2.0f + just_y;
    }
    return out;
}
static std::vector<Tile> interp_tiles(const std::vector<Tile> &start, const std::vector<Tile> &end, double s) {
    // tiles_component/interpolation.rs:17-64
    if (s >= 1.0) return end;
    std::vector<Tile> out;
    for (const Tile &tile : end) {
        if (!tile.present) { out.push_back(Tile{}); continue; }
        const Tile *old = nullptr;
        for (const Tile &st : start)
            if (st.present && st.same_id(tile)) { old = &st; }  // HashMap: the last duplicate wins
        if (old) {
            Tile t = tile;
            t.top = interp_f32(old->top, tile.top, s); t.left = interp_f32(old->left, tile.left, s);
            t.width = interp_f32(old->width, tile.width, s); t.height = interp_f32(old->height, tile.height, s);
            out.push_back(t);
            continue;
        }
        const Tile *same_pos = nullptr;
        for (const Tile &st : start) {
            if (!st.present) continue;
            const float TOL = 0.001f;
            if (std::fabs(st.top - tile.top) <= TOL && std::fabs(st.left - tile.left) <= TOL && std::fabs(st.width - tile.width) <= TOL &&
                std::fabs(st.height - tile.height) <= TOL) { same_pos = &st; break; }
        }
        if (same_pos) {
            bool still_there = false;
            for (const Tile &e : end)
                if (e.present && e.same_id(*same_pos)) still_there = true;
            if (still_there) out.push_back(Tile{});  // hidden until the end of the transition
            else out.push_back(tile);
        } else {
            out.push_back(Tile{});  // .find(..) found nothing -> and_then yields None
        }
    }
    return out;
}

NestedLayout Stateful::layout(Size size, int64_t pts) {
    if (kind == Kind::View) return view_layout(view(pts), size, children, pts);
    if (kind == Kind::Rescaler) return rescaler_layout(rescaler(pts), size, *children[0], pts);
    // Tiles: tiles_component.rs:63-75,113-123, tiles_component/layout.rs:10-151
    std::vector<Tile> tl = tiles_end(tiles, size, children);
    if (tiles_start && transition) {
        const Size &ss = tiles_start->second;
        float sc = rmin(size.width / ss.width, size.height / ss.height);  // resize_tiles
        std::vector<Tile> st = tiles_start->first;
        for (Tile &t : st)
            if (t.present) { t.top *= sc; t.left *= sc; t.width *= sc; t.height *= sc; }
        tl = interp_tiles(st, tl, transition->state(pts));
    }
    NestedLayout out;
    out.children.reserve(children.size());
    for (size_t i = 0; i < children.size() && i < tl.size(); i++) {
        Stateful &ch = *children[i];
        const Tile &tile = tl[i];
        if (!tile.present) {
            out.children.push_back(placeholder(ch.is_layout() ? ch.node_children_count() : 1));
            continue;
        }
        if (ch.is_layout()) {
            out.children.push_back(wrap_layout_child(ch, tile.top, tile.left, tile.width, tile.height, 0.0f, pts));
        } else {
            float top = tile.top, left = tile.left, w = tile.width, h = tile.height;
            OptF kw = ch.width(pts), kh = ch.height(pts);
            if (kw && kh) {  // fit_into_tile
                float sf = rmin(tile.width / *kw, tile.height / *kh);
                float top_off = (tile.height - sf * *kh) / 2.0f, left_off = (tile.width - sf * *kw) / 2.0f;
                top = tile.top + top_off; left = tile.left + left_off; w = sf * *kw; h = sf * *kh;
            }
            NestedLayout n;
            n.top = top; n.left = left; n.width = w; n.height = h;
            set_layout_content(n, ch, 0);
            n.child_nodes_count = 1;
            out.children.push_back(std::move(n));
        }
    }
    out.width = size.width; out.height = size.height;
    out.content = 1; out.content_color = tiles.background_color;
    for (auto &k : out.children) out.child_nodes_count += k.child_nodes_count;
    tiles_last_layout = std::make_pair(tl, size);
    return out;
}

// ------------------------------------------------------------------------------------------------ flatten (layout/flatten.rs)
namespace {
// (this code runs once per frame and output on the renderer thread: lists are transformed in place and moved, not rebuilt)
void child_parent_masks(const NestedLayout &n, MaskList &masks) {  // :358-371
    float s = rmin(n.scale_x, n.scale_y);
    for (auto &m : masks)
        m = {radius_mul(m.radius, 1.0f / s), (m.top - n.top) / n.scale_y, (m.left - n.left) / n.scale_x, m.width / n.scale_x,
             m.height / n.scale_y};
}
void parent_parent_masks(const NestedLayout &n, MaskList &masks) {  // :373-389
    float s = rmin(n.scale_x, n.scale_y);
    for (auto &m : masks)
        m = {radius_mul(m.radius, s), (m.top * n.scale_y) + n.top, (m.left * n.scale_x) + n.left, m.width * n.scale_x,
             m.height * n.scale_y};
}
void flatten_child(const NestedLayout &n, RenderLayout &o) {  // :167-305 (in place: the record is large, it travels once per level)
    const RenderLayout c = [&] {  // the child's own values (without its mask list)
        RenderLayout t;
        t.top = o.top; t.left = o.left; t.width = o.width; t.height = o.height; t.rotation_degrees = o.rotation_degrees;
        t.border_radius = o.border_radius; t.content = o.content; t.border_width = o.border_width; t.crop = o.crop; t.blur_radius = o.blur_radius;
        return t;
    }();
    float us = rmin(n.scale_x, n.scale_y);
    if (!n.crop) {
        o.top = n.top + (c.top * n.scale_y); o.left = n.left + (c.left * n.scale_x);
        o.width = c.width * n.scale_x; o.height = c.height * n.scale_y;
        o.border_width = c.border_width * us;
        o.blur_radius = c.blur_radius * us;
    } else {
        const Crop &cr = *n.crop;
        float ctop = rmax(c.top - cr.top, 0.0f), cleft = rmax(c.left - cr.left, 0.0f);
        float cbottom = rmin(c.top + c.height - cr.top, cr.height), cright = rmin(c.left + c.width - cr.left, cr.width);
        float cw = cright - cleft, chh = cbottom - ctop;
        o.top = n.top + (ctop * n.scale_y); o.left = n.left + (cleft * n.scale_x);
        o.width = cw * n.scale_x; o.height = chh * n.scale_y;
        if (c.content == 0) {
            float top_diff = rmax(cr.top - c.top, 0.0f), left_diff = rmax(cr.left - c.left, 0.0f);
            float hs = c.crop.width / c.width, vs = c.crop.height / c.height;
            o.crop = {c.crop.top + (top_diff * vs), c.crop.left + (left_diff * hs), cw * hs, chh * vs};
            // ChildNode keeps its border width under a crop (:262-279)
        } else {
            o.border_width = c.border_width * us;
        }
        o.blur_radius = c.blur_radius * us;
    }
    o.rotation_degrees = c.rotation_degrees + n.rotation_degrees;
    o.border_radius = radius_mul(c.border_radius, us);
    parent_parent_masks(n, o.masks);
}
// scratch lists of the recursion, one pair per depth, reused from frame to frame (the lists of a level are complete before its
// parent reads them, and a level never sees another level's pair)
struct FlattenScratch {
    std::deque<std::vector<RenderLayout>> lists;  // (a deque: growing it leaves the references handed out to the outer levels valid)
    std::vector<RenderLayout> &get(size_t i) {
        while (lists.size() <= i) lists.emplace_back();
        lists[i].clear();
        return lists[i];
    }
};
void inner_flatten(const NestedLayout &n, size_t offset, const MaskList &parent_masks, std::vector<RenderLayout> &shadows,
                   std::vector<RenderLayout> &layouts, FlattenScratch &scratch, size_t depth) {  // :24-82
    RenderLayout me;
    me.top = n.top; me.left = n.left; me.width = n.width; me.height = n.height; me.rotation_degrees = n.rotation_degrees;
    me.border_radius = n.border_radius; me.masks = parent_masks;
    me.border_color = n.border_color; me.border_width = n.border_width;
    if (n.content == 0) {
        me.content = 0; me.index = n.content_index + offset; me.crop = {0.0f, 0.0f, n.content_size.width, n.content_size.height};
        offset += 1;
    } else {
        me.content = 1;
        me.color = n.content == 1 ? n.content_color : RGBA{0, 0, 0, 0};
    }
    for (auto &s : n.box_shadow) {  // box_shadow_layout :339-354
        RenderLayout sh;
        sh.top = n.top + s.offset_y; sh.left = n.left + s.offset_x; sh.width = n.width; sh.height = n.height;
        sh.rotation_degrees = n.rotation_degrees;
        sh.border_radius = radius_add(n.border_radius, s.blur_radius / 2.0f);
        sh.content = 2; sh.color = s.color; sh.blur_radius = s.blur_radius; sh.masks = parent_masks;
        shadows.push_back(std::move(sh));
    }
    MaskList masks = parent_masks;
    if (n.mask) masks.push_back(*n.mask);
    child_parent_masks(n, masks);
    std::vector<RenderLayout> &child_shadows = scratch.get(2 * depth), &child_layouts = scratch.get(2 * depth + 1);
    for (auto &ch : n.children) {
        size_t cnt = ch.child_nodes_count;
        inner_flatten(ch, offset, masks, child_shadows, child_layouts, scratch, depth + 1);
        offset += cnt;
    }
    layouts.reserve(layouts.size() + 1 + child_shadows.size() + child_layouts.size());
    layouts.push_back(std::move(me));
    for (auto &c : child_shadows) { flatten_child(n, c); layouts.push_back(std::move(c)); }
    for (auto &c : child_layouts) { flatten_child(n, c); layouts.push_back(std::move(c)); }
}
bool should_render(const RenderLayout &l, const std::vector<std::optional<Size>> &res, uint32_t W, uint32_t H) {  // :121-164
    if (l.width <= 0.0f || l.height <= 0.0f || l.top > (float)H || l.left > (float)W) return false;
    if (l.content == 1) {
        if (l.color.a == 0) return l.border_color.a != 0 || l.border_width > 0.0f;
        return true;
    }
    if (l.content == 0) {
        if (l.index < res.size() && res[l.index]) {
            // Resolution is usize in the reference: the size went through `as usize`
            if (l.crop.left > (float)as_usize(res[l.index]->width) || l.crop.top > (float)as_usize(res[l.index]->height)) return false;
        }
        if (l.crop.top + l.crop.height < 0.0f || l.crop.left + l.crop.width < 0.0f) return false;
        return true;
    }
    return l.color.a != 0;
}
void fix_final(RenderLayout &l) {  // :84-116
    if (l.content != 2 && l.border_width < 1.0f) l.border_width = 0.0f;
    size_t kept = 0;
    for (auto &m : l.masks) {
        float mt = rmax(m.radius.tl, m.radius.tr), mb = rmax(m.radius.bl, m.radius.br);
        float ml = rmax(m.radius.tl, m.radius.bl), mr = rmax(m.radius.tr, m.radius.br);
        bool skip = m.top + mt <= l.top && m.left + ml <= l.left && m.left + m.width - mr >= l.left + l.width &&
                    m.top + m.height - mb >= l.top + l.height;
        if (!skip) l.masks[kept++] = m;
    }
    l.masks.resize(kept);
}
}  // namespace

std::vector<RenderLayout> NestedLayout::flatten(const std::vector<std::optional<Size>> &res, uint32_t W, uint32_t H) const {  // :10-22
    static thread_local FlattenScratch scratch;
    std::vector<RenderLayout> &shadows = scratch.get(0), &layouts = scratch.get(1), out;
    // inner_flatten's child accumulation must not see this node's own entries: every level collects into lists of its own
    inner_flatten(*this, 0, MaskList{}, shadows, layouts, scratch, 1);
    out.reserve(shadows.size() + layouts.size());
    for (auto *v : {&shadows, &layouts})
        for (auto &l : *v)
            if (should_render(l, res, W, H)) { fix_final(l); out.push_back(std::move(l)); }
    return out;
}

// ------------------------------------------------------------------------------------------------ colours
static double srgb_to_linear(uint8_t c8) {  // wgpu/utils.rs:74-81 (the 256 values, computed once)
    static const std::array<double, 256> table = [] {
        std::array<double, 256> t{};
        for (int i = 0; i < 256; i++) {
            double c = (double)i / 255.0;
            t[i] = c < 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
        }
        return t;
    }();
    return table[c8];
}
void convert_to_shader_color(RGBA c, bool srgb, float out[4]) {  // wgpu/utils.rs:51-72
    double a = (double)c.a / 255.0;
    if (srgb) {
        out[0] = (float)(a * srgb_to_linear(c.r)); out[1] = (float)(a * srgb_to_linear(c.g)); out[2] = (float)(a * srgb_to_linear(c.b));
    } else {
        out[0] = (float)(a * (double)c.r / 255.0); out[1] = (float)(a * (double)c.g / 255.0); out[2] = (float)(a * (double)c.b / 255.0);
    }
    out[3] = (float)a;
}

}  // namespace smr_host
