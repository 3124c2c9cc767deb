// scene_build.cpp — JSON scene (the smelter-api schema) -> stateful component tree -> render graph, and the per-frame
// LayoutProvider call.  Restates smelter-api/src/video/{component,component_into,transition,color,common_into}.rs and
// smelter-render/src/scene/{scene_state,validation,*_component}.rs.  Error strings follow the reference's so that the
// caller's error handling reads the same.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <set>

#include "css_colors.h"
#include "scene.h"

namespace smr_host {

// ------------------------------------------------------------------------------------------------ colours (color.rs)
static std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
static bool parse_channel(const std::string &raw, int radix, uint8_t &out, std::string &err) {  // u8::from_str_radix
    std::string v = trim(raw);
    size_t i = 0;
    if (!v.empty() && v[0] == '+') i = 1;
    unsigned acc = 0;
    bool any = false;
    for (; i < v.size(); i++) {
        int d;
        char c = v[i];
        if (c >= '0' && c <= '9') d = c - '0';
        else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
        else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
        else d = 99;
        if (d >= radix) { any = false; break; }
        acc = acc * radix + d;
        any = true;
        if (acc > 255) { any = false; break; }
    }
    if (!any) { err = "Invalid format. Color representation is not a valid number."; return false; }
    out = (uint8_t)acc;
    return true;
}
static std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    size_t p = 0;
    while (true) {
        size_t q = s.find(sep, p);
        if (q == std::string::npos) { out.push_back(s.substr(p)); return out; }
        out.push_back(s.substr(p, q - p));
        p = q + 1;
    }
}
static std::string strip_call(const std::string &s, const char *prefix) {  // trim_start_matches(prefix).trim_end_matches(')')
    std::string v = s;
    size_t n = strlen(prefix);
    while (v.compare(0, n, prefix) == 0) v = v.substr(n);
    while (!v.empty() && v.back() == ')') v.pop_back();
    return v;
}
bool parse_color(const std::string &raw, RGBA &out, std::string &err) {
    std::string s = trim(raw);
    const NamedColor *lo = CSS_COLORS, *hi = CSS_COLORS + sizeof(CSS_COLORS) / sizeof(CSS_COLORS[0]);
    const NamedColor *it = std::lower_bound(lo, hi, s, [](const NamedColor &c, const std::string &k) { return k.compare(c.name) > 0; });
    if (it != hi && s == it->name) { out = {it->r, it->g, it->b, 255}; return true; }
    if (!s.empty() && s[0] == '#') {
        if (s.size() != 7 && s.size() != 9) { err = "Invalid format. Color has to be in #RRGGBB or #RRGGBBAA format."; return false; }
        out.a = 255;
        if (!parse_channel(s.substr(1, 2), 16, out.r, err) || !parse_channel(s.substr(3, 2), 16, out.g, err) ||
            !parse_channel(s.substr(5, 2), 16, out.b, err))
            return false;
        if (s.size() == 9 && !parse_channel(s.substr(7, 2), 16, out.a, err)) return false;
        return true;
    }
    if (s.compare(0, 4, "rgb(") == 0) {
        auto parts = split(strip_call(s, "rgb("), ',');
        if (parts.size() != 3) { err = "Invalid RGB format."; return false; }
        out.a = 255;
        return parse_channel(parts[0], 10, out.r, err) && parse_channel(parts[1], 10, out.g, err) && parse_channel(parts[2], 10, out.b, err);
    }
    if (s.compare(0, 5, "rgba(") == 0) {
        auto parts = split(strip_call(s, "rgba("), ',');
        if (parts.size() != 4) { err = "Expected three color components and alpha channel."; return false; }
        char *end = nullptr;
        const std::string &as = parts[3];  // str::parse::<f32>: no surrounding whitespace allowed
        float a = as.empty() || isspace((unsigned char)as[0]) ? NAN : std::strtof(as.c_str(), &end);
        if (as.empty() || isspace((unsigned char)as[0]) || end != as.c_str() + as.size()) { err = "Alpha channel parsing failed."; return false; }
        if (!(a >= 0.0f && a <= 1.0f)) { err = "Alpha value out of range. It must be between 0.0 and 1.0"; return false; }
        if (!parse_channel(parts[0], 10, out.r, err) || !parse_channel(parts[1], 10, out.g, err) || !parse_channel(parts[2], 10, out.b, err))
            return false;
        out.a = (uint8_t)std::round(a * 255.0f);
        return true;
    }
    err = "Unsupported color format.";
    return false;
}

// ------------------------------------------------------------------------------------------------ JSON field access
namespace {
struct BuildCtx {
    std::map<std::string, const Stateful *> prev;  // BuildStateTreeCtx::prev_state
    int64_t last_pts = 0;
    const std::map<std::string, Size> *input_resolutions = nullptr;
    const std::map<std::string, Size> *images = nullptr;
    std::set<std::string> ids;  // validate_component_ids_uniqueness
    bool api_only = false;      // stop after the smelter-api conversion (TryFrom<Component>): no renderer registry, no shaper
    smr_text_measure_fn measure = nullptr;  // the caller's text shaper
    void *measure_user = nullptr;
    std::string err;
};

bool fail(BuildCtx &c, const std::string &msg) { c.err = msg; return false; }

// canonical description of the converted component (scene::Component's Debug shape as JSON), see Scene::parse
Json jopt(const OptF &v) { return v ? Json::number(*v) : Json(); }
Json jcolor(RGBA c) { return Json::array({Json::number(c.r), Json::number(c.g), Json::number(c.b), Json::number(c.a)}); }
Json jid(const Stateful &s) { return s.has_id ? Json::string(s.id) : Json(); }
Json jradius(const BorderRadius &r) { return Json::array({Json::number(r.tl), Json::number(r.tr), Json::number(r.br), Json::number(r.bl)}); }
Json jposition(const Position &p) {
    Json j = Json::object();
    j.set("kind", Json::string(p.absolute ? "Absolute" : "Static")).set("width", jopt(p.width)).set("height", jopt(p.height));
    if (p.absolute) {
        j.set("horizontal", Json::array({Json::string(p.from_right ? "Right" : "Left"), Json::number(p.h_offset)}));
        j.set("vertical", Json::array({Json::string(p.from_bottom ? "Bottom" : "Top"), Json::number(p.v_offset)}));
        j.set("rotation_degrees", Json::number(p.rotation_degrees));
    }
    return j;
}
Json jtransition(const std::optional<Transition> &t) {
    if (!t) return Json();
    Json j = Json::object();
    j.set("duration_ns", Json::number((double)t->duration_ns));
    if (t->interp.kind == InterpKind::CubicBezier)
        j.set("interpolation", Json::array({Json::string("CubicBezier"), Json::number(t->interp.x1), Json::number(t->interp.y1),
                                            Json::number(t->interp.x2), Json::number(t->interp.y2)}));
    else j.set("interpolation", Json::string(t->interp.kind == InterpKind::Linear ? "Linear" : "Bounce"));
    j.set("should_interrupt", Json::boolean(t->should_interrupt));
    return j;
}
Json jshadows(const std::vector<BoxShadow> &v) {
    Json a = Json::array();
    for (auto &s : v) {
        Json j = Json::object();
        j.set("offset_x", Json::number(s.offset_x)).set("offset_y", Json::number(s.offset_y)).set("blur_radius", Json::number(s.blur_radius))
            .set("color", jcolor(s.color));
        a.arr.push_back(j);
    }
    return a;
}
Json jchildren(const std::vector<std::unique_ptr<Stateful>> &v) {
    Json a = Json::array();
    for (auto &k : v) a.arr.push_back(k->desc);
    return a;
}
const char *halign_name(HAlign a) { return a == HAlign::Left ? "Left" : a == HAlign::Right ? "Right" : a == HAlign::Justified ? "Justified" : "Center"; }
const char *valign_name(VAlign a) { return a == VAlign::Top ? "Top" : a == VAlign::Bottom ? "Bottom" : a == VAlign::Justified ? "Justified" : "Center"; }

bool check_fields(const Json &j, const char *type_name, std::initializer_list<const char *> allowed, BuildCtx &c) {
    for (auto &kv : j.obj) {
        bool ok = kv.first == "type";
        for (const char *a : allowed) ok = ok || kv.first == a;
        if (!ok) return fail(c, std::string("unknown field `") + kv.first + "` in \"" + type_name + "\" component");
    }
    return true;
}
bool get_f32(const Json &j, const char *key, OptF &out, BuildCtx &c) {
    const Json *v = j.get(key);
    if (!v || v->is_null()) { out = std::nullopt; return true; }
    if (v->kind != Json::Number) return fail(c, std::string("field `") + key + "`: expected a number");
    out = (float)v->num;
    return true;
}
bool get_str(const Json &j, const char *key, std::string &out, bool &present, BuildCtx &c) {
    const Json *v = j.get(key);
    present = false;
    if (!v || v->is_null()) return true;
    if (v->kind != Json::String) return fail(c, std::string("field `") + key + "`: expected a string");
    out = v->str;
    present = true;
    return true;
}
bool get_color(const Json &j, const char *key, RGBA dflt, RGBA &out, BuildCtx &c) {
    std::string s;
    bool present;
    if (!get_str(j, key, s, present, c)) return false;
    if (!present) { out = dflt; return true; }
    return parse_color(s, out, c.err);
}
template <typename E>
bool get_enum(const Json &j, const char *key, std::initializer_list<std::pair<const char *, E>> names, E dflt, E &out, BuildCtx &c) {
    std::string s;
    bool present;
    if (!get_str(j, key, s, present, c)) return false;
    out = dflt;
    if (!present) return true;
    for (auto &n : names)
        if (s == n.first) { out = n.second; return true; }
    return fail(c, std::string("field `") + key + "`: unknown variant `" + s + "`");
}
bool get_shadows(const Json &j, std::vector<BoxShadow> &out, BuildCtx &c) {  // component_into.rs TryFrom<BoxShadow>
    const Json *v = j.get("box_shadow");
    if (!v || v->is_null()) return true;
    if (v->kind != Json::Array) return fail(c, "field `box_shadow`: expected a list");
    for (const Json &s : v->arr) {
        if (s.kind != Json::Object) return fail(c, "field `box_shadow`: expected a list of objects");
        for (auto &kv : s.obj)
            if (kv.first != "offset_x" && kv.first != "offset_y" && kv.first != "color" && kv.first != "blur_radius")
                return fail(c, "unknown field `" + kv.first + "` in box shadow");
        OptF ox, oy, blur;
        BoxShadow b;
        if (!get_f32(s, "offset_x", ox, c) || !get_f32(s, "offset_y", oy, c) || !get_f32(s, "blur_radius", blur, c) ||
            !get_color(s, "color", RGBA{255, 255, 255, 255}, b.color, c))
            return false;
        b.offset_x = ox.value_or(0.0f); b.offset_y = oy.value_or(0.0f); b.blur_radius = blur.value_or(0.0f);
        out.push_back(b);
    }
    return true;
}
bool get_transition(const Json &j, std::optional<Transition> &out, BuildCtx &c) {  // transition.rs:35-72
    const Json *v = j.get("transition");
    if (!v || v->is_null()) return true;
    if (v->kind != Json::Object) return fail(c, "field `transition`: expected an object");
    const Json *d = v->get("duration_ms");
    if (!d || d->kind != Json::Number) return fail(c, "missing field `duration_ms`");
    Transition t;
    const Json *e = v->get("easing_function");
    if (e && !e->is_null()) {
        const Json *fn = e->kind == Json::Object ? e->get("function_name") : nullptr;
        if (!fn || fn->kind != Json::String) return fail(c, "missing field `function_name`");
        if (fn->str == "linear") t.interp.kind = InterpKind::Linear;
        else if (fn->str == "bounce") t.interp.kind = InterpKind::Bounce;
        else if (fn->str == "cubic_bezier") {
            const Json *p = e->get("points");
            if (!p || p->kind != Json::Array || p->arr.size() != 4) return fail(c, "field `points`: expected an array of 4 numbers");
            double pts[4];
            for (int i = 0; i < 4; i++) {
                if (p->arr[i].kind != Json::Number) return fail(c, "field `points`: expected an array of 4 numbers");
                pts[i] = p->arr[i].num;
            }
            if (pts[0] < 0.0 || pts[0] > 1.0) return fail(c, "Control point x1 has to be in the range [0, 1].");
            if (pts[2] < 0.0 || pts[2] > 1.0) return fail(c, "Control point x2 has to be in the range [0, 1].");
            t.interp = {InterpKind::CubicBezier, pts[0], pts[1], pts[2], pts[3]};
        } else return fail(c, "unknown variant `" + fn->str + "`, expected one of `linear`, `bounce`, `cubic_bezier`");
    }
    double s = d->num / 1000.0;  // Duration::try_from_secs_f64
    if (s < 0.0) return fail(c, "Invalid duration. cannot convert float seconds to Duration: value is negative");
    if (!(s < 1.8446744073709552e19)) return fail(c, "Invalid duration. cannot convert float seconds to Duration: value is either too big or NaN");
    double whole = std::floor(s);
    t.duration_ns = s > 2.0e9 ? INT64_MAX / 4  // > 63 years: saturate (the reference's u64 seconds never matter at that range)
                              : (int64_t)whole * 1000000000LL + (int64_t)std::llround((s - whole) * 1e9);
    const Json *si = v->get("should_interrupt");
    if (si && si->kind == Json::Bool) t.should_interrupt = si->b;
    out = t;
    return true;
}
// position of View / Rescaler (component_into.rs:33-79, :138-182)
bool get_position(const Json &j, const char *type_name, Position &pos, BuildCtx &c) {
    OptF top, left, bottom, right, rotation;
    if (!get_f32(j, "width", pos.width, c) || !get_f32(j, "height", pos.height, c) || !get_f32(j, "top", top, c) ||
        !get_f32(j, "left", left, c) || !get_f32(j, "bottom", bottom, c) || !get_f32(j, "right", right, c) ||
        !get_f32(j, "rotation", rotation, c))
        return false;
    pos.absolute = top || bottom || left || right || rotation;
    if (!pos.absolute) return true;
    std::string t = std::string("\"") + type_name + "\"";
    if (top && bottom) return fail(c, "Fields \"top\" and \"bottom\" are mutually exclusive, you can only specify one on a " + t + " component.");
    if (!top && !bottom) return fail(c, t + " component with absolute positioning requires either \"top\" or \"bottom\" coordinate.");
    if (left && right) return fail(c, "Fields \"left\" and \"right\" are mutually exclusive, you can only specify one on a " + t + " component.");
    if (!left && !right) return fail(c, "Non-static " + t + " component requires either \"left\" or \"right\" coordinate.");
    pos.from_bottom = (bool)bottom; pos.v_offset = bottom ? *bottom : *top;
    pos.from_right = (bool)right; pos.h_offset = right ? *right : *left;
    pos.rotation_degrees = rotation.value_or(0.0f);
    return true;
}

std::unique_ptr<Stateful> build(const Json &j, BuildCtx &c);

// ShaderParam (smelter-api/src/video/component.rs:300-324): {"type": f32|u32|i32|list|struct, "value": ...}
bool shader_param(const Json &j, bool struct_field, Json &out, BuildCtx &c) {
    if (j.kind != Json::Object) return fail(c, "shader_param: expected an object");
    for (auto &kv : j.obj)
        if (kv.first != "type" && kv.first != "value" && !(struct_field && kv.first == "field_name"))
            return fail(c, "unknown field `" + kv.first + "` in shader_param");
    const Json *ty = j.get("type"), *v = j.get("value");
    if (!ty || ty->kind != Json::String) return fail(c, "shader_param: missing field `type`");
    if (!v) return fail(c, "shader_param: missing field `value`");
    auto integer = [&](double lo, double hi) { return v->kind == Json::Number && v->num == std::floor(v->num) && v->num >= lo && v->num <= hi; };
    if (ty->str == "f32") {
        if (v->kind != Json::Number) return fail(c, "shader_param: f32 value has to be a number");
        out = Json::array({Json::string("F32"), Json::number((float)v->num)});
    } else if (ty->str == "u32") {
        if (!integer(0.0, 4294967295.0)) return fail(c, "shader_param: u32 value out of range");
        out = Json::array({Json::string("U32"), Json::number(v->num)});
    } else if (ty->str == "i32") {
        if (!integer(-2147483648.0, 2147483647.0)) return fail(c, "shader_param: i32 value out of range");
        out = Json::array({Json::string("I32"), Json::number(v->num)});
    } else if (ty->str == "list" || ty->str == "struct") {
        if (v->kind != Json::Array) return fail(c, "shader_param: list/struct value has to be an array");
        Json items = Json::array();
        for (const Json &e : v->arr) {
            Json p;
            if (!shader_param(e, ty->str == "struct", p, c)) return false;
            if (ty->str == "struct") {
                const Json *name = e.get("field_name");
                if (!name || name->kind != Json::String) return fail(c, "shader_param: missing field `field_name`");
                items.arr.push_back(Json::array({Json::string(name->str), p}));
            } else items.arr.push_back(p);
        }
        out = Json::array({Json::string(ty->str == "list" ? "List" : "Struct"), items});
    } else return fail(c, "shader_param: unknown variant `" + ty->str + "`");
    return true;
}

bool build_children(const Json &j, std::vector<std::unique_ptr<Stateful>> &out, BuildCtx &c) {
    const Json *ch = j.get("children");
    if (!ch || ch->is_null()) return true;
    if (ch->kind != Json::Array) return fail(c, "field `children`: expected a list");
    for (const Json &k : ch->arr) {
        auto s = build(k, c);
        if (!s) return false;
        out.push_back(std::move(s));
    }
    return true;
}
const Stateful *previous(const Stateful &s, Kind kind, BuildCtx &c) {
    if (!s.has_id) return nullptr;
    auto it = c.prev.find(s.id);
    return it != c.prev.end() && it->second->kind == kind ? it->second : nullptr;
}

std::unique_ptr<Stateful> build(const Json &j, BuildCtx &c) {
    if (j.kind != Json::Object) { fail(c, "expected a component object"); return nullptr; }
    const Json *ty = j.get("type");
    if (!ty || ty->kind != Json::String) { fail(c, "missing field `type`"); return nullptr; }
    auto s = std::make_unique<Stateful>();
    if (!get_str(j, "id", s->id, s->has_id, c)) return nullptr;
    if (s->has_id) {
        if (c.ids.count(s->id)) {
            fail(c, "More than one component has an id \"" + s->id + "\". Component IDs in scene definition need to be unique.");
            return nullptr;
        }
        c.ids.insert(s->id);
    }
    const std::string &t = ty->str;
    bool present;
    if (t == "input_stream") {  // input_stream_component.rs:22-44
        s->kind = Kind::InputStream;
        if (!check_fields(j, "InputStream", {"id", "input_id"}, c) || !get_str(j, "input_id", s->ref_id, present, c)) return nullptr;
        if (!present) { fail(c, "missing field `input_id`"); return nullptr; }
        s->desc = Json::object();
        s->desc.set("type", Json::string("InputStream")).set("id", jid(*s)).set("input_id", Json::string(s->ref_id));
        if (c.api_only) return s;
        auto it = c.input_resolutions->find(s->ref_id);
        s->leaf_size = it != c.input_resolutions->end() ? it->second : Size{0.0f, 0.0f};
        return s;
    }
    if (t == "view") {
        s->kind = Kind::View;
        if (!check_fields(j, "View", {"id", "children", "width", "height", "direction", "top", "left", "bottom", "right", "rotation", "transition",
                                      "overflow", "background_color", "border_radius", "border_width", "border_color", "box_shadow", "padding",
                                      "padding_vertical", "padding_horizontal", "padding_top", "padding_right", "padding_bottom", "padding_left"}, c))
            return nullptr;
        ViewParam &v = s->view_end;
        v.id = s->id; v.has_id = s->has_id;
        OptF radius, bw, p, pv, ph, pt, pr, pb, pl;
        std::optional<Transition> tr;
        if (!get_position(j, "View", v.position, c) ||
            !get_enum<bool>(j, "direction", {{"row", false}, {"column", true}}, false, v.column, c) ||
            !get_enum<int>(j, "overflow", {{"visible", 0}, {"hidden", 1}, {"fit", 2}}, 1, v.overflow, c) ||
            !get_color(j, "background_color", RGBA{0, 0, 0, 0}, v.background_color, c) || !get_f32(j, "border_radius", radius, c) ||
            !get_f32(j, "border_width", bw, c) || !get_color(j, "border_color", RGBA{0, 0, 0, 0}, v.border_color, c) ||
            !get_shadows(j, v.box_shadow, c) || !get_f32(j, "padding", p, c) || !get_f32(j, "padding_vertical", pv, c) ||
            !get_f32(j, "padding_horizontal", ph, c) || !get_f32(j, "padding_top", pt, c) || !get_f32(j, "padding_right", pr, c) ||
            !get_f32(j, "padding_bottom", pb, c) || !get_f32(j, "padding_left", pl, c) || !get_transition(j, tr, c))
            return nullptr;
        auto pick = [](OptF a, OptF b, OptF d) { return a ? *a : b ? *b : d ? *d : 0.0f; };
        v.padding = {pick(pt, pv, p), pick(pr, ph, p), pick(pb, pv, p), pick(pl, ph, p)};
        if (v.padding.top < 0.0f || v.padding.right < 0.0f || v.padding.bottom < 0.0f || v.padding.left < 0.0f) {
            fail(c, "Padding values cannot be negative.");
            return nullptr;
        }
        float r = radius.value_or(0.0f);
        v.border_radius = {r, r, r, r};
        v.border_width = bw.value_or(0.0f);
        // view_component.rs:98-160
        const Stateful *prev = previous(*s, Kind::View, c);
        if (prev) s->view_start = prev->view(c.last_pts);
        bool changed = prev ? !(prev->view_end == v) : false;
        s->transition = TransitionState::make(tr, prev ? prev->transition : std::nullopt, changed, tr ? tr->should_interrupt : false, c.last_pts);
        if (!build_children(j, s->children, c)) return nullptr;
        s->desc = Json::object();
        s->desc.set("type", Json::string("View")).set("id", jid(*s)).set("children", jchildren(s->children))
            .set("direction", Json::string(v.column ? "Column" : "Row")).set("position", jposition(v.position)).set("transition", jtransition(tr))
            .set("overflow", Json::string(v.overflow == 0 ? "Visible" : v.overflow == 1 ? "Hidden" : "Fit"))
            .set("background_color", jcolor(v.background_color)).set("border_radius", jradius(v.border_radius))
            .set("border_width", Json::number(v.border_width)).set("border_color", jcolor(v.border_color)).set("box_shadow", jshadows(v.box_shadow))
            .set("padding", Json::array({Json::number(v.padding.top), Json::number(v.padding.right), Json::number(v.padding.bottom),
                                         Json::number(v.padding.left)}));
        return s;
    }
    if (t == "rescaler") {
        s->kind = Kind::Rescaler;
        if (!check_fields(j, "Rescaler", {"id", "child", "mode", "horizontal_align", "vertical_align", "width", "height", "top", "left", "bottom",
                                          "right", "rotation", "transition", "border_radius", "border_width", "border_color", "box_shadow"}, c))
            return nullptr;
        RescalerParam &r = s->resc_end;
        r.id = s->id; r.has_id = s->has_id;
        OptF radius, bw;
        std::optional<Transition> tr;
        if (!get_position(j, "Rescaler", r.position, c) || !get_enum<bool>(j, "mode", {{"fit", false}, {"fill", true}}, false, r.fill, c) ||
            !get_enum<HAlign>(j, "horizontal_align", {{"left", HAlign::Left}, {"right", HAlign::Right}, {"justified", HAlign::Justified}, {"center", HAlign::Center}},
                              HAlign::Center, r.horizontal_align, c) ||
            !get_enum<VAlign>(j, "vertical_align", {{"top", VAlign::Top}, {"center", VAlign::Center}, {"bottom", VAlign::Bottom}, {"justified", VAlign::Justified}},
                              VAlign::Center, r.vertical_align, c) ||
            !get_f32(j, "border_radius", radius, c) || !get_f32(j, "border_width", bw, c) ||
            !get_color(j, "border_color", RGBA{0, 0, 0, 0}, r.border_color, c) || !get_shadows(j, r.box_shadow, c) || !get_transition(j, tr, c))
            return nullptr;
        float rr = radius.value_or(0.0f);
        r.border_radius = {rr, rr, rr, rr};
        r.border_width = bw.value_or(0.0f);
        const Json *child = j.get("child");
        if (!child || child->is_null()) { fail(c, "missing field `child`"); return nullptr; }
        // rescaler_component.rs stateful_component — same shape as the View's
        const Stateful *prev = previous(*s, Kind::Rescaler, c);
        if (prev) s->resc_start = prev->rescaler(c.last_pts);
        bool changed = prev ? !(prev->resc_end == r) : false;
        s->transition = TransitionState::make(tr, prev ? prev->transition : std::nullopt, changed, tr ? tr->should_interrupt : false, c.last_pts);
        auto k = build(*child, c);
        if (!k) return nullptr;
        s->children.push_back(std::move(k));
        s->desc = Json::object();
        s->desc.set("type", Json::string("Rescaler")).set("id", jid(*s)).set("child", s->children[0]->desc).set("position", jposition(r.position))
            .set("transition", jtransition(tr)).set("mode", Json::string(r.fill ? "Fill" : "Fit"))
            .set("horizontal_align", Json::string(halign_name(r.horizontal_align))).set("vertical_align", Json::string(valign_name(r.vertical_align)))
            .set("border_radius", jradius(r.border_radius)).set("border_width", Json::number(r.border_width))
            .set("border_color", jcolor(r.border_color)).set("box_shadow", jshadows(r.box_shadow));
        return s;
    }
    if (t == "tiles") {
        s->kind = Kind::Tiles;
        if (!check_fields(j, "Tiles", {"id", "children", "width", "height", "background_color", "tile_aspect_ratio", "margin", "padding",
                                       "horizontal_align", "vertical_align", "transition"}, c))
            return nullptr;
        TilesParam &tp = s->tiles;
        tp.id = s->id; tp.has_id = s->has_id;
        OptF margin, padding;
        std::string ar;
        std::optional<Transition> tr;
        if (!get_f32(j, "width", tp.width, c) || !get_f32(j, "height", tp.height, c) ||
            !get_color(j, "background_color", RGBA{0, 0, 0, 0}, tp.background_color, c) || !get_str(j, "tile_aspect_ratio", ar, present, c) ||
            !get_f32(j, "margin", margin, c) || !get_f32(j, "padding", padding, c) ||
            !get_enum<HAlign>(j, "horizontal_align", {{"left", HAlign::Left}, {"right", HAlign::Right}, {"justified", HAlign::Justified}, {"center", HAlign::Center}},
                              HAlign::Center, tp.horizontal_align, c) ||
            !get_enum<VAlign>(j, "vertical_align", {{"top", VAlign::Top}, {"center", VAlign::Center}, {"bottom", VAlign::Bottom}, {"justified", VAlign::Justified}},
                              VAlign::Center, tp.vertical_align, c) ||
            !get_transition(j, tr, c))
            return nullptr;
        if (present) {  // common_into.rs:27-43
            const char *MSG = "Aspect ratio needs to be a string in the \"W:H\" format, where W and H are both unsigned integers.";
            size_t colon = ar.find(':');
            auto parse_u32 = [](const std::string &v, uint32_t &o) {
                size_t i = !v.empty() && v[0] == '+' ? 1 : 0;
                if (i >= v.size()) return false;
                uint64_t acc = 0;
                for (; i < v.size(); i++) {
                    if (v[i] < '0' || v[i] > '9') return false;
                    acc = acc * 10 + (uint64_t)(v[i] - '0');
                    if (acc > 0xFFFFFFFFull) return false;
                }
                o = (uint32_t)acc;
                return true;
            };
            if (colon == std::string::npos || !parse_u32(ar.substr(0, colon), tp.ar_w) || !parse_u32(ar.substr(colon + 1), tp.ar_h)) {
                fail(c, MSG);
                return nullptr;
            }
        }
        tp.margin = margin.value_or(0.0f);
        tp.padding = padding.value_or(0.0f);
        if (!build_children(j, s->children, c)) return nullptr;
        // tiles_component.rs:116-180
        const Stateful *prev = previous(*s, Kind::Tiles, c);
        if (prev) { s->tiles_start = prev->tiles_last_layout; s->tiles_last_layout = prev->tiles_last_layout; }
        bool changed = false;
        if (prev) {
            changed = !(prev->tiles == tp) || prev->children.size() != s->children.size();
            for (size_t i = 0; !changed && i < s->children.size(); i++) {
                const Stateful &a = *prev->children[i], &b = *s->children[i];
                changed = a.has_id != b.has_id || (a.has_id && a.id != b.id);
            }
        }
        s->transition = TransitionState::make(tr, prev ? prev->transition : std::nullopt, changed, tr ? tr->should_interrupt : false, c.last_pts);
        s->desc = Json::object();
        s->desc.set("type", Json::string("Tiles")).set("id", jid(*s)).set("width", jopt(tp.width)).set("height", jopt(tp.height))
            .set("margin", Json::number(tp.margin)).set("padding", Json::number(tp.padding)).set("children", jchildren(s->children))
            .set("transition", jtransition(tr)).set("vertical_align", Json::string(valign_name(tp.vertical_align)))
            .set("horizontal_align", Json::string(halign_name(tp.horizontal_align))).set("background_color", jcolor(tp.background_color))
            .set("tile_aspect_ratio", Json::array({Json::number(tp.ar_w), Json::number(tp.ar_h)}));
        return s;
    }
    if (t == "image") {  // image_component.rs
        s->kind = Kind::Image;
        OptF w, h;
        if (!check_fields(j, "Image", {"id", "image_id", "width", "height"}, c) || !get_str(j, "image_id", s->ref_id, present, c) ||
            !get_f32(j, "width", w, c) || !get_f32(j, "height", h, c))
            return nullptr;
        if (!present) { fail(c, "missing field `image_id`"); return nullptr; }
        s->desc = Json::object();
        s->desc.set("type", Json::string("Image")).set("id", jid(*s)).set("image_id", Json::string(s->ref_id)).set("width", jopt(w)).set("height", jopt(h));
        if (c.api_only) return s;
        auto it = c.images->find(s->ref_id);
        if (it == c.images->end()) {
            fail(c, "Image \"" + s->ref_id + "\" does not exist. You have to register it first before using it in the scene definition.");
            return nullptr;
        }
        // the reference divides the usize dimensions before the float conversion (integer aspect ratio)
        size_t iw = as_usize(it->second.width), ih = as_usize(it->second.height);
        float aspect = ih ? (float)(iw / ih) : 0.0f;
        size_t rw, rh;
        if (w && h) { rw = as_usize(std::round(*w)); rh = as_usize(std::round(*h)); }
        else if (w) { rw = as_usize(std::round(*w)); rh = as_usize(std::round(*w / aspect)); }
        else if (h) { rw = as_usize(std::round(*h * aspect)); rh = as_usize(std::round(*h)); }
        else { rw = iw; rh = ih; }
        s->leaf_size = {(float)rw, (float)rh};
        return s;
    }
    if (t == "text") {  // text_component.rs; only TextDimensions::Fixed can be sized without the text shaper
        s->kind = Kind::Text;
        OptF w, h, mw, mh, fs, lh;
        std::string family = "Verdana", style, wrap, weight;
        bool has_family;
        HAlign align;
        RGBA color, bg;
        if (!check_fields(j, "Text", {"id", "text", "width", "height", "max_width", "max_height", "font_size", "line_height", "color",
                                      "background_color", "font_family", "style", "align", "wrap", "weight"}, c) ||
            !get_str(j, "text", s->text, present, c) || !get_f32(j, "width", w, c) || !get_f32(j, "height", h, c) ||
            !get_f32(j, "max_width", mw, c) || !get_f32(j, "max_height", mh, c) || !get_f32(j, "font_size", fs, c) ||
            !get_f32(j, "line_height", lh, c))
            return nullptr;
        if (!present) { fail(c, "missing field `text`"); return nullptr; }
        if (!fs) { fail(c, "missing field `font_size`"); return nullptr; }
        std::string fam;
        if (!get_str(j, "font_family", fam, has_family, c) ||
            !get_enum<std::string>(j, "style", {{"normal", "Normal"}, {"italic", "Italic"}, {"oblique", "Oblique"}}, "Normal", style, c) ||
            !get_enum<std::string>(j, "wrap", {{"none", "None"}, {"glyph", "Glyph"}, {"word", "Word"}}, "None", wrap, c) ||
            !get_enum<std::string>(j, "weight", {{"thin", "Thin"}, {"extra_light", "ExtraLight"}, {"light", "Light"}, {"normal", "Normal"},
                                                 {"medium", "Medium"}, {"semi_bold", "SemiBold"}, {"bold", "Bold"}, {"extra_bold", "ExtraBold"},
                                                 {"black", "Black"}}, "Normal", weight, c) ||
            !get_enum<HAlign>(j, "align", {{"left", HAlign::Left}, {"right", HAlign::Right}, {"justified", HAlign::Justified}, {"center", HAlign::Center}},
                              HAlign::Left, align, c))
            return nullptr;
        if (has_family) family = fam;
        // component_into.rs:285-320 — the dimensions are resolved before font_size / line_height are validated, colours last
        if (!w && h) { fail(c, "\"height\" property on a Text component can only be provided if \"width\" is also defined."); return nullptr; }
        if (*fs <= 0.0f) { fail(c, "\"font_size\" property has to be larger than 0"); return nullptr; }
        if (lh.value_or(*fs) <= 0.0f) { fail(c, "\"line_height\" property has to be larger than 0"); return nullptr; }
        if (!get_color(j, "color", RGBA{255, 255, 255, 255}, color, c) || !get_color(j, "background_color", RGBA{0, 0, 0, 0}, bg, c)) return nullptr;
        const float MAX_W = 7682.0f, MAX_H = 4320.0f;  // MAX_NODE_RESOLUTION (smelter-render/src/types.rs:146-149)
        Json dims = w && h ? Json::array({Json::string("Fixed"), Json::number(*w), Json::number(*h)})
                    : w    ? Json::array({Json::string("FittedColumn"), Json::number(*w), Json::number(mh.value_or(MAX_H))})
                           : Json::array({Json::string("Fitted"), Json::number(mw.value_or(MAX_W)), Json::number(mh.value_or(MAX_H))});
        s->desc = Json::object();
        s->desc.set("type", Json::string("Text")).set("id", jid(*s)).set("text", Json::string(s->text)).set("font_size", Json::number(*fs))
            .set("line_height", Json::number(lh.value_or(*fs))).set("color", jcolor(color)).set("font_family", Json::string(family))
            .set("style", Json::string(style)).set("align", Json::string(halign_name(align))).set("weight", Json::string(weight))
            .set("wrap", Json::string(wrap)).set("background_color", jcolor(bg)).set("dimensions", dims);
        s->text_spec.font_size = *fs; s->text_spec.line_height = lh.value_or(*fs);
        s->text_spec.color = color; s->text_spec.background = bg;
        s->text_spec.family = family; s->text_spec.style = style; s->text_spec.weight = weight; s->text_spec.wrap = wrap;
        s->text_spec.align = halign_name(align);
        if (c.api_only) return s;
        if (!w || !h) {
            // TextRendererCtx::layout_text (text_renderer.rs:282-346): Fitted { max_width, max_height } / FittedColumn { width, max_height }
            if (!c.measure) {
                fail(c, "Text components without \"width\" and \"height\" need a text shaper: smr_renderer_set_text_measurer (SURVEY.md §8 a12)");
                return nullptr;
            }
            const float line_height = lh.value_or(*fs);
            smr_text_params tp;
            tp.text = s->text.c_str(); tp.font_family = family.c_str(); tp.style = style.c_str(); tp.weight = weight.c_str(); tp.wrap = wrap.c_str();
            const std::string an = halign_name(align);
            tp.align = an.c_str();
            tp.font_size = *fs; tp.line_height = line_height;
            tp.max_width = w ? *w : mw.value_or(MAX_W);
            tp.max_height = mh.value_or(MAX_H);
            float widest = 0.0f;
            uint32_t lines = 0;
            if (c.measure(c.measure_user, &tp, &widest, &lines) != 0) { fail(c, "the text shaper failed to lay out \"" + s->text + "\""); return nullptr; }
            // get_text_resolution (text_renderer.rs:348-368)
            const size_t tw = as_usize(std::ceil(widest > 0.0f ? widest : 0.0f));
            const size_t th = as_usize((float)lines * std::ceil(line_height) + *fs / 5.0f);
            s->leaf_size = {(float)(w ? as_usize(*w) : tw), (float)th};
            s->shader_param = j;
            return s;
        }
        s->leaf_size = {(float)as_usize(*w), (float)as_usize(*h)};  // Resolution { width as usize, height as usize }
        s->shader_param = j;                                    // colours / font properties for the caller's shaper
        return s;
    }
    if (t == "shader") {  // shader_component.rs
        s->kind = Kind::Shader;
        if (!check_fields(j, "Shader", {"id", "children", "shader_id", "shader_param", "resolution"}, c) ||
            !get_str(j, "shader_id", s->ref_id, present, c))
            return nullptr;
        if (!present) { fail(c, "missing field `shader_id`"); return nullptr; }
        const Json *res = j.get("resolution");
        const Json *rw = res ? res->get("width") : nullptr, *rh = res ? res->get("height") : nullptr;
        if (!rw || !rh || rw->kind != Json::Number || rh->kind != Json::Number || rw->num < 0 || rh->num < 0) {
            fail(c, "missing field `resolution`");
            return nullptr;
        }
        s->leaf_size = {(float)as_usize(rw->num), (float)as_usize(rh->num)};
        Json param;
        if (const Json *p = j.get("shader_param")) {
            s->shader_param = *p;
            if (!p->is_null() && !shader_param(*p, false, param, c)) return nullptr;
        }
        if (!build_children(j, s->children, c)) return nullptr;
        s->desc = Json::object();
        s->desc.set("type", Json::string("Shader")).set("id", jid(*s)).set("shader_id", Json::string(s->ref_id)).set("shader_param", param)
            .set("size", Json::array({Json::number(s->leaf_size.width), Json::number(s->leaf_size.height)})).set("children", jchildren(s->children));
        return s;
    }
    if (t == "web_view") {  // parsed for API parity; there is no browser to render it (out of scope)
        std::string instance;
        if (!check_fields(j, "WebView", {"id", "children", "instance_id"}, c) || !get_str(j, "instance_id", instance, present, c)) return nullptr;
        if (!present) { fail(c, "missing field `instance_id`"); return nullptr; }
        if (!c.api_only) { fail(c, "Instance of web renderer \"" + instance + "\" does not exist. You have to register it first before using it in the scene definition."); return nullptr; }
        s->kind = Kind::Shader;
        if (!build_children(j, s->children, c)) return nullptr;
        s->desc = Json::object();
        s->desc.set("type", Json::string("WebView")).set("id", jid(*s)).set("children", jchildren(s->children)).set("instance_id", Json::string(instance));
        return s;
    }
    fail(c, "unknown variant `" + t + "`, expected one of `input_stream`, `view`, `web_view`, `shader`, `image`, `text`, `tiles`, `rescaler`");
    return nullptr;
}

void gather_ids(const Stateful &s, std::map<std::string, const Stateful *> &out) {  // scene_state.rs:266-310
    if (s.has_id) out[s.id] = &s;
    for (auto &ch : s.children) gather_ids(*ch, out);
}
void recalculate_layout(Stateful &s, std::optional<Size> size, int64_t pts, bool parent_is_layout) {  // scene_state.rs:233-264
    if (s.is_layout()) {
        if (!parent_is_layout) {
            OptF w = s.width(pts), h = s.height(pts);
            if (!size && w && h) size = Size{*w, *h};
            if (size) s.layout(*size, pts);
        }
        for (auto &ch : s.children) recalculate_layout(*ch, std::nullopt, pts, true);
    } else {
        for (auto &ch : s.children) recalculate_layout(*ch, std::nullopt, pts, false);
    }
}
// IntermediateNode::build_tree (scene_state.rs:148-231)
bool build_graph(std::vector<GraphNode> &nodes, Stateful *c, int parent, const std::optional<Size> &forced, int64_t pts, std::string &err) {
    int idx = (int)nodes.size();
    GraphNode g;
    g.kind = c->kind; g.component = c; g.parent = parent;
    nodes.push_back(g);
    if (parent >= 0) nodes[parent].children.push_back(idx);
    if (c->is_layout()) {
        Size size;
        if (forced) size = *forced;
        else {
            Position p = c->position(pts);
            if (!p.width || !p.height) {
                const char *name = c->kind == Kind::View ? "View" : c->kind == Kind::Tiles ? "Tiles" : "Rescaler";
                err = std::string("\"") + name + "\" that is a child of an non-layout component e.g. \"Shader\", \"WebView\" need to have known size. " +
                      (c->has_id ? "Please provide width and height values for component with id \"" + c->id + "\"" : "Please provide width and height values.");
                return false;
            }
            size = {*p.width, *p.height};
        }
        nodes[idx].forced_size = size;
        nodes[idx].has_forced_size = true;
        std::vector<Stateful *> kids;
        c->node_children(kids);
        for (Stateful *k : kids)
            if (!build_graph(nodes, k, idx, std::nullopt, pts, err)) return false;
    } else if (c->kind == Kind::Shader) {
        for (auto &k : c->children)
            if (!build_graph(nodes, k.get(), idx, std::nullopt, pts, err)) return false;
    }
    return true;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ Scene
bool Scene::update(const std::string &json, uint32_t out_w, uint32_t out_h, std::string &err) {
    Json j;
    JsonParser parser(json);
    if (!parser.parse(j, err)) return false;
    if (root_) recalculate_layout(*root_, Size{(float)out_w_, (float)out_h_}, last_pts_ns_, false);
    BuildCtx ctx;
    if (root_) gather_ids(*root_, ctx.prev);
    ctx.last_pts = last_pts_ns_;
    ctx.input_resolutions = &input_resolutions_;
    ctx.images = &images_;
    ctx.measure = measure_;
    ctx.measure_user = measure_user_;
    std::unique_ptr<Stateful> root = build(j, ctx);
    if (!root) { err = ctx.err; return false; }
    std::vector<GraphNode> nodes;
    if (!build_graph(nodes, root.get(), -1, Size{(float)out_w, (float)out_h}, last_pts_ns_, err)) return false;
    root_ = std::move(root);
    nodes_ = std::move(nodes);
    out_w_ = out_w; out_h_ = out_h;
    return true;
}

bool Scene::parse(const std::string &json, std::string &out, std::string &err) const {
    Json j;
    JsonParser parser(json);
    if (!parser.parse(j, err)) return false;
    BuildCtx ctx;
    ctx.api_only = true;
    ctx.input_resolutions = &input_resolutions_;
    ctx.images = &images_;
    std::unique_ptr<Stateful> root = build(j, ctx);
    if (!root) { err = ctx.err; return false; }
    out.clear();
    root->desc.dump(out);
    return true;
}

// Is any component under `s` still moving at pts?  (TransitionState::is_finished, scene/transition.rs:79-81)
static bool in_transition(const Stateful &s, int64_t pts_ns) {
    if (s.transition && !s.transition->is_finished(pts_ns)) return true;
    for (const std::unique_ptr<Stateful> &c : s.children)
        if (c && in_transition(*c, pts_ns)) return true;
    return false;
}

bool Scene::node_layouts(int node, int64_t pts_ns, const std::vector<std::optional<Size>> &child_resolutions, bool srgb,
                         std::vector<smr_layout> &out, uint32_t &w, uint32_t &h, std::string &err) {
    if (node < 0 || node >= (int)nodes_.size() || !nodes_[node].component->is_layout()) { err = "node_layouts: not a layout node"; return false; }
    GraphNode &g = nodes_[node];
    if (child_resolutions.size() != g.children.size()) { err = "node_layouts: one resolution per child node is required"; return false; }
    last_pts_ns_ = pts_ns;  // register_render_event
    for (size_t i = 0; i < g.children.size(); i++) {
        const GraphNode &k = nodes_[g.children[i]];
        if (k.kind != Kind::InputStream) continue;
        if (child_resolutions[i]) input_resolutions_[k.component->ref_id] = *child_resolutions[i];
        else input_resolutions_.erase(k.component->ref_id);
    }
    Stateful &root = *g.component;
    if (g.cache.valid && g.cache.srgb == srgb && pts_ns >= g.cache.pts_ns && g.cache.resolutions.size() == child_resolutions.size()) {
        bool same = true;
        for (size_t i = 0; i < child_resolutions.size() && same; i++) {
            const std::optional<Size> &a = g.cache.resolutions[i], &b = child_resolutions[i];
            same = a.has_value() == b.has_value() && (!a || (a->width == b->width && a->height == b->height));
        }
        if (same) {
            out = g.cache.layouts;
            w = g.cache.w; h = g.cache.h;
            return true;
        }
    }
    g.cache.valid = false;
    // SizedLayoutComponent::resolution (scene/layout.rs:245-257)
    Position p = root.position(pts_ns);
    w = as_u32(p.width ? *p.width : g.forced_size.width);
    h = as_u32(p.height ? *p.height : g.forced_size.height);
    root.update_state(child_resolutions, 0);
    NestedLayout nested = root.layout(g.forced_size, pts_ns);
    std::vector<RenderLayout> flat = nested.flatten(child_resolutions, w, h);
    out.clear();
    out.reserve(flat.size());
    for (const RenderLayout &l : flat) {  // ParamsBindGroups::update (transformations/layout/params.rs:223-303)
        smr_layout o;
        memset(&o, 0, sizeof(o));
        o.source_index = SMR_NO_SOURCE;
        o.top = l.top; o.left = l.left; o.width = l.width; o.height = l.height; o.rotation_degrees = l.rotation_degrees;
        o.border_radius[0] = l.border_radius.tl; o.border_radius[1] = l.border_radius.tr;
        o.border_radius[2] = l.border_radius.br; o.border_radius[3] = l.border_radius.bl;
        if (l.content == 0) {
            o.type = 0u; o.source_index = (uint32_t)l.index;
            convert_to_shader_color(l.border_color, srgb, o.border_color);
            o.border_width = l.border_width;
            o.crop[0] = l.crop.top; o.crop[1] = l.crop.left; o.crop[2] = l.crop.width; o.crop[3] = l.crop.height;
        } else if (l.content == 1) {
            o.type = 1u;
            convert_to_shader_color(l.color, srgb, o.color);
            convert_to_shader_color(l.border_color, srgb, o.border_color);
            o.border_width = l.border_width;
        } else {
            o.type = 2u;
            convert_to_shader_color(l.color, srgb, o.color);
            o.blur_radius = l.blur_radius;
        }
        size_t nm = std::min<size_t>(l.masks.size(), SMR_MAX_MASKS);
        o.masks_len = (uint32_t)nm;
        for (size_t m = 0; m < nm; m++) {
            const MaskL &mk = l.masks[m];
            o.masks[m].radius[0] = mk.radius.tl; o.masks[m].radius[1] = mk.radius.tr; o.masks[m].radius[2] = mk.radius.br; o.masks[m].radius[3] = mk.radius.bl;
            o.masks[m].top = mk.top; o.masks[m].left = mk.left; o.masks[m].width = mk.width; o.masks[m].height = mk.height;
        }
        out.push_back(o);
    }
    if (!in_transition(root, pts_ns)) {  // (transitions only ever end between two scene updates: a later pts is at rest too)
        g.cache.valid = true; g.cache.srgb = srgb; g.cache.pts_ns = pts_ns; g.cache.w = w; g.cache.h = h;
        g.cache.resolutions = child_resolutions;
        g.cache.layouts = out;
    }
    return true;
}

}  // namespace smr_host
