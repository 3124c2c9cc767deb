// text.cpp — TrueType reader, line layout and exact-area glyph rasteriser behind smr_fontbook_* (text.h; reference:
// smelter-render/src/transformations/text_renderer.rs:72-167, 236-368).  Host code only, no GPU; the run it produces is drawn by
// smr_blit_glyphs.  Kept operation for operation in step with tests/text_twin.py (tests/test_text_capi.py: byte-identical runs):
// every quantity is a double, products and sums are written in the order the Python evaluates them, and this file is compiled with
// -ffp-contract=off like everything else (smelter_amd/build.py).
#include "text.h"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>

namespace smr_text {

namespace {

constexpr uint32_t tag(const char (&s)[5]) { return ((uint32_t)(uint8_t)s[0] << 24) | ((uint32_t)(uint8_t)s[1] << 16) | ((uint32_t)(uint8_t)s[2] << 8) | (uint8_t)s[3]; }

int popcount16(uint32_t v) {
    int n = 0;
    for (; v; v &= v - 1) n++;
    return n;
}

std::string lower(std::string s) {
    for (char &c : s)
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    return s;
}

std::string strip(const std::string &s) {  // str.strip(): ASCII whitespace is all a font name carries
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || (s[a] >= '\t' && s[a] <= '\r'))) a++;
    while (b > a && (s[b - 1] == ' ' || (s[b - 1] >= '\t' && s[b - 1] <= '\r'))) b--;
    return s.substr(a, b - a);
}

void append_utf8(std::string &out, uint32_t cp) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) { out += (char)(0xc0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3f)); }
    else if (cp < 0x10000) { out += (char)(0xe0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3f)); out += (char)(0x80 | (cp & 0x3f)); }
    else { out += (char)(0xf0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3f)); out += (char)(0x80 | ((cp >> 6) & 0x3f)); out += (char)(0x80 | (cp & 0x3f)); }
}

// strict UTF-8 -> code points (what bytes.decode() accepts)
bool decode_utf8(const std::string &s, std::vector<uint32_t> &out) {
    for (size_t i = 0; i < s.size();) {
        const uint8_t c = (uint8_t)s[i];
        uint32_t cp;
        int n;
        if (c < 0x80) { cp = c; n = 0; }
        else if ((c & 0xe0) == 0xc0) { cp = c & 0x1f; n = 1; }
        else if ((c & 0xf0) == 0xe0) { cp = c & 0x0f; n = 2; }
        else if ((c & 0xf8) == 0xf0) { cp = c & 0x07; n = 3; }
        else return false;
        if (i + n >= s.size() + (n ? 0 : 1) && n) return false;
        for (int k = 1; k <= n; k++) {
            if (i + k >= s.size()) return false;
            const uint8_t d = (uint8_t)s[i + k];
            if ((d & 0xc0) != 0x80) return false;
            cp = (cp << 6) | (d & 0x3f);
        }
        if ((n == 1 && cp < 0x80) || (n == 2 && cp < 0x800) || (n == 3 && cp < 0x10000) || cp > 0x10ffff || (cp >= 0xd800 && cp <= 0xdfff)) return false;
        out.push_back(cp);
        i += (size_t)n + 1;
    }
    return true;
}

// str.isspace() for one code point: bidirectional class WS / B / S or category Zs
bool is_space(uint32_t c) {
    return (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x20) || c == 0x85 || c == 0xa0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200a) || c == 0x2028 ||
           c == 0x2029 || c == 0x202f || c == 0x205f || c == 0x3000;
}

int weight_of(const std::string &name) {
    static const std::pair<const char *, int> W[] = {{"Thin", 100}, {"ExtraLight", 200}, {"Light", 300}, {"Normal", 400}, {"Medium", 500},
                                                     {"SemiBold", 600}, {"Bold", 700}, {"ExtraBold", 800}, {"Black", 900}};
    for (const auto &w : W)
        if (name == w.first) return w.second;
    return 400;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// sfnt reading
// ---------------------------------------------------------------------------------------------------------------------------------

uint32_t Font::u8(size_t o) const {
    if (o >= d_.size()) { bad_ = true; return 0; }
    return d_[o];
}
uint32_t Font::u16(size_t o) const {
    if (o + 2 > d_.size()) { bad_ = true; return 0; }
    return ((uint32_t)d_[o] << 8) | d_[o + 1];
}
int32_t Font::i16(size_t o) const { return (int32_t)(int16_t)u16(o); }
uint32_t Font::u32(size_t o) const {
    if (o + 4 > d_.size()) { bad_ = true; return 0; }
    return ((uint32_t)d_[o] << 24) | ((uint32_t)d_[o + 1] << 16) | ((uint32_t)d_[o + 2] << 8) | d_[o + 3];
}

Font::Table Font::table(uint32_t t) const {
    auto it = tables_.find(t);
    return it == tables_.end() ? Table{} : it->second;
}

std::unique_ptr<Font> Font::load(std::vector<uint8_t> data, std::string &err) {
    std::unique_ptr<Font> f(new Font());
    f->d_ = std::move(data);
    if (!f->parse(err)) return nullptr;
    return f;
}

bool Font::parse(std::string &err) {
    size_t base = 0;
    if (d_.size() >= 16 && u32(0) == tag("ttcf")) base = u32(12);  // a collection: its first face
    const uint32_t version = u32(base);
    if (version == tag("OTTO")) { err = "CFF outlines are not supported (TrueType `glyf` faces only)"; return false; }
    if (version != 0x00010000u && version != tag("true")) { err = "not an sfnt font file"; return false; }
    const uint32_t n = u16(base + 4);
    for (uint32_t i = 0; i < n; i++) {
        const size_t r = base + 12 + 16 * (size_t)i;
        Table t;
        t.off = u32(r + 8); t.len = u32(r + 12);
        if ((size_t)t.off + t.len > d_.size()) { err = "truncated font file"; return false; }
        tables_[u32(r)] = t;
    }
    for (uint32_t need : {tag("head"), tag("hhea"), tag("maxp"), tag("hmtx"), tag("cmap"), tag("loca"), tag("glyf")})
        if (!has(need)) { err = "a required TrueType table is missing"; return false; }
    const Table head = table(tag("head")), hhea = table(tag("hhea")), maxp = table(tag("maxp"));
    upem = (double)u16(head.off + 18);
    loca_long_ = i16(head.off + 50);
    ascent = (double)i16(hhea.off + 4);
    descent = (double)-i16(hhea.off + 6);
    n_hmetrics_ = u16(hhea.off + 34);
    n_glyphs_ = u16(maxp.off + 4);
    // (unitsPerEm: 16 .. 16384 by the OpenType specification — what ttf-parser, the reference's reader, insists on as well)
    if (upem < 16.0 || upem > 16384.0 || !n_hmetrics_ || !n_glyphs_) { err = "degenerate font header"; return false; }
    if (has(tag("OS/2"))) {
        const Table os2 = table(tag("OS/2"));
        weight = (int)u16(os2.off + 4);
        italic = (u16(os2.off + 62) & 1u) != 0;
    }
    parse_name();
    choose_cmap();
    if (!cmap_format_) { err = "no Unicode cmap subtable this reader understands"; return false; }
    parse_gpos_kern();
    if (bad_) { err = "malformed font tables"; return false; }
    return true;
}

// name.getDebugName(16) or getDebugName(1): the first record of the id that is (Macintosh, English) or (Windows, en-US); failing that the
// last record of the id that decodes
void Font::parse_name() {
    if (!has(tag("name"))) return;
    const Table t = table(tag("name"));
    const uint32_t count = u16(t.off + 2), strings = t.off + u16(t.off + 4);
    for (uint32_t want : {16u, 1u}) {
        std::string english, some;
        bool have_some = false;
        for (uint32_t i = 0; i < count; i++) {
            const size_t r = t.off + 6 + 12 * (size_t)i;
            if (u16(r + 6) != want) continue;
            const uint32_t platform = u16(r), language = u16(r + 4), len = u16(r + 8), off = strings + u16(r + 10);
            if ((size_t)off + len > d_.size()) continue;
            std::string s;
            if (platform == 0 || platform == 3) {  // UTF-16BE
                if (len & 1u) continue;
                bool ok = true;
                for (uint32_t k = 0; k + 1 < len; k += 2) {
                    uint32_t cp = ((uint32_t)d_[off + k] << 8) | d_[off + k + 1];
                    if (cp >= 0xd800 && cp <= 0xdbff && k + 3 < len) {
                        const uint32_t lo = ((uint32_t)d_[off + k + 2] << 8) | d_[off + k + 3];
                        if (lo >= 0xdc00 && lo <= 0xdfff) { cp = 0x10000 + ((cp - 0xd800) << 10) + (lo - 0xdc00); k += 2; }
                        else { ok = false; break; }
                    } else if (cp >= 0xd800 && cp <= 0xdfff) { ok = false; break; }
                    append_utf8(s, cp);
                }
                if (!ok) continue;
            } else if (platform == 1) {  // Mac Roman: ASCII as it is (the upper half does not occur in family names this code meets)
                for (uint32_t k = 0; k < len; k++) s += d_[off + k] < 0x80 ? (char)d_[off + k] : '?';
            } else {
                continue;
            }
            some = s; have_some = true;
            if ((platform == 1 && language == 0) || (platform == 3 && language == 0x409)) { english = s; break; }
        }
        const std::string &pick = !english.empty() ? english : some;
        if (!pick.empty() || (have_some && english.empty() && !some.empty())) { family = strip(pick); return; }
    }
}

// cmap.getBestCmap(): (3, 10), (0, 6), (0, 4), (3, 1), (0, 3), (0, 2), (0, 1), (0, 0) — the first that exists
void Font::choose_cmap() {
    const Table t = table(tag("cmap"));
    const uint32_t n = u16(t.off + 2);
    static const uint32_t PREF[][2] = {{3, 10}, {0, 6}, {0, 4}, {3, 1}, {0, 3}, {0, 2}, {0, 1}, {0, 0}};
    for (const auto &p : PREF)
        for (uint32_t i = 0; i < n; i++) {
            const size_t r = t.off + 4 + 8 * (size_t)i;
            if (u16(r) != p[0] || u16(r + 2) != p[1]) continue;
            const uint32_t off = t.off + u32(r + 4), fmt = u16(off);
            if (fmt == 0 || fmt == 4 || fmt == 6 || fmt == 12 || fmt == 13) {
                cmap_sub_.off = off; cmap_format_ = (int)fmt;
                return;
            }
        }
}

uint32_t Font::glyph_of(uint32_t cp) const {
    const size_t o = cmap_sub_.off;
    uint32_t gid = 0;
    switch (cmap_format_) {
    case 0:
        gid = cp < 256 ? u8(o + 6 + cp) : 0;
        break;
    case 4: {
        if (cp > 0xffff) break;
        const uint32_t segx2 = u16(o + 6), segs = segx2 / 2;
        const size_t ends = o + 14, starts = ends + segx2 + 2, deltas = starts + segx2, ranges = deltas + segx2;
        uint32_t lo = 0, hi = segs;  // first segment whose endCode >= cp
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (u16(ends + 2 * (size_t)mid) < cp) lo = mid + 1; else hi = mid;
        }
        if (lo >= segs || u16(starts + 2 * (size_t)lo) > cp) break;
        const uint32_t ro = u16(ranges + 2 * (size_t)lo), delta = u16(deltas + 2 * (size_t)lo);
        if (!ro) gid = (cp + delta) & 0xffffu;
        else {
            const uint32_t g = u16(ranges + 2 * (size_t)lo + ro + 2 * (size_t)(cp - u16(starts + 2 * (size_t)lo)));
            gid = g ? (g + delta) & 0xffffu : 0;
        }
        break;
    }
    case 6: {
        const uint32_t first = u16(o + 6), count = u16(o + 8);
        if (cp >= first && cp < first + count) gid = u16(o + 10 + 2 * (size_t)(cp - first));
        break;
    }
    case 12:
    case 13: {
        uint32_t groups = u32(o + 12);  // (no more groups than the file has bytes for: a corrupt count does not get to steer the search)
        const size_t room = d_.size() > o + 16 ? (d_.size() - o - 16) / 12 : 0;
        if ((size_t)groups > room) groups = (uint32_t)room;
        uint32_t lo = 0, hi = groups;  // first group whose endCharCode >= cp
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (u32(o + 16 + 12 * (size_t)mid + 4) < cp) lo = mid + 1; else hi = mid;
        }
        if (lo < groups && u32(o + 16 + 12 * (size_t)lo) <= cp)
            gid = cmap_format_ == 12 ? u32(o + 16 + 12 * (size_t)lo + 8) + (cp - u32(o + 16 + 12 * (size_t)lo)) : u32(o + 16 + 12 * (size_t)lo + 8);
        break;
    }
    }
    return gid < n_glyphs_ ? gid : 0;
}

double Font::advance(uint32_t gid) const {
    const Table t = table(tag("hmtx"));
    const uint32_t i = gid < n_hmetrics_ ? gid : n_hmetrics_ - 1;
    return (double)u16(t.off + 4 * (size_t)i);
}

int32_t Font::lsb(uint32_t gid) const {
    const Table t = table(tag("hmtx"));
    if (gid < n_hmetrics_) return i16(t.off + 4 * (size_t)gid + 2);
    return i16(t.off + 4 * (size_t)n_hmetrics_ + 2 * (size_t)(gid - n_hmetrics_));
}

bool Font::glyph_range(uint32_t gid, size_t &off, size_t &len) const {
    if (gid >= n_glyphs_) return false;
    const Table loca = table(tag("loca")), glyf = table(tag("glyf"));
    size_t a, b;
    if (loca_long_) { a = u32(loca.off + 4 * (size_t)gid); b = u32(loca.off + 4 * (size_t)gid + 4); }
    else { a = 2 * (size_t)u16(loca.off + 2 * (size_t)gid); b = 2 * (size_t)u16(loca.off + 2 * (size_t)gid + 2); }
    if (b <= a || b > glyf.len) return false;  // an empty glyph (a space)
    off = glyf.off + a; len = b - a;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// outlines: what fontTools' glyf pen protocol emits (ttLib/tables/_g_l_y_f.py Glyph.draw, pens/basePen.py DecomposingPen.addComponent,
// pens/transformPen.py) turned into closed polylines the way text_twin.py's Font.outline does
// ---------------------------------------------------------------------------------------------------------------------------------

void Font::draw(uint32_t gid, const std::vector<Affine> &chain, bool top_level, int depth, std::vector<Contour> &out) {
    size_t off, len;
    // (depth alone bounds nothing: composites that each name N composite children cost N^16 calls.  An outline gets a budget of components
    //  and points — far above any real glyph — and a font that exceeds it is a malformed font)
    if (depth > 16 || draw_calls_ > 4096u) return;
    if (++draw_calls_ > 4096u) return;
    if (!glyph_range(gid, off, len)) return;
    const int32_t n_contours = i16(off);
    if (n_contours < 0) {  // composite: every component through its transform, then through the transforms around it
        size_t p = off + 10;
        for (;;) {
            const uint32_t flags = u16(p), child = u16(p + 2);
            p += 4;
            Affine t;
            double ax = 0.0, ay = 0.0;
            if (flags & 0x0001u) {  // ARG_1_AND_2_ARE_WORDS
                if (flags & 0x0002u) { ax = (double)i16(p); ay = (double)i16(p + 2); }
                p += 4;
            } else {
                if (flags & 0x0002u) { ax = (double)(int8_t)u8(p); ay = (double)(int8_t)u8(p + 1); }
                p += 2;
            }
            // (point-matching components — ARGS_ARE_XY_VALUES clear — carry no offset here, as in fontTools' getComponentInfo)
            if (flags & 0x0008u) { t.xx = t.yy = (double)i16(p) / 16384.0; p += 2; t.identity = false; }
            else if (flags & 0x0040u) { t.xx = (double)i16(p) / 16384.0; t.yy = (double)i16(p + 2) / 16384.0; p += 4; t.identity = false; }
            else if (flags & 0x0080u) {
                t.xx = (double)i16(p) / 16384.0; t.xy = (double)i16(p + 2) / 16384.0; t.yx = (double)i16(p + 4) / 16384.0; t.yy = (double)i16(p + 6) / 16384.0;
                p += 8; t.identity = false;
            }
            t.dx = ax; t.dy = ay;
            if (ax != 0.0 || ay != 0.0) t.identity = false;
            std::vector<Affine> inner = chain;
            if (!t.identity) inner.push_back(t);
            draw(child, inner, false, depth + 1, out);
            if (!(flags & 0x0020u) || bad_ || draw_calls_ > 4096u) break;  // MORE_COMPONENTS
        }
        return;
    }
    if (n_contours == 0) return;
    // ---- simple glyph: end points, flags (with repeats), x then y deltas
    std::vector<uint32_t> ends((size_t)n_contours);
    size_t p = off + 10;
    for (int32_t i = 0; i < n_contours; i++, p += 2) ends[(size_t)i] = u16(p);
    const uint32_t n_pts = ends.back() + 1;
    if (n_pts > 65535u) return;
    p += 2 + u16(p);  // instructions
    std::vector<uint8_t> fl(n_pts);
    for (uint32_t i = 0; i < n_pts;) {
        const uint32_t f = u8(p++);
        fl[i++] = (uint8_t)f;
        if (f & 0x08u) {
            uint32_t rep = u8(p++);
            while (rep-- && i < n_pts) fl[i++] = (uint8_t)f;
        }
        if (bad_) return;
    }
    std::vector<double> xs(n_pts), ys(n_pts);
    int32_t v = 0;
    for (uint32_t i = 0; i < n_pts; i++) {
        if (fl[i] & 0x02u) { const int32_t d = (int32_t)u8(p++); v += (fl[i] & 0x10u) ? d : -d; }
        else if (!(fl[i] & 0x10u)) { v += i16(p); p += 2; }
        xs[i] = (double)v;
    }
    v = 0;
    for (uint32_t i = 0; i < n_pts; i++) {
        if (fl[i] & 0x04u) { const int32_t d = (int32_t)u8(p++); v += (fl[i] & 0x20u) ? d : -d; }
        else if (!(fl[i] & 0x20u)) { v += i16(p); p += 2; }
        ys[i] = (double)v;
    }
    if (bad_) return;
    // _TTGlyphGlyf.draw: a top-level glyph is shifted by lsb - xMin (integers), components are not
    const double shift = top_level ? (double)(lsb(gid) - i16(off + 2)) : 0.0;
    std::vector<Pt> pts(n_pts);
    for (uint32_t i = 0; i < n_pts; i++) {
        double x = xs[i] + shift, y = ys[i];
        for (size_t k = chain.size(); k-- > 0;) {  // TransformPen nesting: the innermost component's transform first
            const Affine &t = chain[k];
            const double nx = t.xx * x + t.yx * y + t.dx, ny = t.xy * x + t.yy * y + t.dy;
            x = nx; y = ny;
        }
        pts[i] = Pt{x, y};
    }
    auto curve = [](Contour &cur, const Pt &c, const Pt &p1) {  // one quadratic segment as 8 chords (text_twin.py)
        const Pt p0 = cur.back();
        for (int k = 1; k < 9; k++) {
            const double t = (double)k / 8.0;
            const double a = (1 - t) * (1 - t), b = 2 * t * (1 - t), d = t * t;
            cur.push_back(Pt{a * p0.x + b * c.x + d * p1.x, a * p0.y + b * c.y + d * p1.y});
        }
    };
    // decomposeQuadraticSegment: off-curve points ... on-curve point, implied on-curve points halfway between consecutive off-curve ones
    auto qcurve = [&](Contour &cur, const std::vector<Pt> &q) {
        const size_t n = q.size() - 1;
        for (size_t i = 0; i + 1 < n; i++) curve(cur, q[i], Pt{0.5 * (q[i].x + q[i + 1].x), 0.5 * (q[i].y + q[i + 1].y)});
        curve(cur, q[n - 1], q[n]);
    };
    uint32_t start = 0;
    for (int32_t ci = 0; ci < n_contours; ci++) {
        const uint32_t end = ends[(size_t)ci] + 1;
        if (end <= start || end > n_pts) return;
        std::vector<Pt> c(pts.begin() + start, pts.begin() + end);
        std::vector<uint8_t> on(end - start);
        for (uint32_t i = start; i < end; i++) on[i - start] = fl[i] & 1u;
        start = end;
        Contour cur;
        const auto first_on = std::find(on.begin(), on.end(), (uint8_t)1);
        if (first_on == on.end()) {  // off-curve points only: start halfway between the last and the first
            const Pt s{(c.back().x + c.front().x) / 2.0, (c.back().y + c.front().y) / 2.0};
            cur.push_back(s);
            std::vector<Pt> q = c;
            q.push_back(s);
            qcurve(cur, q);
        } else {
            // rotated so that the contour ENDS in its first on-curve point, which starts the path
            const size_t k = (size_t)(first_on - on.begin()) + 1;
            std::rotate(c.begin(), c.begin() + (std::ptrdiff_t)k, c.end());
            std::rotate(on.begin(), on.begin() + (std::ptrdiff_t)k, on.end());
            cur.push_back(c.back());
            size_t at = 0;
            while (at < c.size()) {
                size_t next = at;
                while (!on[next]) next++;
                next++;  // one past the segment's on-curve point
                if (next - at == 1) {
                    if (c.size() - at > 1) cur.push_back(c[at]);  // (the final lineTo is implied by closePath)
                } else {
                    qcurve(cur, std::vector<Pt>(c.begin() + (std::ptrdiff_t)at, c.begin() + (std::ptrdiff_t)next));
                }
                at = next;
            }
        }
        if (cur.size() > 1) out.push_back(std::move(cur));
    }
}

const std::vector<Contour> &Font::outline(uint32_t gid) {
    auto it = outlines_.find(gid);
    if (it != outlines_.end()) return it->second;
    std::vector<Contour> out;
    draw_calls_ = 0;
    draw(gid, {}, true, 0, out);
    if (draw_calls_ > 4096u) out.clear();  // (budget exceeded: no outline rather than a fraction of an absurd one)
    return outlines_.emplace(gid, std::move(out)).first->second;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// GPOS `kern`: the PairPos subtables (lookup type 2, through extension lookups) of the feature, default script first
// ---------------------------------------------------------------------------------------------------------------------------------

void Font::parse_gpos_kern() {
    if (!has(tag("GPOS"))) return;
    const size_t g = table(tag("GPOS")).off;
    const size_t scripts = g + u16(g + 4), features = g + u16(g + 6), lookups = g + u16(g + 8);
    if (features == g || lookups == g) return;
    const uint32_t n_features = u16(features);
    // ScriptList: latn, else DFLT; its DefaultLangSys's feature indices, else every feature
    std::vector<uint32_t> feature_indices;
    bool have_script = false;
    if (scripts != g) {
        size_t latn = 0, dflt = 0;
        const uint32_t ns = u16(scripts);
        for (uint32_t i = 0; i < ns; i++) {
            const uint32_t t = u32(scripts + 2 + 6 * (size_t)i);
            const size_t o = scripts + u16(scripts + 2 + 6 * (size_t)i + 4);
            if (t == tag("latn")) latn = o;
            if (t == tag("DFLT")) dflt = o;
        }
        const size_t script = latn ? latn : dflt;
        if (script && u16(script)) {
            const size_t ls = script + u16(script);
            const uint32_t n = u16(ls + 4);
            for (uint32_t i = 0; i < n; i++) feature_indices.push_back(u16(ls + 6 + 2 * (size_t)i));
            have_script = true;
        }
    }
    if (!have_script)
        for (uint32_t i = 0; i < n_features; i++) feature_indices.push_back(i);
    std::vector<uint32_t> wanted;
    for (uint32_t fi : feature_indices) {
        if (fi >= n_features) continue;
        const size_t rec = features + 2 + 6 * (size_t)fi;
        if (u32(rec) != tag("kern")) continue;
        const size_t f = features + u16(rec + 4);
        const uint32_t n = u16(f + 2);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t li = u16(f + 4 + 2 * (size_t)i);
            if (std::find(wanted.begin(), wanted.end(), li) == wanted.end()) wanted.push_back(li);
        }
    }
    std::sort(wanted.begin(), wanted.end());  // (lookups apply in lookup-list order)
    const uint32_t n_lookups = u16(lookups);
    for (uint32_t li : wanted) {
        if (li >= n_lookups) continue;
        const size_t l = lookups + u16(lookups + 2 + 2 * (size_t)li);
        const uint32_t type = u16(l), n = u16(l + 4);
        std::vector<uint32_t> subs;
        for (uint32_t i = 0; i < n; i++) {
            size_t st = l + u16(l + 6 + 2 * (size_t)i);
            uint32_t t = type;
            if (t == 9) { t = u16(st + 2); st = st + u32(st + 4); }
            if (t == 2) subs.push_back((uint32_t)st);
        }
        if (!subs.empty()) kern_lookups_.push_back(std::move(subs));
    }
}

int Font::coverage_index(size_t cov, uint32_t gid) const {
    const uint32_t fmt = u16(cov), n = u16(cov + 2);
    if (fmt == 1) {
        for (uint32_t i = 0; i < n; i++)
            if (u16(cov + 4 + 2 * (size_t)i) == gid) return (int)i;
        return -1;
    }
    uint32_t index = 0;  // format 2: the glyph list the ranges expand to, in range order (65 535 ranges of 65 536 glyphs overflow an int: unsigned, and capped)
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a = u16(cov + 4 + 6 * (size_t)i), b = u16(cov + 4 + 6 * (size_t)i + 2);
        if (gid >= a && gid <= b) return index + (gid - a) < 0x7fffffffu ? (int)(index + (gid - a)) : -1;
        if (b >= a) index = index + (b - a + 1) < 0x7fffffffu ? index + (b - a + 1) : 0x7fffffffu;
    }
    return -1;
}

uint32_t Font::class_of(size_t cd, uint32_t gid) const {
    const uint32_t fmt = u16(cd);
    if (fmt == 1) {
        const uint32_t first = u16(cd + 2), n = u16(cd + 4);
        return gid >= first && gid < first + n ? u16(cd + 6 + 2 * (size_t)(gid - first)) : 0;
    }
    const uint32_t n = u16(cd + 2);
    uint32_t cls = 0;  // (later ranges win, as a dict built range by range)
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a = u16(cd + 4 + 6 * (size_t)i), b = u16(cd + 4 + 6 * (size_t)i + 2);
        if (gid >= a && gid <= b) cls = u16(cd + 4 + 6 * (size_t)i + 4);
    }
    return cls;
}

double Font::kerning(uint32_t left, uint32_t right) {
    const uint64_t key = ((uint64_t)left << 32) | right;
    auto it = kern_cache_.find(key);
    if (it != kern_cache_.end()) return it->second;
    double total = 0.0;
    for (const auto &subs : kern_lookups_)
        for (uint32_t st : subs) {  // the first subtable of a lookup that covers the pair decides
            const int ci = coverage_index(st + u16(st + 2), left);
            if (ci < 0) continue;
            const uint32_t fmt = u16(st), vf1 = u16(st + 4), vf2 = u16(st + 6);
            const size_t rec_values = 2 * (size_t)(popcount16(vf1) + popcount16(vf2));
            size_t value = 0;
            bool found = false;
            if (fmt == 1) {
                if ((uint32_t)ci >= u16(st + 8)) continue;
                const size_t ps = st + u16(st + 10 + 2 * (size_t)ci);
                const uint32_t n = u16(ps);
                for (uint32_t i = 0; i < n; i++) {
                    const size_t r = ps + 2 + (2 + rec_values) * (size_t)i;
                    if (u16(r) == right) { value = r + 2; found = true; break; }
                }
                if (!found) continue;
            } else {
                const uint32_t c1n = u16(st + 12), c2n = u16(st + 14);
                const uint32_t c1 = class_of(st + u16(st + 8), left), c2 = class_of(st + u16(st + 10), right);
                if (c1 < c1n && c2 < c2n) { value = st + 16 + rec_values * ((size_t)c1 * c2n + c2); found = true; }
            }
            if (found && (vf1 & 0x0004u)) total += (double)i16(value + 2 * (size_t)popcount16(vf1 & 0x0003u));
            break;
        }
    kern_cache_[key] = total;
    return total;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// font book
// ---------------------------------------------------------------------------------------------------------------------------------

bool FontBook::add_memory(std::vector<uint8_t> data, std::string &err) {
    std::unique_ptr<Font> f = Font::load(std::move(data), err);
    if (!f) return false;
    fonts_.push_back(std::move(f));
    return true;
}

bool FontBook::add_file(const std::string &path, std::string &err) {
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { err = "cannot open font file " + path; return false; }
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(fp);
    std::string e;
    if (!add_memory(std::move(data), e)) { err = path + ": " + e; return false; }
    return true;
}

static void walk_ttf(const std::string &dir, std::vector<std::string> &out, int depth) {
    DIR *d = depth < 16 ? opendir(dir.c_str()) : nullptr;
    if (!d) return;
    while (dirent *e = readdir(d)) {
        const std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        const std::string path = dir + "/" + name;
        struct stat st;
        if (stat(path.c_str(), &st) != 0) continue;
        if (S_ISDIR(st.st_mode)) walk_ttf(path, out, depth + 1);
        else if (name.size() >= 4 && lower(name.substr(name.size() - 4)) == ".ttf") out.push_back(path);
    }
    closedir(d);
}

int FontBook::add_dir(const std::string &dir, std::string &err) {
    std::string root = dir;
    while (root.size() > 1 && root.back() == '/') root.pop_back();
    std::vector<std::string> paths;
    walk_ttf(root, paths, 0);
    std::sort(paths.begin(), paths.end());
    int added = 0;
    for (const std::string &p : paths) {
        std::string e;
        if (add_file(p, e)) added++;
        else err = e;  // (a face this reader cannot take is skipped: the last reason is kept)
    }
    return added;
}

Font *FontBook::match(const std::string &family, const std::string &weight, const std::string &style) {
    if (fonts_.empty()) return nullptr;
    const int want_w = weight_of(weight);
    const bool want_i = style == "Italic" || style == "Oblique";
    const std::string fam = lower(family);
    std::vector<Font *> pool;
    for (auto &f : fonts_)
        if (lower(f->family) == fam) pool.push_back(f.get());
    if (pool.empty())  // unknown family: any face (fallback)
        for (auto &f : fonts_) pool.push_back(f.get());
    Font *best = nullptr;
    for (Font *f : pool) {
        const bool bi = best && (best->italic != want_i), fi = f->italic != want_i;
        if (!best || fi < bi || (fi == bi && std::abs(f->weight - want_w) < std::abs(best->weight - want_w))) best = f;
    }
    return best;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// layout
// ---------------------------------------------------------------------------------------------------------------------------------

static std::vector<Line> layout_cp(Font &font, const std::vector<uint32_t> &cps, double font_size, const std::string &wrap, double max_width, bool kerning) {
    const double scale = font_size / font.upem;
    const uint32_t space = font.glyph_of(' ');
    std::vector<Line> lines;
    size_t at = 0;
    for (;;) {  // text.split("\n")
        size_t end = at;
        while (end < cps.size() && cps[end] != '\n') end++;
        std::vector<LineGlyph> cur;
        double x = 0.0;
        long last_space = -1;  // index in cur after which a Word wrap may break
        for (size_t i = at; i < end; i++) {
            const uint32_t ch = cps[i];
            const uint32_t g = font.glyph_of(ch);
            if (!cur.empty() && kerning) x += font.kerning(cur.back().gid, g) * scale;  // the previous glyph's advance, adjusted for this pair
            const double adv = font.advance(g) * scale;
            if (wrap != "None" && !cur.empty() && x + adv > max_width && !is_space(ch)) {
                if (wrap == "Word" && last_space >= 0) {
                    std::vector<LineGlyph> head(cur.begin(), cur.begin() + last_space + 1), tail(cur.begin() + last_space + 1, cur.end());
                    while (!head.empty() && head.back().gid == space) head.pop_back();  // the break swallows the trailing space
                    Line l;
                    l.width = head.empty() ? 0.0 : head.back().x + font.advance(head.back().gid) * scale;
                    l.glyphs = std::move(head);
                    lines.push_back(std::move(l));
                    const double shift = tail.empty() ? x : tail.front().x;
                    for (LineGlyph &t : tail) t.x = t.x - shift;
                    cur = std::move(tail);
                    x -= shift;
                } else {
                    Line l;
                    l.glyphs = std::move(cur); l.width = x;
                    lines.push_back(std::move(l));
                    cur.clear();
                    x = 0.0;
                }
                last_space = -1;
            }
            cur.push_back(LineGlyph{g, x});
            x += adv;
            if (is_space(ch)) last_space = (long)cur.size() - 1;
        }
        Line l;
        l.glyphs = std::move(cur); l.width = x;
        lines.push_back(std::move(l));
        if (end >= cps.size()) break;
        at = end + 1;
    }
    return lines;
}

std::vector<Line> layout(Font &font, const std::string &utf8, double font_size, const std::string &wrap, double max_width, bool kerning) {
    std::vector<uint32_t> cps;
    if (!decode_utf8(utf8, cps)) return {};
    return layout_cp(font, cps, font_size, wrap, max_width, kerning);
}

// what a caller may hand in: a finite font size in (0, 1e5] (the scene front-end only lets positive sizes through; 1e5 pixels is an order of
// magnitude beyond MAX_NODE_RESOLUTION), a finite line height
static bool sane_sizes(const smr_text_params &p, std::string &err) {
    if (!(p.font_size > 0.0f && p.font_size <= 1e5f)) { err = "font_size must be a finite number in (0, 100000]"; return false; }
    if (!(p.line_height == p.line_height) || std::fabs(p.line_height) > 1e6f) { err = "line_height must be a finite number"; return false; }
    return true;
}

bool measure(FontBook &book, const smr_text_params &p, float &widest, uint32_t &count, std::string &err) {
    if (!sane_sizes(p, err)) return false;
    Font *font = book.match(p.font_family ? p.font_family : "", p.weight ? p.weight : "Normal", p.style ? p.style : "Normal");
    if (!font) { err = "the font book is empty"; return false; }
    std::vector<uint32_t> cps;
    if (!decode_utf8(p.text ? p.text : "", cps)) { err = "text is not valid UTF-8"; return false; }
    const std::vector<Line> lines = layout_cp(*font, cps, (double)p.font_size, p.wrap ? p.wrap : "None", (double)p.max_width, true);
    double w = 0.0;
    for (const Line &l : lines) w = l.width > w ? l.width : w;
    widest = (float)w;
    count = (uint32_t)lines.size();
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// rasteriser: signed-area accumulation — every edge adds, to each pixel row it crosses, the area it sweeps to its right; a running sum
// along the row turns the deltas into coverage (non-zero winding for outlines that do not self-overlap)
// ---------------------------------------------------------------------------------------------------------------------------------

void accumulate_edge(std::vector<double> &a, int w, int h, double x0, double y0, double x1, double y1) {  // (not static: tests/san/host_fuzz.cpp --edges drives it directly)
    if (y0 == y1) return;
    double d = 1.0;
    if (y0 > y1) { std::swap(x0, x1); std::swap(y0, y1); d = -1.0; }
    const double dxdy = (x1 - x0) / (y1 - y0);
    double x = x0;
    // x advances by dxdy * dy row by row: rounding can carry it a hair past the edge's own end (an edge that ends exactly on the bitmap's left
    // column arrives at -1e-17: floor -> column -1, eight bytes in front of the accumulator).  An edge never leaves its own x range.
    const double x_lo = std::min(x0, x1), x_hi = std::max(x0, x1);
    const int ya = std::max((int)std::floor(y0), 0), yb = std::min(h, (int)std::ceil(y1));
    for (int y = ya; y < yb; y++) {
        const double dy = std::min((double)y + 1.0, y1) - std::max((double)y, y0);
        const double xn = std::min(std::max(x + dxdy * dy, x_lo), x_hi);
        const double s = d * dy;
        const double xa = x < xn ? x : xn, xb = x < xn ? xn : x;
        const int ia = (int)std::floor(xa), ib = (int)std::ceil(xb);
        const size_t row = (size_t)y * (size_t)w;
        if (ib <= ia + 1) {  // the edge stays inside one pixel column
            const double xm = 0.5 * (x + xn) - ia;
            a[row + ia] += s - s * xm;
            a[row + ia + 1] += s * xm;
        } else {
            const double inv = 1.0 / (xb - xa);
            const double fa = xa - ia;
            const double a0 = 0.5 * inv * (1.0 - fa) * (1.0 - fa);
            const double fb = xb - ib + 1.0;
            const double am = 0.5 * inv * fb * fb;
            a[row + ia] += s * a0;
            if (ib == ia + 2) {
                a[row + ia + 1] += s * (1.0 - a0 - am);
            } else {
                const double a1 = inv * (1.5 - fa);
                a[row + ia + 1] += s * (a1 - a0);
                for (int xi = ia + 2; xi < ib - 1; xi++) a[row + xi] += s * inv;
                const double a2 = a1 + (ib - ia - 3) * inv;
                a[row + ib - 1] += s * (1.0 - a2 - am);
            }
            a[row + ib] += s * am;
        }
        x = xn;
    }
}

GlyphBitmap rasterise_glyph(Font &font, uint32_t gid, double scale, double fx, double fy) {
    GlyphBitmap out;
    const std::vector<Contour> &contours = font.outline(gid);
    if (contours.empty()) return out;
    std::vector<Contour> pts(contours.size());
    double min_x = std::numeric_limits<double>::infinity(), max_x = -min_x, min_y = min_x, max_y = -min_x;
    for (size_t i = 0; i < contours.size(); i++) {
        pts[i].reserve(contours[i].size());
        for (const Pt &p : contours[i]) {  // pixels, y down, relative to the pen's pixel
            const Pt q{p.x * scale + fx, -p.y * scale + fy};
            pts[i].push_back(q);
            min_x = std::min(min_x, q.x); max_x = std::max(max_x, q.x);
            min_y = std::min(min_y, q.y); max_y = std::max(max_y, q.y);
        }
    }
    // (a size no node can hold: MAX_NODE_RESOLUTION is 7682 x 4320; NaN; contours without a point leave the extents at +-infinity)
    if (!(min_x <= max_x && min_y <= max_y && min_x > -1e6 && max_x < 1e6 && min_y > -1e6 && max_y < 1e6)) return out;
    const int left = (int)std::floor(min_x), top = (int)std::floor(min_y);
    const int w = (int)std::ceil(max_x) - left + 1, h = (int)std::ceil(max_y) - top + 1;
    if (w <= 0 || h <= 0 || (long long)w * h > (1ll << 24)) return out;  // (a 4 000 pixel glyph is 16 M pixels: nothing a node shows is larger)
    std::vector<double> acc((size_t)w * h + 4, 0.0);
    for (const Contour &c : pts) {
        const size_t n = c.size();
        for (size_t i = 0; i < n; i++) {
            const Pt &p = c[i], &q = c[(i + 1) % n];
            accumulate_edge(acc, w, h, p.x - left, p.y - top, q.x - left, q.y - top);
        }
    }
    out.px.resize((size_t)w * h);
    double run = 0.0;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        run += acc[i];
        double cov = std::fabs(run);
        cov = cov < 0.0 ? 0.0 : (cov > 1.0 ? 1.0 : cov);
        out.px[i] = (uint8_t)(cov * 255.0 + 0.5);
    }
    out.w = w; out.h = h; out.left = left; out.top = -top;
    return out;
}

bool rasterise(FontBook &book, const smr_text_params &p, uint32_t width, uint32_t height, const float color[4], TextRun &out, std::string &err) {
    out.glyphs.clear(); out.atlas.clear(); out.atlas_w = out.atlas_h = 0;
    if (!sane_sizes(p, err)) return false;
    Font *font = book.match(p.font_family ? p.font_family : "", p.weight ? p.weight : "Normal", p.style ? p.style : "Normal");
    if (!font) { err = "the font book is empty"; return false; }
    std::vector<uint32_t> cps;
    if (!decode_utf8(p.text ? p.text : "", cps)) { err = "text is not valid UTF-8"; return false; }
    const double font_size = (double)p.font_size;
    const double line_height = p.line_height > 0.0f ? (double)p.line_height : font_size;
    const double scale = font_size / font->upem;
    const std::vector<Line> lines = layout_cp(*font, cps, font_size, p.wrap ? p.wrap : "None", (double)width, true);
    const double asc = font->ascent * scale, desc = font->descent * scale;
    const std::string align = p.align ? p.align : "Left";
    struct Placed {
        uint32_t gid;
        double x, base;
    };
    std::vector<Placed> placed;
    for (size_t li = 0; li < lines.size(); li++) {
        const double free_w = (double)width - lines[li].width;
        const double x0 = align == "Center" ? free_w / 2.0 : align == "Right" ? free_w : 0.0;
        const double base = (double)li * line_height + (line_height - (asc + desc)) / 2.0 + asc;
        for (const LineGlyph &g : lines[li].glyphs) placed.push_back(Placed{g.gid, x0 + g.x, base});
    }
    // every (glyph, fractional offset to three decimals) rasterised once, at the offset of its first occurrence
    struct Cell {
        std::string key;
        GlyphBitmap bmp;
        int ax = 0, ay = 0;
    };
    std::vector<Cell> cells;  // in order of first use
    std::unordered_map<std::string, size_t> index;
    struct Use {
        size_t cell;
        int dx, dy;
    };
    std::vector<Use> order;
    for (const Placed &g : placed) {
        if (!(std::fabs(g.x) < 1e9 && std::fabs(g.base) < 1e9)) continue;  // (far outside any node; keeps the pen's pixel an int)
        const double fx = g.x - std::floor(g.x), fy = g.base - std::floor(g.base);
        char key[96];
        snprintf(key, sizeof(key), "%u/%.3f/%.3f", g.gid, fx, fy);  // round(f, 3): the correctly rounded three-decimal value
        auto it = index.find(key);
        if (it == index.end()) {
            Cell c;
            c.key = key;
            c.bmp = rasterise_glyph(*font, g.gid, scale, fx, fy);
            it = index.emplace(key, cells.size()).first;
            cells.push_back(std::move(c));
        }
        const GlyphBitmap &b = cells[it->second].bmp;
        if (!b.px.empty()) order.push_back(Use{it->second, (int)std::floor(g.x) + b.left, (int)std::floor(g.base) - b.top});
    }
    // the bitmaps row by row into one atlas, at least 256 wide
    int aw = 1;
    for (const Cell &c : cells)
        if (!c.bmp.px.empty()) aw = std::max(aw, c.bmp.w);
    aw = std::max(aw, 256);
    int cx = 0, cy = 0, rowh = 0;
    for (Cell &c : cells) {
        if (c.bmp.px.empty()) continue;
        if (cx + c.bmp.w > aw) { cx = 0; cy = cy + rowh; rowh = 0; }
        c.ax = cx; c.ay = cy;
        cx += c.bmp.w;
        rowh = std::max(rowh, c.bmp.h);
    }
    const int ah = std::max(cy + rowh, 1);
    out.atlas.assign((size_t)aw * ah, 0);
    out.atlas_w = (uint32_t)aw; out.atlas_h = (uint32_t)ah;
    for (const Cell &c : cells)
        for (int y = 0; y < c.bmp.h && !c.bmp.px.empty(); y++) memcpy(&out.atlas[(size_t)(c.ay + y) * aw + c.ax], &c.bmp.px[(size_t)y * c.bmp.w], (size_t)c.bmp.w);
    for (const Use &u : order) {
        const Cell &c = cells[u.cell];
        // clip to the node (smr_blit_glyphs wants quads inside the target)
        const int x0 = std::max(u.dx, 0), y0 = std::max(u.dy, 0), x1 = std::min(u.dx + c.bmp.w, (int)width), y1 = std::min(u.dy + c.bmp.h, (int)height);
        if (x1 > x0 && y1 > y0) {
            smr_glyph g;
            g.dst_x = x0; g.dst_y = y0; g.w = x1 - x0; g.h = y1 - y0;
            g.atlas_x = c.ax + (x0 - u.dx); g.atlas_y = c.ay + (y0 - u.dy);
            for (int k = 0; k < 4; k++) g.color[k] = color[k];
            out.glyphs.push_back(g);
        }
    }
    return true;
}

}  // namespace smr_text
