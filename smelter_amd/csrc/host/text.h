// text.h — the host half of a12 behind the C ABI: font database, line layout, glyph rasteriser.
//
// The reference lays text out and rasterises it inside the renderer (smelter-render/src/transformations/text_renderer.rs:72-167,
// 282-368) with glyphon / cosmic-text / swash — third-party crates that are not in the reference tree.  What this file restates is the
// part of that pipeline the reference's own code determines — the font database (`TextRendererCtx::new / add_font`: bundled faces +
// registered ones, matched by family / weight / style), the line model as far as the reference drives it (explicit newlines,
// `Wrap::None | Glyph | Word` against the buffer width, `Align`), the sizing rule (`get_text_resolution`) and the placement of a line's
// baseline inside its line box — over a TrueType reader of its own: `cmap` (formats 4, 6, 12), `hmtx`, `glyf` outlines (simple and
// composite), pair kerning from the GPOS `kern` feature (PairPos formats 1 and 2, through extension lookups), an exact-area coverage
// rasteriser.  NOT restated (stated in include/smr.h too): GSUB ligatures / contextual alternates, mark positioning, bidi, font
// fallback, hinting — glyph SHAPES and sub-pixel positions are this file's, not glyphon's; parity for text pixels stays unpinned.
//
// The C++ here is the product path; tests/text_twin.py is its Python twin (fontTools-based) and tests/test_text_capi.py holds the two
// to each other byte for byte (glyph runs and atlases).  Everything is double precision in the order text_twin.py evaluates it.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "smr.h"

namespace smr_text {

struct Pt {
    double x, y;
};
using Contour = std::vector<Pt>;

class Font {
public:
    // nullptr + err on anything that is not a TrueType-outline sfnt this reader understands
    static std::unique_ptr<Font> load(std::vector<uint8_t> data, std::string &err);

    double upem = 1000.0, ascent = 0.0, descent = 0.0;  // hhea ascent / -descent: both positive, font units
    std::string family;                                 // name ID 16, else 1 (English record first)
    int weight = 400;                                   // OS/2 usWeightClass
    bool italic = false;                                // OS/2 fsSelection bit 0

    uint32_t glyph_of(uint32_t codepoint) const;        // 0 (.notdef) when the cmap has no entry
    double advance(uint32_t gid) const;                 // hmtx, font units
    double kerning(uint32_t left, uint32_t right);      // x-advance adjustment of `left` followed by `right`, font units
    const std::vector<Contour> &outline(uint32_t gid);  // closed polylines, font units, y up: quadratic segments as 8 chords each

private:
    std::vector<uint8_t> d_;
    struct Table {
        uint32_t off = 0, len = 0;
    };
    std::unordered_map<uint32_t, Table> tables_;
    uint32_t n_glyphs_ = 0, n_hmetrics_ = 0;
    int loca_long_ = 0;
    Table cmap_sub_;  // the chosen cmap subtable
    int cmap_format_ = 0;
    std::unordered_map<uint32_t, std::vector<Contour>> outlines_;
    std::unordered_map<uint64_t, double> kern_cache_;
    std::vector<std::vector<uint32_t>> kern_lookups_;  // per `kern` lookup (lookup-list order): absolute offsets of its PairPos subtables

    bool parse(std::string &err);
    bool has(uint32_t tag) const { return tables_.count(tag) != 0; }
    Table table(uint32_t tag) const;
    // bounds-checked big-endian reads: out of range reads give 0 and set bad_
    mutable bool bad_ = false;
    uint32_t draw_calls_ = 0;  // components visited by the outline being drawn (Font::draw's budget)
    uint32_t u8(size_t o) const;
    uint32_t u16(size_t o) const;
    int32_t i16(size_t o) const;
    uint32_t u32(size_t o) const;
    int32_t lsb(uint32_t gid) const;
    bool glyph_range(uint32_t gid, size_t &off, size_t &len) const;
    void parse_name();
    void choose_cmap();
    void parse_gpos_kern();
    int coverage_index(size_t cov, uint32_t gid) const;  // -1 = not covered
    uint32_t class_of(size_t classdef, uint32_t gid) const;
    // pen protocol of fontTools' glyf drawing, with the component transforms of DecomposingRecordingPen on top (see text.cpp)
    struct Affine {
        double xx = 1, xy = 0, yx = 0, yy = 1, dx = 0, dy = 0;
        bool identity = true;
    };
    void draw(uint32_t gid, const std::vector<Affine> &chain, bool top_level, int depth, std::vector<Contour> &out);
};

class FontBook {
public:
    bool add_memory(std::vector<uint8_t> data, std::string &err);
    bool add_file(const std::string &path, std::string &err);
    int add_dir(const std::string &dir, std::string &err);  // every *.ttf below dir (sorted by path); number added
    size_t size() const { return fonts_.size(); }
    // fontdb's matching, reduced: faces of the family (any face when the family is unknown), then the requested slant, then the closest weight
    Font *match(const std::string &family, const std::string &weight, const std::string &style);

private:
    std::vector<std::unique_ptr<Font>> fonts_;
};

struct LineGlyph {
    uint32_t gid;
    double x;  // pen x of its origin, pixels
};
struct Line {
    std::vector<LineGlyph> glyphs;
    double width = 0.0;
};

// Buffer::set_text + set_wrap + shape_until_scroll as far as widths and line breaks go (one Line per LayoutLine).
// wrap: "None" | "Glyph" | "Word"; max_width in pixels (infinity = no limit)
std::vector<Line> layout(Font &font, const std::string &utf8, double font_size, const std::string &wrap, double max_width, bool kerning = true);

struct GlyphBitmap {
    std::vector<uint8_t> px;  // h rows of w coverage bytes
    int w = 0, h = 0, left = 0, top = 0;  // the bitmap's top-left pixel sits `left` right of and `top` above the pen's pixel
};
// Exact-area coverage of one glyph at `scale` pixels per font unit, origin at the fractional pixel offset (fx, fy) of its cell.
GlyphBitmap rasterise_glyph(Font &font, uint32_t gid, double scale, double fx, double fy);
// one edge into the signed-area accumulator of a w x h bitmap (w * h + 1 doubles; x in [0, w - 1], any y): internal, declared for the sanitizer harness
void accumulate_edge(std::vector<double> &a, int w, int h, double x0, double y0, double x1, double y1);

struct TextRun {
    std::vector<smr_glyph> glyphs;
    std::vector<uint8_t> atlas;
    uint32_t atlas_w = 0, atlas_h = 0;
};
// The glyph run of one Text node of width x height pixels (what TextRendererNode::render draws): lines `line_height` apart, a line's
// ascent + descent centred in its line box, glyph bitmaps rasterised at their fractional pen position, quads clipped to the node.
bool rasterise(FontBook &book, const smr_text_params &p, uint32_t width, uint32_t height, const float color[4], TextRun &out, std::string &err);
bool measure(FontBook &book, const smr_text_params &p, float &widest, uint32_t &lines, std::string &err);

}  // namespace smr_text

struct smr_fontbook {
    smr_text::FontBook book;
    smr_text::TextRun run;  // the last smr_fontbook_rasterise result (what smr_text_run points into)
    std::string error;
};
