"""a12 host side: the reference's text sizing rule in the scene engine (text_renderer.rs:282-368, through the C ABI's
smr_text_measure_fn) and the bundled TrueType shaper / rasteriser of tests/text_twin.py.  No GPU."""
import math
import os

import numpy as np
import pytest

from smelter_amd import _ffi
from tests import text_twin as T
from smelter_amd.scene import Scene, SceneError

REF_FONTS = "/root/reference/smelter-render/fonts"          # Inter, bundled by the reference (only in the build container)
SYS_FONTS = "/usr/share/fonts/truetype/dejavu"              # part of the image, on the GPU box too


def _book():
    for d in (REF_FONTS, SYS_FONTS):
        if os.path.isdir(d):
            return T.FontBook.from_dir(d)
    pytest.skip("no TrueType fonts on this machine")


def _fixed_measurer(widest, lines, seen=None):
    def fn(_user, params, w, n):
        if seen is not None:
            p = params.contents
            seen.append((p.text.decode(), p.wrap.decode(), p.font_size, p.line_height, p.max_width, p.max_height, p.font_family.decode(),
                         p.weight.decode(), p.style.decode(), p.align.decode()))
        w[0], n[0] = widest, lines
        return 0
    return _ffi.TEXT_MEASURE_FN(fn)


def _text_node(scene_json, measurer, W=640, H=360):
    s = Scene()
    try:
        if measurer is not None:
            s.set_text_measurer(measurer)
        nodes = s.update(scene_json, W, H)
        return [n for n in nodes if n.kind == _ffi.NODE_TEXT][0]
    finally:
        s.close()


def test_fitted_text_uses_the_reference_sizing_rule():
    """get_text_resolution: width = max ceil(line width), height = trunc(lines * ceil(line_height) + font_size / 5)."""
    seen = []
    n = _text_node({"type": "text", "text": "two\nlines", "font_size": 33.0, "line_height": 40.5, "wrap": "word", "max_width": 500.0},
                   _fixed_measurer(211.25, 2, seen))
    assert (n.width, n.height) == (212, int(2 * math.ceil(40.5) + 33.0 / 5.0)) == (212, 88)
    assert seen == [("two\nlines", "Word", 33.0, 40.5, 500.0, 4320.0, "Verdana", "Normal", "Normal", "Left")]
    # line_height defaults to font_size; Fitted's default bounds are MAX_NODE_RESOLUTION
    seen.clear()
    n = _text_node({"type": "text", "text": "x", "font_size": 50.0}, _fixed_measurer(10.0, 1, seen))
    assert (n.width, n.height) == (10, 60) and seen[0][3:6] == (50.0, 7682.0, 4320.0)


def test_fitted_column_keeps_its_width():
    n = _text_node({"type": "text", "text": "column", "font_size": 20.0, "width": 300.0, "max_height": 90.0}, _fixed_measurer(123.4, 3))
    assert (n.width, n.height) == (300, 3 * 20 + 4)


def test_fixed_text_needs_no_shaper_and_fitted_text_without_one_is_refused():
    n = _text_node({"type": "text", "text": "fixed", "font_size": 20.0, "width": 120.0, "height": 48.0}, None)
    assert (n.width, n.height) == (120, 48)
    with pytest.raises(SceneError) as e:
        _text_node({"type": "text", "text": "fitted", "font_size": 20.0}, None)
    assert "smr_renderer_set_text_measurer" in str(e.value)

    def failing(_u, _p, _w, _n):
        return 1
    with pytest.raises(SceneError):
        _text_node({"type": "text", "text": "fitted", "font_size": 20.0}, _ffi.TEXT_MEASURE_FN(failing))


def test_shaper_measures_through_the_c_abi():
    book = _book()
    sh = T.Shaper(book)
    font = book.match("Inter 18pt")
    txt, fs, lh = "The quick brown fox\njumps over the lazy dog", 28.0, 34.0
    lines = T.layout(font, txt, fs)
    n = _text_node({"type": "text", "text": txt, "font_size": fs, "line_height": lh, "font_family": "Inter 18pt"}, sh.measurer)
    assert (n.width, n.height) == T.text_resolution(lines, fs, lh)
    assert len(lines) == 2
    # advance widths (hmtx) + pair kerning (GPOS `kern`), * font_size / unitsPerEm
    names = [font.glyph_name(c) for c in "The quick brown fox"]
    want = (sum(font.advance(g) for g in names) + sum(font.kerning(a, b) for a, b in zip(names, names[1:]))) * fs / font.upem
    assert abs(lines[0].width - want) < 1e-4


def test_word_and_glyph_wrap():
    font = _book().match("Inter 18pt")
    txt = "alpha beta gamma delta epsilon"
    one = T.layout(font, txt, 24.0)[0].width
    word = T.layout(font, txt, 24.0, "Word", one / 2.5)
    assert len(word) >= 3 and all(l.width <= one / 2.5 + 1e-6 for l in word)
    names = [[g for g, _ in l.glyphs] for l in word]
    assert all(n[0] != font.glyph_name(" ") and n[-1] != font.glyph_name(" ") for n in names)  # a break swallows its space
    assert sum(len(n) for n in names) + len(word) - 1 == len(txt)
    glyph = T.layout(font, txt, 24.0, "Glyph", one / 2.5)
    assert len(glyph) == 3 and sum(len(l.glyphs) for l in glyph) == len(txt)
    assert len(T.layout(font, txt, 24.0, "None", 10.0)) == 1


def _polygon_area(font, glyph, scale):
    a = 0.0
    for c in font.outline(glyph):
        a += sum(c[i][0] * c[(i + 1) % len(c)][1] - c[(i + 1) % len(c)][0] * c[i][1] for i in range(len(c))) / 2.0
    return abs(a) * scale * scale


@pytest.mark.parametrize("ch", ["o", "B", "8", "S", "@", "O"])  # (Inter draws e.g. "e" and "A" with overlapping contours)
def test_coverage_is_the_exact_area_of_the_outline(ch):
    """Signed-area accumulation: summed over the bitmap the coverage is the polygon's area (glyphs whose contours do not overlap)."""
    font = _book().match("Inter 18pt")
    g = font.glyph_name(ch)
    for size, fx, fy in ((40.0, 0.0, 0.0), (17.0, 0.3, 0.6), (96.0, 0.5, 0.25)):
        bmp, left, top = T.rasterise_glyph(font, g, size / font.upem, fx, fy)
        got, want = bmp.astype(np.float64).sum() / 255.0, _polygon_area(font, g, size / font.upem)
        assert abs(got - want) <= 0.004 * bmp.size + 1e-3, (ch, size, got, want)  # u8 quantisation: <= half a level per pixel
        assert bmp.max() == 255 or size < 20  # a stem at least a pixel wide is fully covered somewhere
        assert bmp[0].any() or bmp[1].any()   # tight box


def test_glyph_run_fits_the_node_and_the_atlas():
    sh = T.Shaper(_book())
    W, H = 260, 96
    glyphs, atlas = sh.rasterise("Fitted text\nsecond line", W, H, 30.0, 36.0, family="Inter 18pt", align="Center", color=(1.0, 0.5, 0.25, 1.0))
    assert atlas.dtype == np.uint8 and atlas.ndim == 2 and len(glyphs) == len("Fittedtextsecondline")
    for g in glyphs:
        assert 0 <= g.dst_x and g.dst_x + g.w <= W and 0 <= g.dst_y and g.dst_y + g.h <= H
        assert 0 <= g.atlas_x and g.atlas_x + g.w <= atlas.shape[1] and 0 <= g.atlas_y and g.atlas_y + g.h <= atlas.shape[0]
        assert g.color == (1.0, 0.5, 0.25, 1.0)
    # centred: the two lines' ink is symmetric about the node's centre to within a glyph's side bearings
    first = [g for g in glyphs if g.dst_y < 36]
    ink0, ink1 = min(g.dst_x for g in first), max(g.dst_x + g.w for g in first)
    assert abs((ink0 + ink1) / 2.0 - W / 2.0) < 4.0
    # the second line sits one line_height below the first
    second = [g for g in glyphs if g.dst_y >= 36]
    assert abs(min(g.dst_y for g in second) - min(g.dst_y for g in first) - 36) <= 3


def test_pair_kerning_comes_from_the_gpos_kern_feature():
    """rustybuzz applies the font's `kern` feature under Shaping::Advanced: Inter tucks V under A and o under T; pairs without an
    entry keep their advances; the adjustment is in font units and scales with the size."""
    if not os.path.isdir(REF_FONTS):
        pytest.skip("needs the reference's bundled Inter")
    font = T.FontBook.from_dir(REF_FONTS).match("Inter 18pt")
    g = font.glyph_name
    assert font.kerning(g("A"), g("V")) < -50 and font.kerning(g("T"), g("o")) < -50
    assert font.kerning(g("a"), g("b")) == 0.0 and font.kerning(g("o"), g("o")) == 0.0
    kerned, plain = T.layout(font, "AVATAR", 40.0)[0].width, T.layout(font, "AVATAR", 40.0, kerning=False)[0].width
    pairs = sum(font.kerning(g(a), g(b)) for a, b in zip("AVATAR", "VATAR"))
    assert abs((kerned - plain) - pairs * 40.0 / font.upem) < 1e-4 and kerned < plain
    assert abs(T.layout(font, "AVATAR", 80.0)[0].width - 2.0 * kerned) < 1e-3
