"""a12 / f4: the text pipeline BEHIND the C ABI (smr_fontbook_*, smelter_amd/csrc/host/text.cpp) against its Python twin
(tests/text_twin.py, fontTools-based): the same font matching, the same line breaks and widths, and the same glyph runs and atlases
BYTE FOR BYTE — so everything tests/test_text.py establishes about the Python shaper (hmtx advances + GPOS kern pairs, word / glyph wrap,
exact-area coverage) holds for the C++ one a C host links.  No GPU.  Glyph SHAPES remain unpinned against glyphon / swash (not in the
reference tree): both implementations here are this repository's reading of the same TrueType outlines."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from smelter_amd import _ffi
from tests import text_twin as T
from smelter_amd.scene import Scene

REF_FONTS = "/root/reference/smelter-render/fonts"          # Inter, bundled by the reference (only in the build container)
SYS_FONTS = "/usr/share/fonts/truetype/dejavu"              # part of the image, on the GPU box too
FONT_DIRS = [d for d in (REF_FONTS, SYS_FONTS) if os.path.isdir(d)]


@pytest.fixture(scope="module", params=FONT_DIRS or [None])
def books(request):
    if request.param is None:
        pytest.skip("no TrueType fonts on this machine")
    native = T.NativeFontBook()
    n = native.add_dir(request.param)
    py = T.FontBook.from_dir(request.param)
    assert n == len(py.fonts) == len(native)
    yield native, py, ("Inter 18pt" if request.param == REF_FONTS else "DejaVu Sans")
    native.close()


CASES = [
    ("CAM 3 LIVE", 24.0, 24.0, 176, 32, "Left", "None", "Normal", "Normal"),
    ("Fitted text\nsecond line", 30.0, 36.0, 260, 96, "Center", "None", "Normal", "Normal"),
    ("The quick brown fox jumps over the lazy dog AVATAR To. Ty fi", 19.5, 23.0, 200, 160, "Right", "Word", "Bold", "Normal"),
    ("éàü ß ÅÄÖ ñ glyph-wrap-test-of-a-long-word", 17.0, 20.0, 90, 220, "Left", "Glyph", "Normal", "Italic"),
    ("clipped by its node: jgpqy ÂÊ", 41.0, 30.0, 120, 40, "Justified", "None", "Normal", "Normal"),
    ("  leading and  double  spaces \n\n trailing ", 13.25, 17.5, 150, 90, "Center", "Word", "Medium", "Oblique"),
    ("", 20.0, 20.0, 64, 24, "Left", "None", "Normal", "Normal"),
    ("tab\tand nbsp and  em space, unknown 世界 glyphs", 16.0, 19.0, 400, 60, "Left", "Word", "Normal", "Normal"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0][:18].strip() or "empty" for c in CASES])
def test_glyph_run_and_atlas_equal_the_python_twin_byte_for_byte(books, case):
    native, py, family = books
    txt, fs, lh, W, H, align, wrap, weight, style = case
    sh = T.Shaper(py)
    color = (1.0, 0.5, 0.25, 0.75)
    g1, a1 = sh.rasterise(txt, W, H, fs, lh, family=family, weight=weight, style=style, wrap=wrap, align=align, color=color)
    g2, a2 = native.rasterise(txt, W, H, fs, lh, family=family, weight=weight, style=style, wrap=wrap, align=align, color=color)
    assert g1 == g2
    assert a1.shape == a2.shape and np.array_equal(a1, a2)


def test_measuring_equals_the_python_twin(books):
    native, py, family = books
    sh = T.Shaper(py)
    for txt, fs, wrap, mw in [("The quick brown fox\njumps over the lazy dog", 28.0, "None", 7682.0), ("alpha beta gamma delta epsilon", 24.0, "Word", 130.0),
                              ("alpha beta gamma delta epsilon", 24.0, "Glyph", 77.5), ("AVATAR Ty To", 61.0, "None", 7682.0), ("", 12.0, "Word", 10.0)]:
        p = _ffi.TextParams()
        p.text, p.font_family, p.style, p.weight, p.wrap, p.align = txt.encode(), family.encode(), b"Normal", b"Normal", wrap.encode(), b"Left"
        p.font_size, p.line_height, p.max_width, p.max_height = fs, fs, mw, 4320.0
        w, n = C.c_float(), C.c_uint32()
        assert sh._measure(None, C.pointer(p), C.pointer(w), C.pointer(n)) == 0
        assert native.measure(txt, fs, wrap, mw, family) == (w.value, n.value)


def test_font_matching_follows_family_slant_and_weight(books):
    """fontdb's matching, reduced, as both implementations state it: the family's faces (any face for an unknown family), the requested
    slant first, then the closest weight — observable through the runs they produce."""
    native, py, family = books
    sh = T.Shaper(py)
    for weight, style, fam in [("Bold", "Normal", family), ("Thin", "Italic", family), ("Black", "Oblique", family), ("Normal", "Normal", "No Such Family")]:
        assert native.rasterise("Rag", 120, 40, 30.0, family=fam, weight=weight, style=style)[0] == sh.rasterise("Rag", 120, 40, 30.0, family=fam, weight=weight, style=style)[0]
        a, b = native.rasterise("Rag", 120, 40, 30.0, family=fam, weight=weight, style=style)[1], sh.rasterise("Rag", 120, 40, 30.0, family=fam, weight=weight, style=style)[1]
        assert np.array_equal(a, b)
    if family == "Inter 18pt":  # bold and regular are different faces: the runs differ
        assert not np.array_equal(native.rasterise("Rag", 120, 40, 30.0, family=family, weight="Bold")[1], native.rasterise("Rag", 120, 40, 30.0, family=family)[1])


def test_the_book_is_the_scene_engines_measurer():
    """smr_fontbook_measure has smr_text_measure_fn's signature: the scene engine sizes fitted Text nodes with the reference's rule
    (get_text_resolution, text_renderer.rs:348-368) from the C++ layout — no Python in between."""
    if not FONT_DIRS:
        pytest.skip("no TrueType fonts on this machine")
    native = T.NativeFontBook()
    native.add_dir(FONT_DIRS[0])
    lib = _ffi.load()
    fn = C.cast(lib.smr_fontbook_measure, _ffi.TEXT_MEASURE_FN)
    s = Scene()
    try:
        assert lib.smr_scene_set_text_measurer(s._h, fn, native.handle) == 0
        txt, fs, lh = "Two lines\nof fitted text", 31.0, 36.5
        nodes = s.update({"type": "text", "text": txt, "font_size": fs, "line_height": lh}, 640, 360)
        node = [n for n in nodes if n.kind == _ffi.NODE_TEXT][0]
        widest, lines = native.measure(txt, fs)
        assert lines == 2 and (node.width, node.height) == (math.ceil(widest), int(2 * math.ceil(lh) + fs / 5.0))
    finally:
        s.close()
        native.close()


def test_bad_fonts_and_bad_text_are_errors_not_crashes(tmp_path):
    native = T.NativeFontBook()
    try:
        with pytest.raises(ValueError):
            native.add_font(str(tmp_path / "missing.ttf"))
        junk = tmp_path / "junk.ttf"
        junk.write_bytes(b"\x00\x01\x00\x00" + os.urandom(4096))
        with pytest.raises(ValueError):
            native.add_font(str(junk))
        with pytest.raises(ValueError):
            native.add_dir(str(tmp_path))  # nothing loadable below it
        with pytest.raises(ValueError):
            native.rasterise("x", 10, 10, 10.0)  # an empty book
        if FONT_DIRS:
            native.add_dir(FONT_DIRS[0])
            # truncated copies of a real face: refused or harmless, never a crash
            src = sorted(os.path.join(FONT_DIRS[0], f) for f in os.listdir(FONT_DIRS[0]) if f.endswith(".ttf"))[0]
            data = open(src, "rb").read()
            for cut in (12, 300, len(data) // 3, len(data) - 7):
                try:
                    native.add_font_bytes(data[:cut])
                except ValueError:
                    pass
            p = _ffi.TextParams()
            p.text, p.font_family, p.style, p.weight, p.wrap, p.align = b"\xff\xfe not utf-8", b"", b"Normal", b"Normal", b"None", b"Left"
            p.font_size = p.line_height = 12.0
            p.max_width = p.max_height = 100.0
            w, n = C.c_float(), C.c_uint32()
            assert native.lib.smr_fontbook_measure(native.handle, C.byref(p), C.byref(w), C.byref(n)) == 1
            # sizes no node can have are refused before any arithmetic is done with them (found by tests/san/host_fuzz.cpp under UBSan:
            # a pen position of a 1e30-pixel font cast to int)
            for bad in (0.0, -3.0, float("nan"), float("inf"), 1e6, 1e30):
                with pytest.raises(ValueError, match="font_size"):
                    native.measure("x", bad)
                with pytest.raises(ValueError, match="font_size"):
                    native.rasterise("x", 10, 10, bad)
            with pytest.raises(ValueError, match="line_height"):
                native.rasterise("x", 10, 10, 12.0, line_height=float("nan"))
            # unitsPerEm outside 16 .. 16384 (OpenType; ttf-parser refuses such a face too): with 1 unit per em every glyph would be
            # thousands of pixels wide
            tables = {data[12 + 16 * i:16 + 16 * i]: int.from_bytes(data[20 + 16 * i:24 + 16 * i], "big") for i in range(int.from_bytes(data[4:6], "big"))}
            head = tables[b"head"]
            for upem in (0, 1, 15, 16385, 65535):
                d = bytearray(data)
                d[head + 18:head + 20] = upem.to_bytes(2, "big")
                with pytest.raises(ValueError, match="degenerate font header"):
                    native.add_font_bytes(bytes(d))
    finally:
        native.close()


def test_corrupted_fonts_never_crash_the_reader():
    """Fonts are untrusted input (Renderer::register_font takes bytes from the API): every table read is bounds-checked.  A real face with
    random bytes overwritten — header, table directory, cmap, loca, glyf, GPOS alike — is refused or renders something, never crashes."""
    if not FONT_DIRS:
        pytest.skip("no TrueType fonts on this machine")
    src = sorted(os.path.join(FONT_DIRS[-1], f) for f in os.listdir(FONT_DIRS[-1]) if f.endswith(".ttf"))[0]
    data = bytearray(open(src, "rb").read())
    rng = np.random.default_rng(2024)
    loaded = 0
    for trial in range(150):
        d = bytearray(data)
        if trial % 3 == 0:  # the table directory and the headers
            lo, hi = 0, 1024
        elif trial % 3 == 1:
            lo, hi = 0, len(d)
        else:  # one dense burst somewhere
            lo = int(rng.integers(0, len(d) - 4096)); hi = lo + 4096
        for _ in range(int(rng.integers(1, 64))):
            d[int(rng.integers(lo, hi))] = int(rng.integers(0, 256))
        book = T.NativeFontBook()
        try:
            book.add_font_bytes(bytes(d))
            loaded += 1
            book.measure("Hamburgefonstiv AVATAR To. éà", 23.0, "Word", 90.0)
            book.rasterise("Hamburgefonstiv AVATAR To. éà 世", 120, 60, 23.0, wrap="Glyph")
        except ValueError:
            pass
        finally:
            book.close()
    assert loaded > 20  # (most mutations leave a loadable face: the layout and the rasteriser ran on damaged tables too)
