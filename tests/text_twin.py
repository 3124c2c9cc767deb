"""TEST INFRASTRUCTURE — the pure-Python twin of the library's text pipeline (smelter_amd/csrc/host/text.cpp behind smr_fontbook_*), fontTools-based;
tests/test_text_capi.py holds the two to each other byte for byte.  Not part of the product package (the binding is smelter_amd/fontbook.py).

Host-side text shaper and rasteriser for Text nodes: the caller-side half of a12.

The reference lays text out and rasterises it with glyphon / cosmic-text (third-party, not in the reference tree) and keeps only
the sizing rule and the layout parameters in its own code (smelter-render/src/transformations/text_renderer.rs:282-368).  The C ABI
mirrors that split: the scene engine applies the sizing rule (scene_build.cpp, get_text_resolution) to line metrics it gets from a
`smr_text_measure_fn`, and draws the glyph run the caller hands to `smr_renderer_set_text`.  This module is one such caller, for
TrueType fonts read with fontTools (the reference bundles Inter: smelter-render/fonts/*.ttf):

* `FontBook`   — (family, weight, style) -> font file, like the reference's font database (`TextRendererCtx::add_font`)
* `layout`     — cosmic-text's line model as far as the reference uses it: explicit newlines, `Wrap::None | Glyph | Word`
                 against the buffer width, advance widths from `hmtx` plus the pair kerning of the font's GPOS `kern` feature
                 (what rustybuzz applies under `Shaping::Advanced`), scaled by font_size / unitsPerEm.  NOT restated: GSUB
                 ligatures / contextual alternates, mark positioning, bidi and font fallback.
* `measurer`   — the `smr_text_measure_fn` for `Scene.set_text_measurer` / `Renderer.set_text_measurer`
* `rasterise`  — glyph outlines (quadratic B-splines of `glyf`) -> exact-area coverage (signed-area accumulation, one pass per
                 edge) -> R8 atlas + glyph run for `smr_renderer_set_text` / `Context.blit_glyphs`

Pure host code: no GPU, no oracle.  The GPU path (smr_blit_glyphs) is exercised with its output in tests/."""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from smelter_amd import _ffi
from smelter_amd.fontbook import NativeFontBook, TextGlyph  # noqa: F401  (re-exported: the tests address both pipelines through this module)

WEIGHTS = {"Thin": 100, "ExtraLight": 200, "Light": 300, "Normal": 400, "Medium": 500, "SemiBold": 600, "Bold": 700, "ExtraBold": 800,
           "Black": 900}


class Font:
    """One TrueType face: character map, advances, vertical metrics, flattened outlines."""

    def __init__(self, path: str):
        from fontTools.ttLib import TTFont
        self.path = path
        self.tt = TTFont(path, lazy=True)
        self.upem = float(self.tt["head"].unitsPerEm)
        hhea = self.tt["hhea"]
        self.ascent, self.descent = float(hhea.ascent), float(-hhea.descent)  # both positive, font units
        self.cmap = self.tt.getBestCmap()
        self.hmtx = self.tt["hmtx"]
        self.glyph_set = self.tt.getGlyphSet()
        name = self.tt["name"]
        self.family = (name.getDebugName(16) or name.getDebugName(1) or "").strip()
        os2 = self.tt["OS/2"] if "OS/2" in self.tt else None
        self.weight = int(os2.usWeightClass) if os2 is not None else 400
        self.italic = bool(os2.fsSelection & 1) if os2 is not None else False
        self._outlines: Dict[str, List[List[Tuple[float, float]]]] = {}
        self._kern_lookups = self._gpos_kern_lookups()
        self._kern_cache: Dict[Tuple[str, str], float] = {}

    def _gpos_kern_lookups(self):
        """PairPos subtables (GPOS lookup type 2, through extension lookups too) of the `kern` feature, default script first."""
        if "GPOS" not in self.tt:
            return []
        table = self.tt["GPOS"].table
        if not table.FeatureList or not table.LookupList:
            return []
        wanted: List[int] = []
        scripts = {sr.ScriptTag: sr.Script for sr in (table.ScriptList.ScriptRecord if table.ScriptList else [])}
        script = scripts.get("latn") or scripts.get("DFLT")
        feature_indices = script.DefaultLangSys.FeatureIndex if script is not None and script.DefaultLangSys else range(len(table.FeatureList.FeatureRecord))
        for fi in feature_indices:
            fr = table.FeatureList.FeatureRecord[fi]
            if fr.FeatureTag == "kern":
                wanted += [i for i in fr.Feature.LookupListIndex if i not in wanted]
        lookups = []
        for li in sorted(wanted):  # (lookups apply in lookup-list order)
            subs = []
            for st in table.LookupList.Lookup[li].SubTable:
                if getattr(st, "LookupType", None) == 9:
                    st = st.ExtSubTable
                if getattr(st, "LookupType", None) == 2:
                    subs.append(st)
            if subs:
                lookups.append(subs)
        return lookups

    def kerning(self, left: str, right: str) -> float:
        """x-advance adjustment of `left` when followed by `right`, font units (GPOS PairPos formats 1 and 2)."""
        key = (left, right)
        if key in self._kern_cache:
            return self._kern_cache[key]
        total = 0.0
        for subs in self._kern_lookups:
            for st in subs:  # the first subtable of a lookup that covers the pair decides
                if left not in st.Coverage.glyphs:
                    continue
                value = None
                if st.Format == 1:
                    for rec in st.PairSet[st.Coverage.glyphs.index(left)].PairValueRecord:
                        if rec.SecondGlyph == right:
                            value = rec.Value1
                            break
                    if value is None:
                        continue
                else:
                    c1 = st.ClassDef1.classDefs.get(left, 0)
                    c2 = st.ClassDef2.classDefs.get(right, 0)
                    value = st.Class1Record[c1].Class2Record[c2].Value1
                total += float(getattr(value, "XAdvance", 0) or 0) if value is not None else 0.0
                break
        self._kern_cache[key] = total
        return total

    def glyph_name(self, ch: str) -> str:
        return self.cmap.get(ord(ch), ".notdef")

    def advance(self, glyph: str) -> float:
        return float(self.hmtx[glyph][0])

    def outline(self, glyph: str) -> List[List[Tuple[float, float]]]:
        """Closed polylines in font units (y up): quadratic segments flattened to 8 chords each."""
        if glyph in self._outlines:
            return self._outlines[glyph]
        from fontTools.pens.basePen import decomposeQuadraticSegment
        from fontTools.pens.recordingPen import DecomposingRecordingPen
        pen = DecomposingRecordingPen(self.glyph_set)
        self.glyph_set[glyph].draw(pen)
        contours: List[List[Tuple[float, float]]] = []
        cur: List[Tuple[float, float]] = []
        for op, pts in pen.value:
            if op == "moveTo":
                cur = [tuple(map(float, pts[0]))]
            elif op == "lineTo":
                cur.append(tuple(map(float, pts[0])))
            elif op == "qCurveTo":
                if pts[-1] is None:  # closed contour of off-curve points only: start at the midpoint of the last and first
                    offs = list(pts[:-1])
                    start = ((offs[-1][0] + offs[0][0]) / 2.0, (offs[-1][1] + offs[0][1]) / 2.0)
                    cur = [start]
                    pts = tuple(offs) + (start,)
                for c, p1 in decomposeQuadraticSegment(pts):
                    p0 = cur[-1]
                    for k in range(1, 9):
                        t = k / 8.0
                        a, b, d = (1 - t) * (1 - t), 2 * t * (1 - t), t * t
                        cur.append((a * p0[0] + b * c[0] + d * p1[0], a * p0[1] + b * c[1] + d * p1[1]))
            elif op == "curveTo":  # (CFF outlines: not expected in a TrueType face)
                p0, (c1, c2, p1) = cur[-1], pts
                for k in range(1, 13):
                    t = k / 12.0
                    m = 1 - t
                    cur.append((m ** 3 * p0[0] + 3 * m * m * t * c1[0] + 3 * m * t * t * c2[0] + t ** 3 * p1[0],
                                m ** 3 * p0[1] + 3 * m * m * t * c1[1] + 3 * m * t * t * c2[1] + t ** 3 * p1[1]))
            elif op in ("closePath", "endPath"):
                if len(cur) > 1:
                    contours.append(cur)
                cur = []
        self._outlines[glyph] = contours
        return contours


class FontBook:
    """font_system.db(): faces by family; the closest weight of the requested style wins (fontdb's matching, reduced)."""

    def __init__(self, paths: Sequence[str] = ()):
        self.fonts: List[Font] = []
        for p in paths:
            self.add_font(p)

    def add_font(self, path: str) -> Font:
        f = Font(path)
        self.fonts.append(f)
        return f

    @staticmethod
    def from_dir(directory: str) -> "FontBook":
        paths = []
        for root, _dirs, names in os.walk(directory):
            paths += [os.path.join(root, n) for n in names if n.lower().endswith(".ttf")]
        return FontBook(sorted(paths))

    @staticmethod
    def system() -> "FontBook":
        """The machine's TrueType fonts (FontSystem::new loads the system font database), SMR_FONT_DIR first."""
        for d in (os.environ.get("SMR_FONT_DIR"), "/usr/share/fonts/truetype", "/usr/share/fonts"):
            if d and os.path.isdir(d):
                book = FontBook.from_dir(d)
                if book.fonts:
                    return book
        raise FileNotFoundError("no TrueType fonts found (set SMR_FONT_DIR)")

    def match(self, family: str, weight: str = "Normal", style: str = "Normal") -> Font:
        if not self.fonts:
            raise ValueError("the font book is empty")
        want_w, want_i = WEIGHTS.get(weight, 400), style in ("Italic", "Oblique")
        fam = [f for f in self.fonts if f.family.lower() == family.lower()] or self.fonts  # unknown family: any face (fallback)
        return min(fam, key=lambda f: (f.italic != want_i, abs(f.weight - want_w)))


@dataclass
class Line:
    glyphs: List[Tuple[str, float]]  # (glyph name, pen x of its origin in pixels)
    width: float


def layout(font: Font, text: str, font_size: float, wrap: str = "None", max_width: float = math.inf, kerning: bool = True) -> List[Line]:
    """Buffer::set_text + set_wrap + shape_until_scroll as far as widths and line breaks go (one Line per LayoutLine)."""
    scale = font_size / font.upem
    lines: List[Line] = []
    for para in text.split("\n"):
        cur: List[Tuple[str, float]] = []
        x = 0.0
        last_space = -1  # index in cur after which a Word wrap may break
        i = 0
        chars = list(para)
        while i < len(chars):
            ch = chars[i]
            g = font.glyph_name(ch)
            if cur and kerning:  # the previous glyph's advance, adjusted for this pair
                x += font.kerning(cur[-1][0], g) * scale
            adv = font.advance(g) * scale
            if wrap != "None" and cur and x + adv > max_width and not ch.isspace():
                if wrap == "Word" and last_space >= 0:
                    head, tail = cur[:last_space + 1], cur[last_space + 1:]
                    while head and head[-1][0] == font.glyph_name(" "):  # the break swallows the trailing space
                        head.pop()
                    lines.append(Line(head, (head[-1][1] + font.advance(head[-1][0]) * scale) if head else 0.0))
                    shift = tail[0][1] if tail else x
                    cur = [(n, px - shift) for n, px in tail]
                    x -= shift
                else:
                    lines.append(Line(cur, x))
                    cur, x = [], 0.0
                last_space = -1
            cur.append((g, x))
            x += adv
            if ch.isspace():
                last_space = len(cur) - 1
            i += 1
        lines.append(Line(cur, x))
    return lines


def text_resolution(lines: Sequence[Line], font_size: float, line_height: float) -> Tuple[int, int]:
    """TextRendererCtx::get_text_resolution (text_renderer.rs:348-368)."""
    width = max([int(math.ceil(l.width)) for l in lines] + [0])
    height = int(len(lines) * math.ceil(line_height) + font_size / 5.0)
    return width, height


class Shaper:
    """What a caller plugs into the C ABI: measuring for the scene engine, glyph runs for smr_renderer_set_text."""

    def __init__(self, book: FontBook):
        self.book = book
        self._cb = _ffi.TEXT_MEASURE_FN(self._measure)  # (kept alive with the object)

    def _measure(self, _user, params, widest, count):
        try:
            p = params.contents
            font = self.book.match((p.font_family or b"").decode(), (p.weight or b"Normal").decode(), (p.style or b"Normal").decode())
            lines = layout(font, (p.text or b"").decode(), p.font_size, (p.wrap or b"None").decode(), p.max_width)
            widest[0] = max([l.width for l in lines] + [0.0])
            count[0] = len(lines)
            return 0
        except Exception:  # never let an exception cross the C boundary
            return 1

    @property
    def measurer(self):
        return self._cb

    def rasterise(self, text: str, width: int, height: int, font_size: float, line_height: Optional[float] = None, family: str = "",
                  weight: str = "Normal", style: str = "Normal", wrap: str = "None", align: str = "Left",
                  color: Sequence[float] = (1.0, 1.0, 1.0, 1.0)):
        """The glyph run of one Text node of `width` x `height` pixels: (glyphs, atlas) for smr_renderer_set_text.

        Lines are `line_height` apart; inside its line box a line's ascent + descent is centred and the baseline follows
        (cosmic-text's LayoutRun placement).  Glyph bitmaps are rasterised at their fractional pen position."""
        font = self.book.match(family, weight, style)
        line_height = font_size if line_height is None else line_height
        scale = font_size / font.upem
        lines = layout(font, text, font_size, wrap, float(width))
        asc, desc = font.ascent * scale, font.descent * scale
        placed = []  # (glyph, x, baseline_y)
        for li, line in enumerate(lines):
            free = float(width) - line.width
            x0 = {"Left": 0.0, "Justified": 0.0, "Center": free / 2.0, "Right": free}.get(align, 0.0)
            base = li * line_height + (line_height - (asc + desc)) / 2.0 + asc
            placed += [(g, x0 + px, base) for g, px in line.glyphs]
        # rasterise every (glyph, fractional offset) once, pack the bitmaps into one atlas row by row
        cache: Dict[Tuple[str, float, float], Tuple[np.ndarray, int, int]] = {}
        boxes, order = [], []
        for g, px, base in placed:
            fx, fy = px - math.floor(px), base - math.floor(base)
            key = (g, round(fx, 3), round(fy, 3))
            if key not in cache:
                cache[key] = rasterise_glyph(font, g, scale, fx, fy)
            bmp, left, top = cache[key]
            if bmp.size:
                order.append((key, int(math.floor(px)) + left, int(math.floor(base)) - top))
        keys = [k for k in cache if cache[k][0].size]
        aw = max([cache[k][0].shape[1] for k in keys] + [1])
        aw = max(aw, 256)
        pos, cx, cy, rowh = {}, 0, 0, 0
        for k in keys:
            h, w = cache[k][0].shape
            if cx + w > aw:
                cx, cy, rowh = 0, cy + rowh, 0
            pos[k] = (cx, cy)
            cx += w
            rowh = max(rowh, h)
        atlas = np.zeros((max(cy + rowh, 1), aw), np.uint8)
        for k in keys:
            bmp = cache[k][0]
            x, y = pos[k]
            atlas[y:y + bmp.shape[0], x:x + bmp.shape[1]] = bmp
        glyphs = []
        for key, dx, dy in order:
            h, w = cache[key][0].shape
            ax, ay = pos[key]
            # clip to the node (smr_blit_glyphs wants quads inside the target)
            x0, y0, x1, y1 = max(dx, 0), max(dy, 0), min(dx + w, width), min(dy + h, height)
            if x1 > x0 and y1 > y0:
                glyphs.append(TextGlyph(x0, y0, x1 - x0, y1 - y0, ax + (x0 - dx), ay + (y0 - dy), tuple(float(c) for c in color)))
        return glyphs, atlas


def rasterise_glyph(font: Font, glyph: str, scale: float, fx: float = 0.0, fy: float = 0.0) -> Tuple[np.ndarray, int, int]:
    """Exact-area coverage of one glyph at `scale` pixels per font unit, origin at the fractional pixel offset (fx, fy) of its
    cell.  Returns (u8 bitmap, left, top): the bitmap's top-left pixel sits `left` right of and `top` above the pen's pixel.

    Signed-area accumulation: every edge adds, to each pixel row it crosses, the area it sweeps to its right; a running sum along
    the row turns the deltas into coverage (non-zero winding for outlines that do not self-overlap)."""
    contours = font.outline(glyph)
    if not contours:
        return np.zeros((0, 0), np.uint8), 0, 0
    pts = [[(x * scale + fx, -y * scale + fy) for x, y in c] for c in contours]  # pixels, y down, relative to the pen's pixel
    xs = [p[0] for c in pts for p in c]
    ys = [p[1] for c in pts for p in c]
    left, top = int(math.floor(min(xs))), int(math.floor(min(ys)))
    w, h = int(math.ceil(max(xs))) - left + 1, int(math.ceil(max(ys))) - top + 1
    acc = np.zeros(w * h + 4, np.float64)
    for c in pts:
        n = len(c)
        for i in range(n):
            _accumulate_edge(acc, w, h, c[i][0] - left, c[i][1] - top, c[(i + 1) % n][0] - left, c[(i + 1) % n][1] - top)
    cov = np.abs(np.cumsum(acc)[:w * h]).reshape(h, w)
    bmp = (np.clip(cov, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    return bmp, left, -top


def _accumulate_edge(a: np.ndarray, w: int, h: int, x0: float, y0: float, x1: float, y1: float):
    if y0 == y1:
        return
    d = 1.0
    if y0 > y1:
        x0, y0, x1, y1, d = x1, y1, x0, y0, -1.0
    dxdy = (x1 - x0) / (y1 - y0)
    x = x0
    x_lo, x_hi = min(x0, x1), max(x0, x1)  # (rounding must not carry x past the edge's own end: text.cpp accumulate_edge)
    for y in range(max(int(math.floor(y0)), 0), min(h, int(math.ceil(y1)))):
        dy = min(y + 1.0, y1) - max(float(y), y0)
        xn = min(max(x + dxdy * dy, x_lo), x_hi)
        s = d * dy
        xa, xb = (x, xn) if x < xn else (xn, x)
        ia, ib = int(math.floor(xa)), int(math.ceil(xb))
        row = y * w
        if ib <= ia + 1:  # the edge stays inside one pixel column
            xm = 0.5 * (x + xn) - ia
            a[row + ia] += s - s * xm
            a[row + ia + 1] += s * xm
        else:
            inv = 1.0 / (xb - xa)
            fa = xa - ia
            a0 = 0.5 * inv * (1.0 - fa) * (1.0 - fa)
            fb = xb - ib + 1.0
            am = 0.5 * inv * fb * fb
            a[row + ia] += s * a0
            if ib == ia + 2:
                a[row + ia + 1] += s * (1.0 - a0 - am)
            else:
                a1 = inv * (1.5 - fa)
                a[row + ia + 1] += s * (a1 - a0)
                for xi in range(ia + 2, ib - 1):
                    a[row + xi] += s * inv
                a2 = a1 + (ib - ia - 3) * inv
                a[row + ib - 1] += s * (1.0 - a2 - am)
            a[row + ib] += s * am
        x = xn
