"""ctypes binding of the library's text pipeline — smr_fontbook_* (smelter_amd/csrc/host/text.cpp: font database, layout, rasteriser in C++
behind the C ABI, TextRendererCtx of smelter-render/src/transformations/text_renderer.rs:236-368).  Binding only: the fontTools-based
Python twin the tests hold the C++ pipeline to lives with the tests (tests/text_twin.py)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _ffi


class NativeFontBook:
    """smr_fontbook (smelter_amd/csrc/host/text.cpp): the product's text pipeline — font database, layout, rasteriser in C++ behind
    the C ABI.  tests/text_twin.py is its pure-Python twin; tests/test_text_capi.py holds the two to each other byte for byte."""

    def __init__(self, paths: Sequence[str] = ()):
        self.lib = _ffi.load()
        h = _ffi.C.c_void_p()
        if self.lib.smr_fontbook_create(_ffi.C.byref(h)) != 0:
            raise MemoryError("smr_fontbook_create")
        self._h = h
        for p in paths:
            self.add_font(p)

    def _check(self, rc: int) -> int:
        if rc < 0:
            raise ValueError(self.lib.smr_fontbook_last_error(self._h).decode())
        return rc

    def add_font(self, path: str):
        self._check(self.lib.smr_fontbook_add_file(self._h, path.encode()))

    def add_font_bytes(self, data: bytes):
        self._check(self.lib.smr_fontbook_add_memory(self._h, data, len(data)))

    def add_dir(self, directory: str) -> int:
        return self._check(self.lib.smr_fontbook_add_dir(self._h, directory.encode()))

    @staticmethod
    def system() -> "NativeFontBook":
        for d in (os.environ.get("SMR_FONT_DIR"), "/usr/share/fonts/truetype", "/usr/share/fonts"):
            if d and os.path.isdir(d):
                book = NativeFontBook()
                try:
                    book.add_dir(d)
                    return book
                except ValueError:
                    book.close()
        raise FileNotFoundError("no TrueType fonts found (set SMR_FONT_DIR)")

    def __len__(self):
        return int(self.lib.smr_fontbook_count(self._h))

    @property
    def handle(self):
        return self._h

    @staticmethod
    def _params(text, font_size, line_height, family, weight, style, wrap, align, max_width, max_height):
        p = _ffi.TextParams()
        keep = [text.encode(), family.encode(), style.encode(), weight.encode(), wrap.encode(), align.encode()]
        p.text, p.font_family, p.style, p.weight, p.wrap, p.align = keep
        p.font_size, p.line_height, p.max_width, p.max_height = font_size, line_height, max_width, max_height
        return p, keep

    def measure(self, text: str, font_size: float, wrap: str = "None", max_width: float = 7682.0, family: str = "", weight: str = "Normal",
                style: str = "Normal") -> Tuple[float, int]:
        p, _keep = self._params(text, font_size, font_size, family, weight, style, wrap, "Left", max_width, 4320.0)
        w, n = _ffi.C.c_float(), _ffi.C.c_uint32()
        if self.lib.smr_fontbook_measure(self._h, _ffi.C.byref(p), _ffi.C.byref(w), _ffi.C.byref(n)) != 0:
            raise ValueError(self.lib.smr_fontbook_last_error(self._h).decode())
        return w.value, n.value

    def rasterise(self, text: str, width: int, height: int, font_size: float, line_height: Optional[float] = None, family: str = "",
                  weight: str = "Normal", style: str = "Normal", wrap: str = "None", align: str = "Left",
                  color: Sequence[float] = (1.0, 1.0, 1.0, 1.0)):
        """(glyphs, atlas): the glyph run and the R8 atlas for smr_renderer_set_text / Context.blit_glyphs, computed by the C++ pipeline."""
        p, _keep = self._params(text, font_size, font_size if line_height is None else line_height, family, weight, style, wrap, align,
                                float(width), float(height))
        col = (_ffi.C.c_float * 4)(*[float(c) for c in color])
        run = _ffi.TextRun()
        self._check(self.lib.smr_fontbook_rasterise(self._h, _ffi.C.byref(p), width, height, col, _ffi.C.byref(run)))
        glyphs = [TextGlyph(g.dst_x, g.dst_y, g.w, g.h, g.atlas_x, g.atlas_y, tuple(g.color)) for g in (run.glyphs[i] for i in range(run.n_glyphs))]
        atlas = np.ctypeslib.as_array(run.atlas, shape=(run.atlas_h, run.atlas_w)).copy()
        return glyphs, atlas

    def close(self):
        if getattr(self, "_h", None):
            self.lib.smr_fontbook_destroy(self._h)
            self._h = None


@dataclass
class TextGlyph:  # field for field include/smr.h smr_glyph
    dst_x: int
    dst_y: int
    w: int
    h: int
    atlas_x: int
    atlas_y: int
    color: Tuple[float, float, float, float]
