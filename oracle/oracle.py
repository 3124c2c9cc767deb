"""ctypes front-end of the CPU oracle (oracle/smr_oracle.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under smelter_amd/ may import this module.

Every function mirrors one pass of the reference renderer; the C source cites the
reference file:line each one restates.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

PX_RGBA8_SRGB = 0
PX_RGBA8_UNORM = 1
PX_RGBA16F = 2

YUV420, YUV422, YUV444, YUVJ420 = 0, 1, 2, 3
MAX_MASKS = 20


def build(force: bool = False) -> None:
    src = os.path.join(_HERE, "smr_oracle.c")
    outs = [os.path.join(_BUILD, "liborc.so"), os.path.join(_BUILD, "liborc_omp.so")]
    if not force and all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(src) for o in outs):
        return
    subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)


class _Mask(C.Structure):
    _fields_ = [("radius", C.c_float * 4), ("top", C.c_float), ("left", C.c_float), ("width", C.c_float), ("height", C.c_float)]


class _Layout(C.Structure):
    _fields_ = [
        ("top", C.c_float), ("left", C.c_float), ("width", C.c_float), ("height", C.c_float),
        ("rotation_degrees", C.c_float),
        ("border_radius", C.c_float * 4),
        ("type", C.c_uint32),
        ("source_index", C.c_uint32),
        ("color", C.c_float * 4),
        ("border_color", C.c_float * 4),
        ("border_width", C.c_float),
        ("crop", C.c_float * 4),
        ("blur_radius", C.c_float),
        ("masks_len", C.c_uint32),
        ("masks", _Mask * MAX_MASKS),
    ]


class _Source(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_int), ("h", C.c_int)]


class _Plan(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("levels", C.c_int * 2), ("reduced_w", C.c_int), ("reduced_h", C.c_int),
        ("axis", C.c_int * 2), ("scale", C.c_float * 2), ("offset", C.c_float * 2), ("perp_offset", C.c_int * 2),
        ("mid_w", C.c_int), ("mid_h", C.c_int),
    ]


class _Glyph(C.Structure):
    _fields_ = [("dst_x", C.c_int32), ("dst_y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("atlas_x", C.c_int32), ("atlas_y", C.c_int32), ("color", C.c_float * 4)]


@dataclass
class Mask:
    radius: Sequence[float]  # tl, tr, br, bl
    top: float
    left: float
    width: float
    height: float


@dataclass
class Layout:
    """POD mirror of RenderLayout + params.rs byte conventions (shared by oracle and smr.h)."""
    top: float
    left: float
    width: float
    height: float
    type: int = 1  # 0 texture, 1 colour, 2 box shadow
    rotation_degrees: float = 0.0
    border_radius: Sequence[float] = (0.0, 0.0, 0.0, 0.0)
    source_index: int = 0xFFFFFFFF
    color: Sequence[float] = (0.0, 0.0, 0.0, 0.0)
    border_color: Sequence[float] = (0.0, 0.0, 0.0, 0.0)
    border_width: float = 0.0
    crop: Sequence[float] = (0.0, 0.0, 0.0, 0.0)  # top, left, width, height
    blur_radius: float = 0.0
    masks: List[Mask] = field(default_factory=list)


_lib = None
_lib_omp = None


def _load(omp: bool = False):
    global _lib, _lib_omp
    build()
    if omp:
        if _lib_omp is None:
            _lib_omp = _bind(C.CDLL(os.path.join(_BUILD, "liborc_omp.so")))
        return _lib_omp
    if _lib is None:
        _lib = _bind(C.CDLL(os.path.join(_BUILD, "liborc.so")))
    return _lib


def _bind(lib):
    P, I, F = C.c_void_p, C.c_int, C.c_float
    lib.orc_init.restype = None
    lib.orc_srgb_decode_table.restype = C.POINTER(C.c_float)
    lib.orc_srgb_threshold_table.restype = C.POINTER(C.c_float)
    lib.orc_srgb_encode8.argtypes = [F]
    lib.orc_srgb_encode8.restype = C.c_uint8
    lib.orc_f32_to_f16.argtypes = [F]
    lib.orc_f32_to_f16.restype = C.c_uint16
    lib.orc_f16_to_f32.argtypes = [C.c_uint16]
    lib.orc_f16_to_f32.restype = F
    lib.orc_planar_yuv_to_rgba.argtypes = [P, P, P, I, I, I, P]
    lib.orc_nv12_to_rgba.argtypes = [P, P, I, I, P]
    lib.orc_interleaved422_to_rgba.argtypes = [P, I, I, I, P]
    lib.orc_swizzle_to_rgba.argtypes = [P, I, I, I, P]
    lib.orc_add_premultiplied_alpha.argtypes = [P, I, I, I, P]
    lib.orc_remove_premultiplied_alpha.argtypes = [P, I, I, P]
    lib.orc_rgba_to_planar_yuv.argtypes = [P, I, I, I, P, P, P]
    lib.orc_rgba_to_nv12.argtypes = [P, I, I, P, P]
    lib.orc_rgb_to_yuv_bytes.argtypes = [I, I, I, P]
    lib.orc_harness_yuv420_to_rgba.argtypes = [P, P, P, I, I, P]
    lib.orc_downsample.argtypes = [P, I, I, I, I, I, P, I, I]
    lib.orc_resample_pass.argtypes = [P, I, I, I, I, F, F, I, P, I, I, I]
    lib.orc_resample_plan_make.argtypes = [I, I, C.POINTER(C.c_float), I, I, C.POINTER(_Plan)]
    lib.orc_resample_plan_make.restype = I
    lib.orc_resample.argtypes = [P, I, I, I, C.POINTER(C.c_float), P, I, I]
    lib.orc_resample.restype = I
    lib.orc_rescale_bilinear.argtypes = [P, I, I, I, P, I, I]
    lib.orc_apply_layouts.argtypes = [P, I, I, C.POINTER(_Layout), I, C.POINTER(_Source), I, I]
    lib.orc_render_frame_yuv420.argtypes = [C.POINTER(P), C.POINTER(P), C.POINTER(P), I, I, I, C.POINTER(_Layout), I, C.POINTER(_Source),
                                            C.POINTER(I), I, I, I, P, P, P]
    lib.orc_render_frame_yuv420.restype = I
    lib.orc_blit_glyphs.argtypes = [P, I, I, C.POINTER(C.c_float), C.POINTER(_Glyph), I, P, I, I, I]
    lib.orc_gaussian_blur.argtypes = [P, I, I, I, F, P, P]
    lib.orc_builtin_shader.argtypes = [I, C.POINTER(_Source), I, P, F, I, I, I, P]
    lib.orc_builtin_shader.restype = I
    lib.orc_sizeof_layout.restype = I
    lib.orc_sizeof_plan.restype = I
    lib.orc_num_threads.restype = I
    for name in ("orc_planar_yuv_to_rgba", "orc_nv12_to_rgba", "orc_interleaved422_to_rgba", "orc_swizzle_to_rgba",
                 "orc_add_premultiplied_alpha", "orc_remove_premultiplied_alpha", "orc_rgba_to_planar_yuv",
                 "orc_rgba_to_nv12", "orc_rgb_to_yuv_bytes", "orc_harness_yuv420_to_rgba", "orc_downsample",
                 "orc_resample_pass", "orc_rescale_bilinear", "orc_apply_layouts", "orc_blit_glyphs", "orc_gaussian_blur"):
        getattr(lib, name).restype = None
    assert lib.orc_sizeof_layout() == C.sizeof(_Layout)
    assert lib.orc_sizeof_plan() == C.sizeof(_Plan)
    lib.orc_init()
    return lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _u8(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint8)


def num_threads(omp: bool = False) -> int:
    return _load(omp).orc_num_threads()


# --------------------------------------------------------------------- scalars
def srgb_decode_table() -> np.ndarray:
    return np.ctypeslib.as_array(_load().orc_srgb_decode_table(), shape=(256,)).copy()


def srgb_threshold_table() -> np.ndarray:
    return np.ctypeslib.as_array(_load().orc_srgb_threshold_table(), shape=(257,)).copy()


def srgb_encode8(x: float) -> int:
    return int(_load().orc_srgb_encode8(float(x)))


def f32_to_f16_bits(x: float) -> int:
    return int(_load().orc_f32_to_f16(float(x)))


def color_to_shader(rgba8: Sequence[int], srgb: bool = True) -> List[float]:
    """convert_to_shader_color, smelter-render/src/wgpu/utils.rs:51-81 (f64 maths, f32 cast)."""
    r, g, b, a8 = rgba8
    a = a8 / 255.0

    def lin(c):
        c = c / 255.0
        return c / 12.92 if c < 0.04045 else ((c + 0.055) / 1.055) ** 2.4

    if srgb:
        vals = [a * lin(r), a * lin(g), a * lin(b), a]
    else:
        vals = [a * r / 255.0, a * g / 255.0, a * b / 255.0, a]
    return [float(np.float32(v)) for v in vals]


# ------------------------------------------------------------------ converters
def chroma_shape(w: int, h: int, variant: int):
    if variant == YUV422:
        return h, w // 2
    if variant == YUV444:
        return h, w
    return h // 2, w // 2


def planar_yuv_to_rgba(y, u, v, w, h, variant=YUV420, omp=False) -> np.ndarray:
    y, u, v = _u8(y), _u8(u), _u8(v)
    out = np.empty((h, w, 4), np.uint8)
    _load(omp).orc_planar_yuv_to_rgba(_p(y), _p(u), _p(v), w, h, variant, _p(out))
    return out


def nv12_to_rgba(y, uv, w, h) -> np.ndarray:
    y, uv = _u8(y), _u8(uv)
    out = np.empty((h, w, 4), np.uint8)
    _load().orc_nv12_to_rgba(_p(y), _p(uv), w, h, _p(out))
    return out


def interleaved422_to_rgba(data, w, h, order) -> np.ndarray:
    data = _u8(data)
    out = np.empty((h, w, 4), np.uint8)
    _load().orc_interleaved422_to_rgba(_p(data), w, h, order, _p(out))
    return out


def swizzle_to_rgba(data, w, h, kind) -> np.ndarray:
    """kind 0 = BGRA input, 1 = ARGB input."""
    data = _u8(data)
    out = np.empty((h, w, 4), np.uint8)
    _load().orc_swizzle_to_rgba(_p(data), w, h, kind, _p(out))
    return out


def add_premultiplied_alpha(rgba, srgb: bool) -> np.ndarray:
    rgba = _u8(rgba)
    h, w = rgba.shape[:2]
    out = np.empty_like(rgba)
    _load().orc_add_premultiplied_alpha(_p(rgba), w, h, int(srgb), _p(out))
    return out


def remove_premultiplied_alpha(rgba) -> np.ndarray:
    rgba = _u8(rgba)
    h, w = rgba.shape[:2]
    out = np.empty_like(rgba)
    _load().orc_remove_premultiplied_alpha(_p(rgba), w, h, _p(out))
    return out


def rgba_to_planar_yuv(rgba, variant=YUV420, omp=False):
    rgba = _u8(rgba)
    h, w = rgba.shape[:2]
    ch, cw = chroma_shape(w, h, variant)
    y = np.empty((h, w), np.uint8)
    u = np.empty((ch, cw), np.uint8)
    v = np.empty((ch, cw), np.uint8)
    _load(omp).orc_rgba_to_planar_yuv(_p(rgba), w, h, variant, _p(y), _p(u), _p(v))
    return y, u, v


def rgba_to_nv12(rgba):
    rgba = _u8(rgba)
    h, w = rgba.shape[:2]
    y = np.empty((h, w), np.uint8)
    uv = np.empty((h // 2, w // 2, 2), np.uint8)
    _load().orc_rgba_to_nv12(_p(rgba), w, h, _p(y), _p(uv))
    return y, uv


def rgb_to_yuv_bytes(r, g, b):
    out = np.zeros(3, np.uint8)
    _load().orc_rgb_to_yuv_bytes(r, g, b, _p(out))
    return tuple(int(x) for x in out)


def harness_yuv420_to_rgba(y, u, v, w, h) -> np.ndarray:
    y, u, v = _u8(y), _u8(u), _u8(v)
    cw, ch = w - w % 2, h - h % 2
    out = np.empty((ch, cw, 4), np.uint8)
    _load().orc_harness_yuv420_to_rgba(_p(y), _p(u), _p(v), w, h, _p(out))
    return out


# ------------------------------------------------------------------- resampler
@dataclass
class ResamplePlan:
    kind: int
    levels: tuple
    reduced: tuple
    axis: tuple
    scale: tuple
    offset: tuple
    perp_offset: tuple
    mid: tuple


def resample_plan(src_w, src_h, crop, dst_w, dst_h) -> ResamplePlan:
    """crop = (top, left, width, height)."""
    p = _Plan()
    c = (C.c_float * 4)(*[float(x) for x in crop])
    _load().orc_resample_plan_make(src_w, src_h, c, dst_w, dst_h, C.byref(p))
    return ResamplePlan(p.kind, tuple(p.levels), (p.reduced_w, p.reduced_h), tuple(p.axis), tuple(p.scale),
                        tuple(p.offset), tuple(p.perp_offset), (p.mid_w, p.mid_h))


def _px_bytes(fmt):
    return 8 if fmt == PX_RGBA16F else 4


def _alloc(fmt, w, h):
    return np.empty((h, w, 4), np.uint16 if fmt == PX_RGBA16F else np.uint8)


def downsample(src, src_fmt, fx, fy) -> np.ndarray:
    src = np.ascontiguousarray(src)
    sh, sw = src.shape[:2]
    dw, dh = -(-sw // fx), -(-sh // fy)
    out = _alloc(PX_RGBA16F, dw, dh)
    _load().orc_downsample(_p(src), src_fmt, sw, sh, fx, fy, _p(out), dw, dh)
    return out


def resample_pass(src, src_fmt, axis, scale, offset, perp_offset, dst_fmt, dw, dh, omp=False) -> np.ndarray:
    src = np.ascontiguousarray(src)
    sh, sw = src.shape[:2]
    out = _alloc(dst_fmt, dw, dh)
    _load(omp).orc_resample_pass(_p(src), src_fmt, sw, sh, axis, float(scale), float(offset), int(perp_offset), _p(out),
                                 dst_fmt, dw, dh)
    return out


def resample(src, crop, dw, dh, src_fmt=PX_RGBA8_SRGB, omp=False):
    """Full ResampledChild::render. Returns (kind, dst or None when direct)."""
    src = _u8(src)
    sh, sw = src.shape[:2]
    out = np.zeros((dh, dw, 4), np.uint8)
    c = (C.c_float * 4)(*[float(x) for x in crop])
    kind = _load(omp).orc_resample(_p(src), src_fmt, sw, sh, c, _p(out), dw, dh)
    return kind, (out if kind > 0 else None)


def rescale_bilinear(src, dw, dh, fmt=PX_RGBA8_SRGB) -> np.ndarray:
    src = _u8(src)
    sh, sw = src.shape[:2]
    out = np.empty((dh, dw, 4), np.uint8)
    _load().orc_rescale_bilinear(_p(src), fmt, sw, sh, _p(out), dw, dh)
    return out


# ------------------------------------------------------------------ compositor
def pack_layouts(layouts: Sequence[Layout], struct=_Layout, mask_struct=_Mask):
    arr = (struct * max(len(layouts), 1))()
    for i, L in enumerate(layouts):
        s = arr[i]
        s.top, s.left, s.width, s.height = L.top, L.left, L.width, L.height
        s.rotation_degrees = L.rotation_degrees
        s.border_radius[:] = list(L.border_radius)
        s.type = L.type
        s.source_index = L.source_index
        s.color[:] = list(L.color)
        s.border_color[:] = list(L.border_color)
        s.border_width = L.border_width
        s.crop[:] = list(L.crop)
        s.blur_radius = L.blur_radius
        s.masks_len = len(L.masks)
        for j, m in enumerate(L.masks[:MAX_MASKS]):
            s.masks[j].radius[:] = list(m.radius)
            s.masks[j].top, s.masks[j].left, s.masks[j].width, s.masks[j].height = m.top, m.left, m.width, m.height
    return arr


def layout_fragments(W, H, layout: Layout) -> np.ndarray:
    """Fragment-stage output (f32, premultiplied) of one colour / shadow layout per pixel; NaN where the quad does not cover."""
    arr = pack_layouts([layout])
    out = np.empty((H, W, 4), np.float32)
    lib = _load()
    lib.orc_layout_fragments.restype = None
    lib.orc_layout_fragments.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.orc_layout_fragments(out.ctypes.data, W, H, C.addressof(arr))
    return out


def apply_layouts(W, H, layouts: Sequence[Layout], sources: Sequence[Optional[np.ndarray]], srgb=True, omp=False) -> np.ndarray:
    arr = pack_layouts(layouts)
    srcs = (_Source * max(len(sources), 1))()
    keep = []
    for i, s in enumerate(sources):
        if s is None:
            srcs[i].data, srcs[i].w, srcs[i].h = None, 1, 1
        else:
            s = _u8(s)
            keep.append(s)
            srcs[i].data, srcs[i].w, srcs[i].h = s.ctypes.data, s.shape[1], s.shape[0]
    out = np.zeros((H, W, 4), np.uint8)
    _load(omp).orc_apply_layouts(_p(out), W, H, arr, len(layouts), srcs, len(sources), int(srgb))
    return out


@dataclass
class Glyph:
    dst_x: int
    dst_y: int
    w: int
    h: int
    atlas_x: int
    atlas_y: int
    color: Sequence[float]


def render_frame_yuv420(planes, layouts: Sequence[Layout], sources, W: int, H: int, omp=False):
    """One output frame of the reference's pass sequence in one C call (orc_render_frame_yuv420): `planes` = [(y, u, v)] of equally
    sized 4:2:0 inputs, sources[i] = an int (input index: its node texture), an RGBA8 array (text / image node) or None."""
    n_in = len(planes)
    ih, iw = planes[0][0].shape
    keep = [[_u8(p[k]) for k in range(3)] for p in planes]
    ptrs = [(C.c_void_p * n_in)(*[kp[k].ctypes.data for kp in keep]) for k in range(3)]
    arr = pack_layouts(layouts)
    srcs = (_Source * max(len(sources), 1))()
    src_in = (C.c_int * max(len(sources), 1))()
    for i, s in enumerate(sources):
        src_in[i] = -1
        if s is None:
            srcs[i].data, srcs[i].w, srcs[i].h = None, 1, 1
        elif isinstance(s, (int, np.integer)):
            srcs[i].data, srcs[i].w, srcs[i].h = None, iw, ih
            src_in[i] = int(s)
        else:
            s = _u8(s)
            keep.append(s)
            srcs[i].data, srcs[i].w, srcs[i].h = s.ctypes.data, s.shape[1], s.shape[0]
    oy, ou, ov = np.empty((H, W), np.uint8), np.empty((H // 2, W // 2), np.uint8), np.empty((H // 2, W // 2), np.uint8)
    rc = _load(omp).orc_render_frame_yuv420(ptrs[0], ptrs[1], ptrs[2], n_in, iw, ih, arr, len(layouts), srcs, src_in, len(sources), W, H,
                                            _p(oy), _p(ou), _p(ov))
    if rc < 0:
        raise RuntimeError(f"orc_render_frame_yuv420: {rc}")
    return [oy, ou, ov]


def blit_glyphs(W, H, bg, glyphs: Sequence[Glyph], atlas, srgb=True) -> np.ndarray:
    atlas = _u8(atlas)
    garr = (_Glyph * max(len(glyphs), 1))()
    for i, g in enumerate(glyphs):
        garr[i].dst_x, garr[i].dst_y, garr[i].w, garr[i].h = g.dst_x, g.dst_y, g.w, g.h
        garr[i].atlas_x, garr[i].atlas_y = g.atlas_x, g.atlas_y
        garr[i].color[:] = list(g.color)
    out = np.zeros((H, W, 4), np.uint8)
    bgc = (C.c_float * 4)(*[float(x) for x in bg])
    _load().orc_blit_glyphs(_p(out), W, H, bgc, garr, len(glyphs), _p(atlas), atlas.shape[1], atlas.shape[0], int(srgb))
    return out


def gaussian_blur(src, sigma, fmt=PX_RGBA8_SRGB) -> np.ndarray:
    src = _u8(src)
    h, w = src.shape[:2]
    tmp = np.empty_like(src)
    out = np.empty_like(src)
    _load().orc_gaussian_blur(_p(src), fmt, w, h, float(sigma), _p(tmp), _p(out))
    return out


SHADER_GAUSSIAN_BLUR, SHADER_GRADIENT, SHADER_RED_BORDER, SHADER_CIRCLE_LAYOUT = 0, 1, 2, 3
SHADER_FADE_TO_BALL, SHADER_LAYOUT_PLANES, SHADER_COLOR_BY_TEXTURE_COUNT, SHADER_SILLY = 4, 5, 6, 7


def circle_layout_params(circles) -> bytes:
    """circles: [(left_px, top_px, width_px, height_px, (r, g, b, a))] -> the bytes ShaderParam::to_bytes gives for the
    reference's list-of-struct parameter (types/shader.rs), 32 B per entry."""
    import struct
    return b"".join(struct.pack("<4I4f", int(l), int(t), int(w), int(h), *[float(c) for c in bg]) for (l, t, w, h, bg) in circles)


def builtin_shader(shader_id: int, sources, W: int, H: int, params: bytes = b"", time: float = 0.0, srgb: bool = True) -> np.ndarray:
    """One ShaderNode render of a built-in port of the reference's WGSL shaders (ids above) over RGBA8 sources -> HxWx4."""
    srcs = [_u8(s) for s in sources]
    arr = (_Source * max(1, len(srcs)))()
    for i, s in enumerate(srcs):
        arr[i].data, arr[i].w, arr[i].h = _p(s), s.shape[1], s.shape[0]
    out = np.empty((H, W, 4), np.uint8)
    buf = C.create_string_buffer(bytes(params), max(1, len(params)))
    rc = _load().orc_builtin_shader(int(shader_id), arr, len(srcs), C.cast(buf, C.c_void_p), float(time), int(srgb), W, H, _p(out))
    if rc != 0:
        raise ValueError(f"unknown built-in shader id {shader_id}")
    return out
