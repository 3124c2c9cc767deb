"""Thin object layer over the C ABI (include/smr.h) used by tests, bench.py and the host mirror.

Everything here calls straight into libsmr_hip.so through ctypes; numpy arrays only carry
host bytes in and out.  There is no CPU implementation behind these classes.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple, List, Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import (FRAME_ARGB, FRAME_BGRA, FRAME_NV12, FRAME_PLANAR_YUV420, FRAME_PLANAR_YUV422, FRAME_PLANAR_YUV444,
                   FRAME_PLANAR_YUVJ420, FRAME_RGBA, FRAME_UYVY422, FRAME_YUYV422, MODE_CPU_OPTIMIZED, MODE_GPU_OPTIMIZED,
                   PX_R8, PX_RG8, PX_RGBA8, PX_RGBA16F)
from ._ffi import (SHADER_CIRCLE_LAYOUT, SHADER_COLOR_BY_TEXTURE_COUNT, SHADER_FADE_TO_BALL, SHADER_GAUSSIAN_BLUR,  # noqa: F401
                   SHADER_GRADIENT, SHADER_LAYOUT_PLANES, SHADER_MAX_SOURCES, SHADER_RED_BORDER, SHADER_SILLY)

STAGE_NAMES = {0: "ingest", 1: "resample", 2: "layouts", 3: "output", 4: "fused_ingest_resample", 5: "fused_compose_output"}


def lab_build() -> bool:
    """smr_build_flags() & 1: a laboratory build of the library (tools/variant.sh, SMR_LIB=...)."""
    return bool(_ffi.load().smr_build_flags() & 1)


class SmrError(RuntimeError):
    """Mirrors RenderSceneError::WgpuError(WgpuError::{Validation, OutOfMemory, Internal}(String))."""

    def __init__(self, code: int, message: str):
        kind = {-1: "Validation", -2: "OutOfMemory", -3: "Internal"}.get(code, str(code))
        super().__init__(f"{kind}: {message}")
        self.code = code


class Surface:
    def __init__(self, ctx: "Context", handle, w: int, h: int, fmt: int):
        self.ctx, self.handle, self.w, self.h, self.fmt = ctx, handle, w, h, fmt

    def _shape_dtype(self):
        if self.fmt == PX_RGBA8:
            return (self.h, self.w, 4), np.uint8
        if self.fmt == PX_RGBA16F:
            return (self.h, self.w, 4), np.uint16
        if self.fmt == PX_RG8:
            return (self.h, self.w, 2), np.uint8
        return (self.h, self.w), np.uint8

    def upload(self, arr) -> "Surface":
        shape, dt = self._shape_dtype()
        a = np.ascontiguousarray(arr, dtype=dt).reshape(shape)
        self.ctx._check(self.ctx.lib.smr_surface_upload(self.ctx.handle, self.handle, a.ctypes.data, 0))
        return self

    def download(self) -> np.ndarray:
        shape, dt = self._shape_dtype()
        out = np.empty(shape, dt)
        self.ctx._check(self.ctx.lib.smr_surface_download(self.ctx.handle, self.handle, out.ctypes.data, 0))
        return out

    def info(self) -> _ffi.SurfaceInfo:
        inf = _ffi.SurfaceInfo()
        self.ctx._check(self.ctx.lib.smr_surface_info_get(self.handle, C.byref(inf)))
        return inf

    def destroy(self):
        if self.handle:
            self.ctx.lib.smr_surface_destroy(self.ctx.handle, self.handle)
            self.handle = None


class DeviceFrame:
    """A video frame resident in HBM (smr_frame)."""

    def __init__(self, ctx: "Context", fmt: int, w: int, h: int):
        self.ctx, self.fmt, self.w, self.h = ctx, fmt, w, h
        self.c = _ffi.Frame()
        ctx._check(ctx.lib.smr_frame_create(ctx.handle, fmt, w, h, C.byref(self.c)))

    def plane_shapes(self):
        w, h, f = self.w, self.h, self.fmt
        if f in (FRAME_PLANAR_YUV420, FRAME_PLANAR_YUVJ420):
            return [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
        if f == FRAME_PLANAR_YUV422:
            return [(h, w), (h, w // 2), (h, w // 2)]
        if f == FRAME_PLANAR_YUV444:
            return [(h, w), (h, w), (h, w)]
        if f == FRAME_NV12:
            return [(h, w), (h // 2, w // 2, 2)]
        if f in (FRAME_UYVY422, FRAME_YUYV422):
            return [(h, w // 2, 4)]
        return [(h, w, 4)]

    def upload(self, planes: Sequence[np.ndarray]) -> "DeviceFrame":
        shapes = self.plane_shapes()
        keep = [np.ascontiguousarray(p, dtype=np.uint8).reshape(s) for p, s in zip(planes, shapes)]
        ptrs = (C.c_void_p * 3)(*[k.ctypes.data for k in keep] + [None] * (3 - len(keep)))
        self.ctx._check(self.ctx.lib.smr_frame_upload(self.ctx.handle, C.byref(self.c), ptrs))
        return self

    def download(self) -> List[np.ndarray]:
        outs = [np.empty(s, np.uint8) for s in self.plane_shapes()]
        ptrs = (C.c_void_p * 3)(*[o.ctypes.data for o in outs] + [None] * (3 - len(outs)))
        self.ctx._check(self.ctx.lib.smr_frame_download(self.ctx.handle, C.byref(self.c), ptrs))
        return outs

    def pinned_planes(self) -> List[np.ndarray]:
        """Host plane buffers in pinned memory (smr_host_alloc), shaped like this frame's planes; freed with the context."""
        return [self.ctx.host_array(s) for s in self.plane_shapes()]

    def upload_async(self, pinned: Sequence[np.ndarray]):
        """Stream-ordered copy from pinned host planes: returns at once, the buffers must stay untouched until ctx.sync()."""
        ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in pinned] + [None] * (3 - len(pinned)))
        self.ctx._check(self.ctx.lib.smr_frame_upload_async(self.ctx.handle, C.byref(self.c), ptrs))

    def download_async(self, pinned: Sequence[np.ndarray]):
        ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in pinned] + [None] * (3 - len(pinned)))
        self.ctx._check(self.ctx.lib.smr_frame_download_async(self.ctx.handle, C.byref(self.c), ptrs))

    def destroy(self):
        self.ctx.lib.smr_frame_destroy(self.ctx.handle, C.byref(self.c))


def pack_layouts(layouts) -> "C.Array":
    """Accepts Layout records (smelter_amd.scene.Layout or any object with the same fields) and fills smr_layout[]."""
    arr = (_ffi.Layout * max(len(layouts), 1))()
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
        for j, m in enumerate(L.masks[: _ffi.MAX_MASKS]):
            s.masks[j].radius[:] = list(m.radius)
            s.masks[j].top, s.masks[j].left, s.masks[j].width, s.masks[j].height = m.top, m.left, m.width, m.height
    return arr


INGEST_AUTO, INGEST_VALU_F32, INGEST_MFMA_F16, INGEST_MFMA_F16_NODE, INGEST_MFMA_F16_FUSED = 0, 1, 2, 4, 5  # (3: retired)
CONVERT_AUTO, CONVERT_GENERAL, CONVERT_BLOCK_4X2 = 0, 1, 2
KERNEL_NAMES = ("ingest_wave", "ingest_wave_rgba", "frame_to_rgba_420", "ingest_valu", "resample_general", "frame_to_rgba", "compose_output", "apply_layouts")
COMM_ID_BYTES = 128


class Comm:
    """smr_comm: the tile gather of the multi-GPU path behind the C ABI (include/smr.h).

    Comm.local([ctx0, ctx1, ...])        one process, one context per device (peer copies)
    Comm.rank(ctx, world, rank, id)      one process per GPU (RCCL); id = Comm.unique_id() made on one rank and distributed by the host
    gather(root, owners, src, dst)       stream-ordered: dst[i] on the root holds tile i (produced in src[i] on rank owners[i])"""

    def __init__(self, lib, handle, ctxs):
        self.lib, self.handle, self.ctxs = lib, handle, ctxs

    @staticmethod
    def unique_id() -> bytes:
        lib = _ffi.load()
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        if lib.smr_comm_unique_id(buf) != 0:
            raise SmrError(-3, "smr_comm_unique_id failed (is librccl.so present?)")
        return bytes(buf)

    @staticmethod
    def local(ctxs: Sequence["Context"]) -> "Comm":
        lib = _ffi.load()
        arr = (C.c_void_p * len(ctxs))(*[c.handle.value for c in ctxs])
        h = C.c_void_p()
        rc = lib.smr_comm_create_local(arr, len(ctxs), C.byref(h))
        if rc != 0:
            raise SmrError(rc, lib.smr_last_error(ctxs[0].handle).decode())
        return Comm(lib, h, list(ctxs))

    @staticmethod
    def rank(ctx: "Context", world: int, rank: int, uid: bytes) -> "Comm":
        lib = _ffi.load()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        h = C.c_void_p()
        rc = lib.smr_comm_create_rank(ctx.handle, world, rank, buf, C.byref(h))
        if rc != 0:
            raise SmrError(rc, lib.smr_last_error(ctx.handle).decode())
        return Comm(lib, h, [ctx])

    @property
    def world(self) -> int:
        return self.lib.smr_comm_world(self.handle)

    def gather(self, root: int, owners: Sequence[int], src: Sequence[Optional["Surface"]], dst: Sequence[Optional["Surface"]]):
        n = len(owners)
        own = (C.c_uint32 * max(n, 1))(*owners)
        s = (C.c_void_p * max(n, 1))(*[(x.handle.value if x is not None else None) for x in src])
        d = (C.c_void_p * max(n, 1))(*[(x.handle.value if x is not None else None) for x in dst])
        rc = self.lib.smr_gather_tiles(self.handle, root, own, s, d, n)
        if rc != 0:
            raise SmrError(rc, self.lib.smr_comm_last_error(self.handle).decode())

    def close(self):
        if self.handle:
            self.lib.smr_comm_destroy(self.handle)
            self.handle = None
OPT_INGEST_IMPL, OPT_INGEST_STRIP_WIDTH, OPT_DIRECT_OUTPUT, OPT_CONVERT_IMPL, OPT_COMPACT_NODES, OPT_FUSED_KERNELS, OPT_COMPOSE_SELECT, OPT_SHARED_DEVICE, OPT_PLANE_SOURCE = 0, 1, 2, 3, 4, 5, 6, 7, 8


class Context:
    def __init__(self, device: int = 0, mode: int = MODE_GPU_OPTIMIZED, max_layouts: int = 100, stream: Optional[int] = None):
        self.lib = _ffi.load()
        h = C.c_void_p()
        rc = self.lib.smr_ctx_create(device, mode, max_layouts, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise SmrError(rc, f"smr_ctx_create(device={device}) failed — is a HIP device visible?")
        self.handle = h
        self.mode = mode
        self.device = device
        self.stream_handle = stream  # the caller's hipStream_t, or None when the context created its own
        self._pinned = []

    # -- plumbing
    def _check(self, rc: int) -> int:
        if rc < 0:
            raise SmrError(rc, self.lib.smr_last_error(self.handle).decode())
        return rc

    def close(self):
        if self.handle:
            for p in self._pinned:
                self.lib.smr_host_free(self.handle, p)
            self._pinned = []
            self.lib.smr_ctx_destroy(self.handle)
            self.handle = None

    def host_array(self, shape) -> np.ndarray:
        """uint8 array in pinned host memory (smr_host_alloc); lives until close()."""
        n = int(np.prod(shape))
        p = C.c_void_p()
        self._check(self.lib.smr_host_alloc(self.handle, max(n, 1), C.byref(p)))
        self._pinned.append(p)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].reshape(shape)

    def sync(self):
        self._check(self.lib.smr_sync(self.handle))

    def set_option(self, option: int, value: int):
        self._check(self.lib.smr_ctx_set_option(self.handle, option, value))

    def set_ingest_impl(self, impl: int):
        """INGEST_AUTO (= INGEST_MFMA_F16 = INGEST_MFMA_F16_NODE: exact converter into the node texture, matrix cores for the resample — within
        1 LSB end to end on every content) / INGEST_VALU_F32 (bit-identical to the pass-per-launch kernels) / INGEST_MFMA_F16_FUSED (opt-in: the
        matrix-core kernel converts planar 4:2:0 / NV12 on the fly, within one code per stage) — SMR_OPT_INGEST_IMPL."""
        self.set_option(OPT_INGEST_IMPL, impl)

    def set_convert_impl(self, impl: int):
        """CONVERT_AUTO (block converters) / CONVERT_GENERAL (one kernel per WGSL pass) / CONVERT_BLOCK_4X2 (round 3's) — SMR_OPT_CONVERT_IMPL."""
        self.set_option(OPT_CONVERT_IMPL, impl)

    def set_compact_nodes(self, on: bool):
        """SMR_OPT_COMPACT_NODES: node textures only the matrix-core resampler reads as RGB12 instead of RGBA8 (default on)."""
        self.set_option(OPT_COMPACT_NODES, 1 if on else 0)

    def set_fused_kernels(self, on: bool):
        """SMR_OPT_FUSED_KERNELS: waves A / B (default) or one general kernel per pass of the reference (what the tests hold the fused kernels to)."""
        self.set_option(OPT_FUSED_KERNELS, 1 if on else 0)

    def set_compose_select(self, on: bool):
        """SMR_OPT_COMPOSE_SELECT: seam tiles of a grid of opaque 1:1 layers as per-pixel copies (default) or through the compositing path."""
        self.set_option(OPT_COMPOSE_SELECT, 1 if on else 0)

    def lab_build(self) -> bool:
        """True for a laboratory build of the library (-DSMR_LAB): the fused-conversion route (INGEST_MFMA_F16_FUSED) exists only there."""
        return lab_build()

    def set_plane_source(self, on: bool):
        """SMR_OPT_PLANE_SOURCE: 4:2:0 frames are converted inside the matrix-core resampler (exactly, through LDS; default on) or by the converter
        kernel into a node texture in memory (off) — the same pixels."""
        self.set_option(OPT_PLANE_SOURCE, 1 if on else 0)

    def set_direct_output(self, on: bool):
        """SMR_OPT_DIRECT_OUTPUT: let the resampling kernel write Y'CbCr for the compositor's copy tiles of a scene at rest (default off)."""
        self.set_option(OPT_DIRECT_OUTPUT, 1 if on else 0)

    def set_strip_width(self, tw: int):
        """0 (automatic), 32 or 64: strip width of the f32 ingest kernel (SMR_OPT_INGEST_STRIP_WIDTH)."""
        self.set_option(OPT_INGEST_STRIP_WIDTH, tw)

    def timer_start(self):
        self._check(self.lib.smr_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self.lib.smr_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def profile_enable(self, on: bool):
        self._check(self.lib.smr_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        self._check(self.lib.smr_profile_reset(self.handle))

    def kernel_launches(self) -> dict:
        """Launch counts per kernel since the context was created (smr_debug_kernel_launches; names of smr_kernel_id)."""
        out = {}
        for k, name in enumerate(KERNEL_NAMES):
            n = C.c_uint64()
            self._check(self.lib.smr_debug_kernel_launches(self.handle, k, C.byref(n)))
            out[name] = n.value
        return out

    def profile_read(self):
        out = {}
        for sid, name in STAGE_NAMES.items():
            ms, n = C.c_float(), C.c_uint32()
            self._check(self.lib.smr_profile_read(self.handle, sid, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    # -- resources
    def surface(self, w: int, h: int, fmt: int = PX_RGBA8) -> Surface:
        h_ = C.c_void_p()
        self._check(self.lib.smr_surface_create(self.handle, w, h, fmt, C.byref(h_)))
        return Surface(self, h_, w, h, fmt)

    def wrap(self, dptr: int, pitch: int, w: int, h: int, fmt: int = PX_RGBA8) -> Surface:
        h_ = C.c_void_p()
        self._check(self.lib.smr_surface_wrap(self.handle, C.c_void_p(dptr), pitch, w, h, fmt, C.byref(h_)))
        return Surface(self, h_, w, h, fmt)

    def surface_from(self, arr, fmt: int = PX_RGBA8) -> Surface:
        a = np.asarray(arr)
        return self.surface(a.shape[1], a.shape[0], fmt).upload(a)

    def wrapped_frame(self, fmt: int, w: int, h: int, planes: Sequence[Tuple[int, int]]) -> DeviceFrame:
        """A frame over device memory the caller owns (a decoder's output, torch tensors): `planes` = (device pointer, pitch in bytes) per
        plane, in the order of smr_frame.planes; each is wrapped in place (smr_surface_wrap: nothing is copied, nothing is owned)."""
        f = DeviceFrame.__new__(DeviceFrame)
        f.ctx, f.fmt, f.w, f.h = self, fmt, w, h
        f.c = _ffi.Frame()
        f.c.format, f.c.width, f.c.height = fmt, w, h
        shapes = f.plane_shapes()
        assert len(shapes) == len(planes), "one (pointer, pitch) pair per plane"
        f.surfaces = []
        for i, ((dptr, pitch), shape) in enumerate(zip(planes, shapes)):
            pf = PX_RGBA8 if len(shape) == 3 and shape[2] == 4 else PX_RG8 if len(shape) == 3 else PX_R8
            s = self.wrap(dptr, pitch, shape[1], shape[0], pf)
            f.surfaces.append(s)  # (kept alive with the frame)
            f.c.planes[i] = s.handle.value if hasattr(s.handle, "value") else s.handle
        return f

    def frame(self, fmt: int, w: int, h: int, planes: Optional[Sequence[np.ndarray]] = None) -> DeviceFrame:
        f = DeviceFrame(self, fmt, w, h)
        if planes is not None:
            f.upload(planes)
        return f

    # -- passes
    def frame_to_rgba(self, frame: DeviceFrame, node: Optional[Surface] = None) -> Surface:
        node = node or self.surface(frame.w, frame.h, PX_RGBA8)
        self._check(self.lib.smr_frame_to_rgba(self.handle, C.byref(frame.c), node.handle))
        return node

    def add_premultiplied_alpha(self, src: Surface, dst: Optional[Surface] = None) -> Surface:
        dst = dst or self.surface(src.w, src.h)
        self._check(self.lib.smr_add_premultiplied_alpha(self.handle, src.handle, dst.handle))
        return dst

    def remove_premultiplied_alpha(self, src: Surface, dst: Optional[Surface] = None) -> Surface:
        dst = dst or self.surface(src.w, src.h)
        self._check(self.lib.smr_remove_premultiplied_alpha(self.handle, src.handle, dst.handle))
        return dst

    def rgba_to_frame(self, node: Surface, fmt: int, out: Optional[DeviceFrame] = None) -> DeviceFrame:
        out = out or self.frame(fmt, node.w, node.h)
        self._check(self.lib.smr_rgba_to_frame(self.handle, node.handle, C.byref(out.c)))
        return out

    def fill_black(self, out: DeviceFrame):
        self._check(self.lib.smr_frame_fill_black(self.handle, C.byref(out.c)))

    def resample_plan(self, src_w, src_h, crop, dst_w, dst_h) -> _ffi.ResamplePlan:
        p = _ffi.ResamplePlan()
        c = (C.c_float * 4)(*[float(x) for x in crop])
        rc = self.lib.smr_resample_plan_make(src_w, src_h, c, dst_w, dst_h, C.byref(p))
        if rc < 0:
            raise SmrError(rc, "smr_resample_plan_make")
        return p

    def resample(self, src: Surface, crop, dst: Surface) -> int:
        c = (C.c_float * 4)(*[float(x) for x in crop])
        return self._check(self.lib.smr_resample(self.handle, src.handle, c, dst.handle))

    def resample_pass(self, src: Surface, axis: int, scale: float, offset: float, perp_offset: int, dst: Surface):
        self._check(self.lib.smr_resample_pass(self.handle, src.handle, axis, scale, offset, perp_offset, dst.handle))

    def downsample(self, src: Surface, fx: int, fy: int, dst: Surface):
        self._check(self.lib.smr_downsample(self.handle, src.handle, fx, fy, dst.handle))

    def frame_preprocess(self, frame: DeviceFrame, size: Optional[tuple] = None) -> np.ndarray:
        """FramePreProcessor::process_to_bytes: frame -> RGBA8 node texture -> optional bilinear rescale to `size` = (w, h) -> bytes."""
        w, h = size if size else (frame.w, frame.h)
        out = np.empty((h, w, 4), np.uint8)
        self._check(self.lib.smr_frame_preprocess(self.handle, C.byref(frame.c), (size or (0, 0))[0], (size or (0, 0))[1], out.ctypes.data, 0))
        return out

    def rescale_bilinear(self, src: Surface, dst: Surface):
        self._check(self.lib.smr_rescale_bilinear(self.handle, src.handle, dst.handle))

    def apply_layouts(self, target: Surface, layouts, sources: Sequence[Optional[Surface]]):
        arr = pack_layouts(layouts)
        n_src = len(sources)
        ptrs = (C.c_void_p * max(n_src, 1))(*[(s.handle if s is not None else None) for s in sources])
        self._check(self.lib.smr_apply_layouts(self.handle, target.handle, arr, len(layouts), ptrs, n_src))

    def render_layouts(self, layouts, sources, out_w: int, out_h: int, out: Optional[DeviceFrame] = None,
                       out_rgba: Optional[Surface] = None, packed=None):
        """sources: sequence of Surface | DeviceFrame | None."""
        arr = packed if packed is not None else pack_layouts(layouts)
        n = len(layouts)
        srcs = (_ffi.Source * max(len(sources), 1))()
        for i, s in enumerate(sources):
            if s is None:
                srcs[i].kind = _ffi.SOURCE_NONE
            elif isinstance(s, DeviceFrame):
                srcs[i].kind = _ffi.SOURCE_FRAME
                srcs[i].frame = C.pointer(s.c)
            else:
                srcs[i].kind = _ffi.SOURCE_OPAQUE_SURFACE if getattr(s, "opaque", False) else _ffi.SOURCE_SURFACE
                srcs[i].surface = s.handle
        self._check(self.lib.smr_render_layouts(self.handle, arr, n, srcs, len(sources), out_w, out_h,
                                                C.byref(out.c) if out is not None else None,
                                                out_rgba.handle if out_rgba is not None else None))

    def ingest_resample_batch(self, frames: Sequence[DeviceFrame], crops, dsts: Sequence[Surface]) -> List[int]:
        """All inputs of a shard in one launch of the ingest kernel; returns the plan kind per input (0 = direct)."""
        n = len(frames)
        fp = (C.POINTER(_ffi.Frame) * max(n, 1))(*[C.pointer(f.c) for f in frames])
        cr = (C.c_float * max(4 * n, 1))(*[float(x) for c in crops for x in c])
        dp = (C.c_void_p * max(n, 1))(*[d.handle for d in dsts])
        kinds = (C.c_int * max(n, 1))()
        self._check(self.lib.smr_ingest_resample_batch(self.handle, fp, cr, dp, n, kinds))
        return list(kinds[:n])

    def ingest_resample(self, frame: DeviceFrame, crop, dst: Surface) -> int:
        c = (C.c_float * 4)(*[float(x) for x in crop])
        return self._check(self.lib.smr_ingest_resample(self.handle, C.byref(frame.c), c, dst.handle))

    def blit_glyphs(self, target: Surface, bg, glyphs, atlas: np.ndarray):
        atlas = np.ascontiguousarray(atlas, dtype=np.uint8)
        garr = (_ffi.Glyph * max(len(glyphs), 1))()
        for i, g in enumerate(glyphs):
            garr[i].dst_x, garr[i].dst_y, garr[i].w, garr[i].h = g.dst_x, g.dst_y, g.w, g.h
            garr[i].atlas_x, garr[i].atlas_y = g.atlas_x, g.atlas_y
            garr[i].color[:] = list(g.color)
        bgc = (C.c_float * 4)(*[float(x) for x in bg])
        self._check(self.lib.smr_blit_glyphs(self.handle, target.handle, bgc, garr, len(glyphs), atlas.ctypes.data,
                                             atlas.shape[1], atlas.shape[0]))

    def gaussian_blur(self, src: Surface, sigma: float, dst: Optional[Surface] = None) -> Surface:
        dst = dst or self.surface(src.w, src.h)
        p = _ffi.GaussianBlurParams(sigma)
        ptrs = (C.c_void_p * 1)(src.handle)
        self._check(self.lib.smr_builtin_shader(self.handle, _ffi.SHADER_GAUSSIAN_BLUR, C.byref(p), C.sizeof(p), ptrs, 1,
                                                dst.handle, 0.0))
        return dst

    def builtin_shader(self, shader_id: int, sources, dst: Surface, params: bytes = b"", time_s: float = 0.0) -> Surface:
        """One ShaderNode render of a built-in port of the reference's in-tree WGSL shaders (include/smr.h smr_builtin_shader_id):
        `sources` are RGBA8 surfaces, `params` the bytes ShaderParam::to_bytes would bind at @group(1)."""
        ptrs = (C.c_void_p * max(1, len(sources)))(*[s.handle for s in sources])
        buf = C.create_string_buffer(bytes(params), max(1, len(params)))
        self._check(self.lib.smr_builtin_shader(self.handle, int(shader_id), C.cast(buf, C.c_void_p), len(params), ptrs, len(sources),
                                                dst.handle, float(time_s)))
        return dst
