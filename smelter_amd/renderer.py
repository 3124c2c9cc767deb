"""`Renderer` of the reference (smelter-render/src/state.rs:96-252) on the HIP library: register inputs / images / shaders,
update_scene(output, resolution, format, scene JSON), render(frame set) -> output frames in HBM.

The implementation is C++ (smelter_amd/csrc/host/renderer.cpp behind `smr_renderer_*`, include/smr.h); this is the ctypes
binding used by bench.py, smoke() and the tests.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _ffi
from .hip import Context, DeviceFrame, FRAME_PLANAR_YUV420
from .scene import Node, SceneError


class BorrowedFrame(DeviceFrame):
    """An output frame owned by the renderer (valid until the render after the next)."""

    def __init__(self, ctx: Context, c_frame: _ffi.Frame):  # noqa: super().__init__ would allocate
        self.ctx, self.fmt, self.w, self.h = ctx, c_frame.format, c_frame.width, c_frame.height
        self.c = c_frame

    def destroy(self):
        pass


class Renderer:
    def __init__(self, ctx: Context, stream_fallback_timeout_s: float = 0.5, lanes: Sequence[Context] = (), max_outputs: int = 16):
        """`lanes`: extra contexts on the same device — consecutive frames rotate through `ctx` and these, so up to
        1 + len(lanes) frames are in flight on the GPU while the scene stays one state (smr_renderer_add_lane).
        `max_outputs`: how many output frames one render call can hand back (the `cap` of smr_renderer_render)."""
        self.ctx, self.lib = ctx, ctx.lib
        h = C.c_void_p()
        if self.lib.smr_renderer_create(ctx.handle, int(stream_fallback_timeout_s * 1e9), C.byref(h)) != 0:
            raise RuntimeError("smr_renderer_create failed")
        self._h = h
        self._max_outputs = max(int(max_outputs), 1)
        self._outs = (_ffi.OutputFrame * self._max_outputs)()
        self._ctx_of = {ctx.handle.value: ctx}
        for c in lanes:
            self._check(self.lib.smr_renderer_add_lane(self._h, c.handle))
            self._ctx_of[c.handle.value] = c

    def sync(self):
        """Waits for the frames in flight on every lane."""
        self._check(self.lib.smr_renderer_sync(self._h))

    def output(self, i: int = 0) -> "BorrowedFrame":
        """Output `i` of the last render call, bound to the context (lane) that produced it."""
        return BorrowedFrame(self._ctx_of[self._outs[i].ctx], self._outs[i].frame.contents)

    def close(self):
        if self._h:
            self.lib.smr_renderer_destroy(self._h)
            self._h = None

    def _check(self, rc: int) -> int:
        if rc < 0:
            raise SceneError(self.lib.smr_renderer_last_error(self._h).decode())
        return rc

    # -- registry (state.rs:102-171)
    def register_input(self, input_id: str):
        self._check(self.lib.smr_renderer_register_input(self._h, input_id.encode()))

    def unregister_input(self, input_id: str):
        self._check(self.lib.smr_renderer_unregister_input(self._h, input_id.encode()))

    def register_image(self, image_id: str, rgba_straight: np.ndarray):
        px = np.ascontiguousarray(rgba_straight, dtype=np.uint8)
        h, w = px.shape[:2]
        self._check(self.lib.smr_renderer_register_image(self._h, image_id.encode(), px.ctypes.data, w, h))

    def set_text_measurer(self, measurer):
        """The caller's text shaper for Text nodes without explicit width / height (tests.text_twin.Shaper(...).measurer)."""
        self._measurer = measurer  # keep the callback alive
        self._check(self.lib.smr_renderer_set_text_measurer(self._h, measurer if measurer is not None else _ffi.TEXT_MEASURE_FN(0), None))

    def set_fontbook(self, book):
        """A smelter_amd.fontbook.NativeFontBook (smr_fontbook): the renderer measures and draws every Text node itself at update_scene."""
        self._fontbook = book  # keep it alive: the renderer does not own it
        self._check(self.lib.smr_renderer_set_fontbook(self._h, book.handle if book is not None else None))

    def register_shader(self, shader_id: str, builtin_id: int = _ffi.SHADER_GAUSSIAN_BLUR):
        self._check(self.lib.smr_renderer_register_shader(self._h, shader_id.encode(), builtin_id))

    # -- scenes (state.rs:177-189)
    def update_scene(self, output_id: str, width: int, height: int, scene: Union[str, dict], output_format: int = FRAME_PLANAR_YUV420) -> List[Node]:
        text = scene if isinstance(scene, str) else json.dumps(scene)
        self._check(self.lib.smr_renderer_update_scene(self._h, output_id.encode(), width, height, output_format, text.encode()))
        return self.nodes(output_id)

    def unregister_output(self, output_id: str):
        self._check(self.lib.smr_renderer_unregister_output(self._h, output_id.encode()))

    def nodes(self, output_id: str) -> List[Node]:
        out = []
        oid = output_id.encode()
        for i in range(self._check(self.lib.smr_renderer_node_count(self._h, oid))):
            info = _ffi.SceneNode()
            self._check(self.lib.smr_renderer_node_info(self._h, oid, i, C.byref(info)))
            out.append(Node(i, info.kind, info.parent, [], info.width, info.height, (info.ref_id or b"").decode(),
                            (info.id or b"").decode(), (info.payload or b"").decode()))
        for n in out:
            if n.parent >= 0:
                out[n.parent].children.append(n.index)
        return out

    def set_text(self, output_id: str, node: int, glyphs: Sequence, atlas: np.ndarray, bg: Sequence[float] = (0.0, 0.0, 0.0, 0.0)):
        """The glyph run of a Text node (shaping / rasterisation is the caller's); to be supplied once after every update_scene."""
        atlas = np.ascontiguousarray(atlas, dtype=np.uint8)
        garr = (_ffi.Glyph * max(len(glyphs), 1))()
        for i, g in enumerate(glyphs):
            garr[i].dst_x, garr[i].dst_y, garr[i].w, garr[i].h = g.dst_x, g.dst_y, g.w, g.h
            garr[i].atlas_x, garr[i].atlas_y = g.atlas_x, g.atlas_y
            garr[i].color[:] = list(g.color)
        bgc = (C.c_float * 4)(*[float(x) for x in bg])
        self._check(self.lib.smr_renderer_set_text(self._h, output_id.encode(), node, bgc, garr, len(glyphs), atlas.ctypes.data,
                                                   atlas.shape[1], atlas.shape[0]))

    # -- per frame (state.rs:173, 220-252)
    def make_frame_set(self, frames: Dict[str, DeviceFrame], pts_s: Optional[float] = None, frame_pts_s: Optional[Dict[str, float]] = None):
        """Pre-packs a FrameSet (ctypes array) for `render_packed`; keeps the DeviceFrames alive."""
        arr = (_ffi.InputFrame * max(len(frames), 1))()
        keep = []
        for i, (k, f) in enumerate(frames.items()):
            key = k.encode()
            keep.append((key, f))
            arr[i].input_id = key
            arr[i].frame = C.pointer(f.c)
            arr[i].pts_ns = int((frame_pts_s or {}).get(k, pts_s or 0.0) * 1e9)
        return arr, len(frames), keep

    def render_packed(self, pts_ns: int, packed) -> int:
        """One frame for every output; returns the number of outputs (frames are read with `output`)."""
        arr, n, _ = packed
        cnt = C.c_uint32()
        self._check(self.lib.smr_renderer_render(self._h, pts_ns, arr, n, self._outs, self._max_outputs, C.byref(cnt)))
        return cnt.value

    def render(self, pts_s: float, frames: Dict[str, DeviceFrame], frame_pts_s: Optional[Dict[str, float]] = None) -> Dict[str, BorrowedFrame]:
        packed = self.make_frame_set(frames, pts_s, frame_pts_s)
        n = self.render_packed(int(pts_s * 1e9), packed)
        return {self._outs[i].output_id.decode(): self.output(i) for i in range(min(n, self._max_outputs))}
