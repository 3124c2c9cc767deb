"""Host scene engine binding: scene JSON -> render graph -> flattened smr_layout list per frame.

Mirrors the reference's `Renderer::update_scene` / `LayoutNode::render` up to the draw list
(smelter-render/src/state.rs:177-189, scene/scene_state.rs:74-127, transformations/layout.rs:176-184); the engine itself
is C++ (smelter_amd/csrc/host/scene*.cpp) behind the `smr_scene_*` entry points of include/smr.h.  No GPU is needed here.
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

from . import _ffi

MODE_GPU_OPTIMIZED, MODE_CPU_OPTIMIZED = 0, 1


class SceneError(ValueError):
    """TypeError / SceneError of the reference (smelter-api TypeError, scene.rs:188-231)."""


@dataclass
class Node:
    index: int
    kind: int            # _ffi.NODE_*
    parent: int
    children: List[int]
    width: int
    height: int
    ref_id: str          # input_id / image_id / shader_id
    id: str
    payload: str


class Scene:
    def __init__(self):
        self._lib = _ffi.load()
        h = C.c_void_p()
        if self._lib.smr_scene_create(C.byref(h)) != 0:
            raise SceneError("smr_scene_create failed")
        self._h = h

    def close(self):
        if self._h:
            self._lib.smr_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc < 0:
            raise SceneError(self._lib.smr_scene_last_error(self._h).decode())
        return rc

    def register_image(self, image_id: str, width: int, height: int):
        self._check(self._lib.smr_scene_register_image(self._h, image_id.encode(), width, height))

    def set_text_measurer(self, measurer):
        """`measurer`: an _ffi.TEXT_MEASURE_FN (e.g. tests.text_twin.Shaper(...).measurer) or None; sizes fitted Text nodes."""
        self._measurer = measurer  # keep the callback alive
        self._check(self._lib.smr_scene_set_text_measurer(self._h, measurer if measurer is not None else _ffi.TEXT_MEASURE_FN(0), None))

    def update(self, scene: Union[str, dict], out_w: int, out_h: int) -> List[Node]:
        text = scene if isinstance(scene, str) else json.dumps(scene)
        self._check(self._lib.smr_scene_update(self._h, text.encode(), out_w, out_h))
        return self.nodes()

    def parse(self, scene: Union[str, dict]) -> dict:
        """The smelter-api -> scene::Component conversion alone; returns the converted tree (canonical form)."""
        text = scene if isinstance(scene, str) else json.dumps(scene)
        out = C.c_char_p()
        self._check(self._lib.smr_scene_parse(self._h, text.encode(), C.byref(out)))
        return json.loads(out.value.decode())

    def nodes(self) -> List[Node]:
        out = []
        for i in range(self._check(self._lib.smr_scene_node_count(self._h))):
            info = _ffi.SceneNode()
            self._check(self._lib.smr_scene_node_info(self._h, i, C.byref(info)))
            kids = (C.c_int32 * max(1, info.n_children))()
            self._check(self._lib.smr_scene_node_children(self._h, i, kids, info.n_children))
            out.append(Node(i, info.kind, info.parent, list(kids[:info.n_children]), info.width, info.height,
                            (info.ref_id or b"").decode(), (info.id or b"").decode(), (info.payload or b"").decode()))
        return out

    def node_layouts(self, node: int, pts_ns: int, child_resolutions: Sequence[Optional[Tuple[int, int]]],
                     mode: int = MODE_GPU_OPTIMIZED, cap: int = 512):
        """-> (ctypes array of smr_layout, count, out_w, out_h) for layout node `node` at `pts_ns`."""
        n = len(child_resolutions)
        wh = (C.c_uint32 * max(1, 2 * n))()
        for i, r in enumerate(child_resolutions):
            wh[2 * i], wh[2 * i + 1] = (_ffi.NO_RESOLUTION, 0) if r is None else (int(r[0]), int(r[1]))
        arr = (_ffi.Layout * cap)()
        cnt, w, h = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self._lib.smr_scene_node_layouts(self._h, node, int(pts_ns), wh, n, mode, arr, cap, C.byref(cnt), C.byref(w), C.byref(h)))
        if cnt.value > cap:
            return self.node_layouts(node, pts_ns, child_resolutions, mode, cnt.value)
        return arr, cnt.value, w.value, h.value


    def layouts(self, node: int, pts_ns: int, child_resolutions, mode: int = MODE_GPU_OPTIMIZED) -> List["Layout"]:
        """node_layouts as plain Python records (what hip.pack_layouts / dist.ShardedCompositor take)."""
        arr, n, _, _ = self.node_layouts(node, pts_ns, child_resolutions, mode)
        return [Layout.from_c(arr[i]) for i in range(n)]


@dataclass
class Mask:
    radius: List[float]
    top: float
    left: float
    width: float
    height: float


@dataclass
class Layout:
    """smr_layout as a Python record (field names of include/smr.h)."""
    top: float
    left: float
    width: float
    height: float
    rotation_degrees: float
    border_radius: List[float]
    type: int
    source_index: int
    color: List[float]
    border_color: List[float]
    border_width: float
    crop: List[float]
    blur_radius: float
    masks: List[Mask]

    @staticmethod
    def from_c(s: "_ffi.Layout") -> "Layout":
        masks = [Mask(list(m.radius), m.top, m.left, m.width, m.height) for m in s.masks[: s.masks_len]]
        return Layout(s.top, s.left, s.width, s.height, s.rotation_degrees, list(s.border_radius), s.type, s.source_index,
                      list(s.color), list(s.border_color), s.border_width, list(s.crop), s.blur_radius, masks)


def cubic_bezier_easing(progress: float, x1: float, y1: float, x2: float, y2: float) -> float:
    return _ffi.load().smr_cubic_bezier_easing(progress, x1, y1, x2, y2)


def bounce_easing(progress: float) -> float:
    return _ffi.load().smr_bounce_easing(progress)


def parse_color(text: str) -> Tuple[int, int, int, int]:
    out = (C.c_uint8 * 4)()
    if _ffi.load().smr_parse_color(text.encode(), out) != 0:
        raise SceneError(f"invalid color {text!r}")
    return tuple(out)
