"""Scenes in transition: the reference's transition state machine and interpolation rules, restated for the oracle.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  Independent of the product's C++ scene engine: written from the Rust sources below,
operating on oracle/scene.py's component objects, so that the oracle picture of a scene in mid-transition is rendered from layouts that
never saw the engine under test.

Restates:
  scene/transition.rs:39-119                      TransitionState::new / state / is_finished, InterpolationKind::state
  scene/transition/bounce.rs, cubic_bezier.rs     the easing functions (cubic Bezier: first root of the cubic in [0, 1], then y(t))
  scene/types/interpolation.rs:10-101             ContinuousValue for f32 / f64 / Option / positions / Padding (f64 lerp, cast back)
  scene/components/interpolation.rs:8-91          Position, AbsolutePosition, BorderRadius, Vec<BoxShadow>, BoxShadow
  scene/view_component.rs:45-52, 103-159 + view_component/interpolation.rs        StatefulViewComponent (start / end / transition by component id)
  scene/rescaler_component.rs:41-49, 92-160 + rescaler_component/interpolation.rs StatefulRescalerComponent
  scene/tiles_component.rs:56-65, 119-196 + tiles_component/interpolation.rs:17-97, layout.rs:141-160   StatefulTilesComponent (tiles by id,
                                                  new tiles hidden while an old tile still occupies their place, resize_tiles)
  scene/scene_state.rs:59-66, 83-103, 233-264     last_pts = the last rendered pts; recalculate_layout of the previous tree at last_pts before
                                                  an update (that is where a Tiles component's `last_layout`, the next transition's start, comes from)

Pinned by the reference's own easing vectors (scene/transition/cubic_bezier.rs:136-148: tests/test_oracle_transition.py) and by hand-computed
mid-transition layouts of the reference's render-test scenes (same file).
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import scene as S

F = np.float32
NS = 1_000_000_000


# ----------------------------------------------------------------------------- easing (scene/transition/{bounce,cubic_bezier}.rs)
ALLOWED_FLOATING_ERROR = 1e-7


def _close(a, b):
    return abs(a - b) < ALLOWED_FLOATING_ERROR


def _clamp_root(x):
    """F64Ext::clamp_valid_root_in_unit_range (cubic_bezier.rs:122-142)."""
    if x != x:
        return x
    if x < 0.0:
        return 0.0 if x >= -ALLOWED_FLOATING_ERROR else math.nan
    if x > 1.0:
        return 1.0 if x <= 1.0 + ALLOWED_FLOATING_ERROR else math.nan
    return x


def _cbrt(x):
    return math.copysign(abs(x) ** (1.0 / 3.0), x) if x == x else x


def _find_first_cubic_root(p0, p1, p2, p3):
    """cubic_bezier.rs:33-112."""
    a = 3.0 * (p0 - 2.0 * p1 + p2)
    b = 3.0 * (p1 - p0)
    c = p0
    d = -p0 + 3.0 * (p1 - p2) + p3
    if _close(d, 0.0):
        if _close(a, 0.0):
            if _close(b, 0.0):
                return math.nan
            return _clamp_root(-c / b)
        disc = b * b - 4.0 * a * c
        q = math.sqrt(disc) if disc >= 0.0 else math.nan
        a2 = 2.0 * a
        root = _clamp_root((q - b) / a2)
        if root == root:
            return root
        return _clamp_root((-b - q) / a2)
    a, b, c = a / d, b / d, c / d
    o3 = (3.0 * b - a * a) / 9.0
    q2 = (2.0 * a * a * a - 9.0 * a * b + 27.0 * c) / 54.0
    a3 = a / 3.0
    discriminant = q2 * q2 + o3 * o3 * o3
    if discriminant < 0.0:
        mp33 = -(o3 * o3 * o3)
        r = math.sqrt(mp33)
        cos_phi = min(max(-q2 / r, -1.0), 1.0)
        phi = math.acos(cos_phi)
        t1 = 2.0 * _cbrt(r)
        for k in (0.0, 2.0 * math.pi, 4.0 * math.pi):
            root = _clamp_root(t1 * math.cos((phi + k) / 3.0) - a3)
            if root == root or k == 4.0 * math.pi:
                return root
    if discriminant == 0.0:
        u1 = -_cbrt(q2)
        root = _clamp_root(2.0 * u1 - a3)
        if root == root:
            return root
        return _clamp_root(-u1 - a3)
    sd = math.sqrt(discriminant)
    u1 = _cbrt(-q2 + sd)
    v1 = _cbrt(q2 + sd)
    return _clamp_root(u1 - v1 - a3)


def cubic_bezier_easing(progress, x1, y1, x2, y2):
    """cubic_bezier.rs:5-31."""
    if _close(progress, 0.0):
        return 0.0
    if _close(progress, 1.0):
        return 1.0
    t = _find_first_cubic_root(-progress, x1 - progress, x2 - progress, 1.0 - progress)
    if t != t:
        return 1.0
    a = 1.0 / 3.0 + (y1 - y2)
    b = y2 - 2.0 * y1
    c = y1
    return min(max(3.0 * ((a * t + b) * t + c) * t, 0.0), 1.0)


def bounce_easing(t):
    """bounce.rs:1-14."""
    n1, d1 = 7.5625, 2.75
    if t < 1.0 / d1:
        return n1 * t * t
    if t < 2.0 / d1:
        return n1 * (t - 1.5 / d1) * (t - 1.5 / d1) + 0.75
    if t < 2.5 / d1:
        return n1 * (t - 2.25 / d1) * (t - 2.25 / d1) + 0.9375
    return n1 * (t - 2.625 / d1) * (t - 2.625 / d1) + 0.984375


@dataclass(frozen=True)
class Easing:
    """InterpolationKind (scene/transition.rs:108-119)."""
    kind: str = "linear"  # linear | bounce | cubic_bezier
    points: Tuple[float, float, float, float] = (0.0, 0.0, 1.0, 1.0)

    def state(self, t: float) -> float:
        if self.kind == "linear":
            return t
        if self.kind == "bounce":
            return bounce_easing(t)
        return cubic_bezier_easing(t, *self.points)


@dataclass(frozen=True)
class TransitionOptions:
    """scene::Transition (smelter-api/src/video/transition.rs:33-69): duration in whole nanoseconds."""
    duration_ns: int
    easing: Easing = Easing()
    should_interrupt: bool = False


def _secs(ns: int) -> float:
    """Duration::as_secs_f64: whole seconds + nanoseconds / 1e9."""
    return float(ns // NS) + float(ns % NS) / 1e9


@dataclass
class TransitionState:
    """scene/transition.rs:19-106."""
    offset_progress: float
    offset_state: float
    start_ns: int
    duration_ns: int
    easing: Easing

    @staticmethod
    def new(current: Optional[TransitionOptions], previous: Optional["TransitionState"], props_changed: bool, interrupt: bool,
            last_ns: int) -> Optional["TransitionState"]:
        if previous is not None and not previous.is_finished(last_ns):
            if props_changed and interrupt:
                return TransitionState(0.0, 0.0, last_ns, current.duration_ns, current.easing) if current is not None else None
            remaining = max(previous.start_ns + previous.duration_ns - last_ns, 0)
            progress_offset = 1.0 - (_secs(remaining) / _secs(previous.duration_ns))
            state_offset = previous.easing.state(progress_offset)
            return TransitionState(progress_offset, state_offset, last_ns, remaining, current.easing if current is not None else previous.easing)
        if props_changed and current is not None:
            return TransitionState(0.0, 0.0, last_ns, current.duration_ns, current.easing)
        return None

    def state(self, pts_ns: int) -> float:
        """TransitionState::state (transition.rs:88-101), f64 throughout; a zero duration divides like IEEE does (inf / NaN, no exception)."""
        with np.errstate(all="ignore"):
            progress = (np.float64(_secs(pts_ns)) - np.float64(_secs(self.start_ns))) / np.float64(_secs(self.duration_ns))
            progress = np.float64(self.offset_progress) + progress * (np.float64(1.0) - np.float64(self.offset_progress))
            progress = float(progress)
            progress = 0.0 if progress < 0.0 else 1.0 if progress > 1.0 else progress  # f64::clamp (a NaN stays a NaN)
            s = self.easing.state(progress) if progress == progress else math.nan
            return float((np.float64(s) - np.float64(self.offset_state)) / (np.float64(1.0) - np.float64(self.offset_state)))

    def is_finished(self, pts_ns: int) -> bool:
        return self.start_ns + self.duration_ns <= pts_ns


# ----------------------------------------------------------------------------- ContinuousValue
def _lerp_f32(a, b, s):
    """f32: interpolate_f64(start as f64, end as f64, state) as f32 (types/interpolation.rs:10-30)."""
    return float(F(float(a) + (float(b) - float(a)) * s))


def _lerp_opt(a, b, s):
    return _lerp_f32(a, b, s) if a is not None and b is not None else b


def _lerp_offsets(s_a, s_b, e_a, e_b, s):
    """Vertical / HorizontalPosition: same variant on both sides -> interpolated, otherwise the end's (types/interpolation.rs:57-87).
    (a, b) = (top, bottom) or (left, right): exactly one of each pair is set."""
    if e_a is not None:
        return (_lerp_f32(s_a, e_a, s) if s_a is not None else e_a), None
    if e_b is not None:
        return None, (_lerp_f32(s_b, e_b, s) if s_b is not None else e_b)
    return None, None


def _lerp_position(start, end, s):
    """Position (components/interpolation.rs:8-29) on oracle.scene's (width, height, absolute) triple: both static or both absolute ->
    interpolated, a change of kind -> the end's."""
    out = {}
    if start.absolute is None and end.absolute is None:
        out["width"], out["height"], out["absolute"] = _lerp_opt(start.width, end.width, s), _lerp_opt(start.height, end.height, s), None
    elif start.absolute is not None and end.absolute is not None:
        a, b = start.absolute, end.absolute
        top, bottom = _lerp_offsets(a.top, a.bottom, b.top, b.bottom, s)
        left, right = _lerp_offsets(a.left, a.right, b.left, b.right, s)
        out["width"], out["height"] = end.width, end.height
        out["absolute"] = S.AbsolutePosition(width=_lerp_opt(a.width, b.width, s), height=_lerp_opt(a.height, b.height, s), top=top, bottom=bottom,
                                             left=left, right=right, rotation_degrees=_lerp_f32(a.rotation_degrees, b.rotation_degrees, s))
    else:
        out["width"], out["height"], out["absolute"] = end.width, end.height, copy.deepcopy(end.absolute)
    return out


def _lerp_shadows(start, end, s):
    out = [S.BoxShadow(_lerp_f32(a.offset_x, b.offset_x, s), _lerp_f32(a.offset_y, b.offset_y, s), _lerp_f32(a.blur_radius, b.blur_radius, s), b.color)
           for a, b in zip(start, end)]
    return out + [copy.deepcopy(b) for b in end[min(len(start), len(end)):]]


def _lerp_view(start: S.View, end: S.View, s) -> S.View:
    """view_component/interpolation.rs:5-29 (children are not part of the parameters: the caller attaches them)."""
    v = copy.copy(end)
    for k, x in _lerp_position(start, end, s).items():
        setattr(v, k, x)
    v.border_radius = _lerp_f32(start.border_radius, end.border_radius, s)
    v.border_width = _lerp_f32(start.border_width, end.border_width, s)
    v.box_shadow = _lerp_shadows(start.box_shadow, end.box_shadow, s)
    v.padding = S.Padding(*[_lerp_f32(getattr(start.padding, k), getattr(end.padding, k), s) for k in ("top", "right", "bottom", "left")])
    return v


def _lerp_rescaler(start: S.Rescaler, end: S.Rescaler, s) -> S.Rescaler:
    """rescaler_component/interpolation.rs:5-28."""
    r = copy.copy(end)
    for k, x in _lerp_position(start, end, s).items():
        setattr(r, k, x)
    r.border_radius = _lerp_f32(start.border_radius, end.border_radius, s)
    r.border_width = _lerp_f32(start.border_width, end.border_width, s)
    r.box_shadow = _lerp_shadows(start.box_shadow, end.box_shadow, s)
    return r


@dataclass
class Tile:
    id: Tuple  # ("id", component id) | ("index", n)
    top: float
    left: float
    width: float
    height: float


def _positions_equal(a: Tile, b: Tile) -> bool:
    tol = F(0.001)
    return all(abs(F(getattr(a, k)) - F(getattr(b, k))) <= tol for k in ("top", "left", "width", "height"))


def _lerp_tiles(start: List[Optional[Tile]], end: List[Optional[Tile]], s) -> List[Optional[Tile]]:
    """tiles_component/interpolation.rs:17-97."""
    if s >= 1.0:
        return list(end)
    start_by_id = {t.id: t for t in start if t is not None}
    end_ids = {t.id for t in end if t is not None}
    out = []
    for t in end:
        if t is None:
            out.append(None)
            continue
        old = start_by_id.get(t.id)
        if old is not None:
            out.append(Tile(t.id, _lerp_f32(old.top, t.top, s), _lerp_f32(old.left, t.left, s), _lerp_f32(old.width, t.width, s),
                            _lerp_f32(old.height, t.height, s)))
            continue
        # a new tile: hidden until the end of the transition if a tile that still exists somewhere held its place before
        same_place = next((st for st in start if st is not None and _positions_equal(st, t)), None)
        out.append(None if same_place is None or same_place.id in end_ids else t)
    return out


def _resize_tiles(tiles, from_size, to_size):
    """tiles_component/layout.rs:141-160."""
    scale = S._fmin(F(to_size[0]) / F(from_size[0]), F(to_size[1]) / F(from_size[1]))
    return [None if t is None else Tile(t.id, float(F(t.top) * scale), float(F(t.left) * scale), float(F(t.width) * scale), float(F(t.height) * scale))
            for t in tiles]


# ----------------------------------------------------------------------------- the stateful tree
@dataclass
class _Node:
    kind: str                      # view | rescaler | tiles | leaf
    end: object                    # oracle.scene component (its parameters; children are kept in `children`)
    start: Optional[object] = None
    transition: Optional[TransitionState] = None
    children: List["_Node"] = field(default_factory=list)
    last_layout: Optional[Tuple[List[Optional[Tile]], Tuple[float, float]]] = None   # tiles: (tiles, size) of the last layout call
    tiles_start: Optional[Tuple[List[Optional[Tile]], Tuple[float, float]]] = None

    @property
    def component_id(self):
        return getattr(self.end, "id", None)


def _params_equal(a, b) -> bool:
    """`state.end != end` on the parameter structs (children excluded)."""
    skip = {"children", "child", "transition", "tiles_override"}
    return all(getattr(a, k) == getattr(b, k) for k in a.__dataclass_fields__ if k not in skip)


class SceneState:
    """One output's scene through updates and renders: SceneState::update_scene / register_render_event for a tree of View / Rescaler /
    Tiles / InputStream components (what the reference's render-test scenes are made of)."""

    def __init__(self):
        self.root: Optional[_Node] = None
        self.last_ns = 0
        self.resolution = None

    # -- update_scene (scene_state.rs:73-124)
    def update(self, root_component, width, height):
        prev = {}
        if self.root is not None:
            # recalculate_layout of the previous tree at the last rendered pts: refreshes every Tiles component's last_layout
            old = self._materialise(self.root, self.last_ns, record=True)
            n_inputs = len(S.node_children(old))
            S.update_state(old, (list(self._last_resolutions) + [None] * n_inputs)[:n_inputs])
            S.layout(old, F(self.resolution[0]), F(self.resolution[1]))  # (for its side effect: tiles_for_layout records each Tiles' tile list)
            self._gather(self.root, prev)
        self.root = self._build(root_component, prev)
        self.resolution = (width, height)

    def _gather(self, n: _Node, out: Dict):
        if n.component_id is not None:
            out[n.component_id] = n
        for c in n.children:
            self._gather(c, out)

    def _build(self, c, prev) -> _Node:
        kids = [self._build(k, prev) for k in S._children(c)] if S._is_layout(c) else []
        if isinstance(c, (S.View, S.Rescaler)):
            kind = "view" if isinstance(c, S.View) else "rescaler"
            p = prev.get(c.id) if c.id is not None else None
            p = p if p is not None and p.kind == kind else None
            start = self._snapshot(p, self.last_ns) if p is not None else None
            changed = (not _params_equal(p.end, c)) if p is not None else False
            opts = c.transition
            tr = TransitionState.new(opts, p.transition if p is not None else None, changed, bool(opts and opts.should_interrupt), self.last_ns)
            return _Node(kind, c, start, tr, kids)
        if isinstance(c, S.Tiles):
            p = prev.get(c.id) if c.id is not None else None
            p = p if p is not None and p.kind == "tiles" else None
            changed = False
            if p is not None:
                changed = (not _params_equal(p.end, c)) or len(p.children) != len(kids) or any(a.component_id != b.component_id for a, b in zip(p.children, kids))
            opts = c.transition
            tr = TransitionState.new(opts, p.transition if p is not None else None, changed, bool(opts and opts.should_interrupt), self.last_ns)
            last = copy.deepcopy(p.last_layout) if p is not None else None
            return _Node("tiles", c, None, tr, kids, last_layout=last, tiles_start=copy.deepcopy(last))
        return _Node("leaf", c)

    # -- a component's parameters at a pts (StatefulViewComponent::view, StatefulRescalerComponent::transition_snapshot)
    def _snapshot(self, n: _Node, pts_ns):
        if n.transition is None or n.start is None:
            return n.end
        s = n.transition.state(pts_ns)
        return _lerp_view(n.start, n.end, s) if n.kind == "view" else _lerp_rescaler(n.start, n.end, s)

    def _tiles_end(self, n: _Node, size):
        pos = S.tiles_positions(n.end, len(n.children), size[0], size[1]) if n.children else []
        out, index = [], 0
        for (top, left, tw, th), child in zip(pos, n.children):
            cid = child.component_id
            if cid is not None:
                tid = ("id", cid)
            else:
                tid = ("index", index)
                index += 1
            out.append(Tile(tid, float(top), float(left), float(tw), float(th)))
        return out

    def _tiles_at(self, n: _Node, size, pts_ns):
        """StatefulTilesComponent::tiles (tiles_component.rs:56-65)."""
        end = self._tiles_end(n, size)
        if n.tiles_start is None or n.transition is None:
            return end
        start = _resize_tiles(n.tiles_start[0], n.tiles_start[1], size)
        return _lerp_tiles(start, end, n.transition.state(pts_ns))

    # -- the component tree as plain oracle.scene objects at a pts.  Tiles carry their tile list (tiles_override), for which the size they
    #    are laid out into is needed: the tree is materialised top-down along oracle.scene's own layout pass.
    def _materialise(self, n: _Node, pts_ns, record=False):
        if n.kind == "leaf":
            return copy.copy(n.end)
        if n.kind == "tiles":
            c = copy.copy(n.end)
            c.children = [self._materialise(k, pts_ns, record) for k in n.children]
            c.tiles_state = (self, n, pts_ns, record)   # resolved by oracle.scene._tiles_layout when the size is known
            return c
        c = copy.copy(self._snapshot(n, pts_ns))
        kids = [self._materialise(k, pts_ns, record) for k in n.children]
        if n.kind == "view":
            c.children = kids
        else:
            c.child = kids[0]
        return c

    _last_resolutions: List = []

    def tiles_for_layout(self, n: _Node, w, h, pts_ns, record):
        tiles = self._tiles_at(n, (float(w), float(h)), pts_ns)
        if record:
            n.last_layout = (copy.deepcopy(tiles), (float(w), float(h)))
        return [None if t is None else (F(t.top), F(t.left), F(t.width), F(t.height)) for t in tiles]

    # -- a frame: LayoutNode::render up to the flattened list, then register_render_event
    def layouts(self, pts_ns, input_resolutions, srgb=True):
        W, H = self.resolution
        self._last_resolutions = list(input_resolutions)
        root = self._materialise(self.root, pts_ns)
        out = S.scene_layouts(root, W, H, input_resolutions, srgb=srgb)
        self.last_ns = pts_ns
        return out
