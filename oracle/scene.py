"""CPU restatement of smelter-render's scene -> layout maths (host side of the hot path).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  All arithmetic is done in numpy float32 in the
same order as the Rust source so the flattened RenderLayout list matches the reference bit for bit.

Restates:
  scene/view_component/layout.rs:31-285      View
  scene/tiles_component/{tiles.rs:29-166, layout.rs:10-151}  Tiles
  scene/rescaler_component/layout.rs:14-162  Rescaler
  scene/layout.rs:109-262                    layout_content / absolute children / update_state
  transformations/layout/flatten.rs:10-390   NestedLayout::flatten
  scene/types.rs:109-117                     BorderRadius::clip_to_size
Transitions (scene/transition.rs, the stateful components, interpolation by component / tile id) are restated in oracle/transition.py,
which drives this module's layout pass with the component parameters of a given pts.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from .oracle import Layout, Mask, color_to_shader

F = np.float32


def f(x) -> np.float32:
    return np.float32(x)


def _fmin(a, b):
    """Rust's f32::min: a NaN operand is ignored (IEEE minNum) — Python's builtin is order dependent."""
    return a if b != b else b if a != a else (b if b < a else a)


def _fmax(a, b=None):
    """Rust's f32::max; with one argument the maximum of a non-empty list."""
    if b is None:
        acc = a[0]
        for x in a[1:]:
            acc = _fmax(acc, x)
        return acc
    return a if b != b else b if a != a else (b if b > a else a)


# ----------------------------------------------------------------------------- components
@dataclass
class BoxShadow:
    offset_x: float
    offset_y: float
    blur_radius: float
    color: Tuple[int, int, int, int]


@dataclass
class Padding:
    top: float = 0.0
    right: float = 0.0
    bottom: float = 0.0
    left: float = 0.0

    def horizontal(self):
        return f(self.left) + f(self.right)

    def vertical(self):
        return f(self.top) + f(self.bottom)


@dataclass
class AbsolutePosition:
    width: Optional[float] = None
    height: Optional[float] = None
    top: Optional[float] = None      # VerticalPosition::TopOffset
    bottom: Optional[float] = None   # VerticalPosition::BottomOffset
    left: Optional[float] = None
    right: Optional[float] = None
    rotation_degrees: float = 0.0


@dataclass
class InputStream:
    input_index: int          # index into the node's input list (order of appearance)
    size: Tuple[float, float] = (0.0, 0.0)   # filled by update_state from the input resolution
    id: Optional[str] = None  # component id (a Tiles parent keys its tiles by it: oracle/transition.py)


@dataclass
class NodeChild:
    """Any non-layout child with an intrinsic size: Text / Image / Shader / WebView."""
    width: float
    height: float


@dataclass
class View:
    children: list = field(default_factory=list)
    direction: str = "row"
    width: Optional[float] = None
    height: Optional[float] = None
    absolute: Optional[AbsolutePosition] = None
    overflow: str = "hidden"  # visible | hidden | fit
    background_color: Tuple[int, int, int, int] = (0, 0, 0, 0)
    border_radius: float = 0.0
    border_width: float = 0.0
    border_color: Tuple[int, int, int, int] = (0, 0, 0, 0)
    box_shadow: List[BoxShadow] = field(default_factory=list)
    padding: Padding = field(default_factory=Padding)
    id: Optional[str] = None          # component id: a scene update continues the component of the same id (oracle/transition.py)
    transition: Optional[object] = None   # oracle.transition.TransitionOptions


@dataclass
class Rescaler:
    child: object = None
    mode: str = "fit"  # fit | fill
    horizontal_align: str = "center"
    vertical_align: str = "center"
    width: Optional[float] = None
    height: Optional[float] = None
    absolute: Optional[AbsolutePosition] = None
    border_radius: float = 0.0
    border_width: float = 0.0
    border_color: Tuple[int, int, int, int] = (0, 0, 0, 0)
    box_shadow: List[BoxShadow] = field(default_factory=list)
    id: Optional[str] = None
    transition: Optional[object] = None


@dataclass
class Tiles:
    children: list = field(default_factory=list)
    width: Optional[float] = None
    height: Optional[float] = None
    background_color: Tuple[int, int, int, int] = (0, 0, 0, 0)
    tile_aspect_ratio: Tuple[int, int] = (16, 9)
    margin: float = 0.0
    padding: float = 0.0
    horizontal_align: str = "center"
    vertical_align: str = "center"
    absolute: None = None  # Tiles are always statically positioned (tiles_component.rs:77-82)
    id: Optional[str] = None
    transition: Optional[object] = None


def _is_layout(c) -> bool:
    return isinstance(c, (View, Rescaler, Tiles))


def _children(c) -> list:
    if isinstance(c, Rescaler):
        return [c.child]
    return c.children


def node_children(c) -> list:
    """StatefulLayoutComponent::node_children, scene/layout.rs:92-101."""
    out = []
    for ch in _children(c):
        out.extend(node_children(ch) if _is_layout(ch) else [ch])
    return out


def update_state(c, input_resolutions) -> None:
    """scene/layout.rs:103-137: InputStream sizes from the resolutions of this node's inputs."""
    off = 0
    for ch in _children(c):
        if isinstance(ch, InputStream):
            r = input_resolutions[off]
            ch.size = (float(r[0]), float(r[1])) if r is not None else (0.0, 0.0)
            off += 1
        elif _is_layout(ch):
            k = len(node_children(ch))
            update_state(ch, input_resolutions[off:off + k])
            off += k
        else:
            off += 1


def _external(c, v, horizontal: bool):
    """Position::with_border / with_padding (scene/components/position.rs:5-53): the external size of a View includes
    its border and padding, a Rescaler's its border (view_component.rs:61-66, rescaler_component.rs:74-77)."""
    if v is None:
        return None
    v = f(v)
    if isinstance(c, (View, Rescaler)):
        v = v + F(2) * f(c.border_width)
    if isinstance(c, View):
        v = v + (c.padding.horizontal() if horizontal else c.padding.vertical())
    return v


def _width(c):
    if isinstance(c, InputStream):
        return f(c.size[0])
    if isinstance(c, NodeChild):
        return f(c.width)
    if c.absolute is not None:
        return _external(c, c.absolute.width, True)
    return _external(c, c.width, True)


def _height(c):
    if isinstance(c, InputStream):
        return f(c.size[1])
    if isinstance(c, NodeChild):
        return f(c.height)
    if c.absolute is not None:
        return _external(c, c.absolute.height, False)
    return _external(c, c.height, False)


# ----------------------------------------------------------------------------- nested layout
@dataclass
class NMask:
    radius: np.ndarray  # 4 x f32: tl, tr, br, bl
    top: np.float32
    left: np.float32
    width: np.float32
    height: np.float32


@dataclass
class Nested:
    top: np.float32
    left: np.float32
    width: np.float32
    height: np.float32
    rotation_degrees: np.float32 = F(0)
    scale_x: np.float32 = F(1)
    scale_y: np.float32 = F(1)
    crop: Optional[Tuple] = None  # (top, left, width, height)
    mask: Optional[NMask] = None
    content: Tuple = ("none",)    # ("color", rgba) | ("child", index, (w, h)) | ("none",)
    children: list = field(default_factory=list)
    child_nodes_count: int = 0
    border_width: np.float32 = F(0)
    border_color: Tuple[int, int, int, int] = (0, 0, 0, 0)
    border_radius: np.ndarray = field(default_factory=lambda: np.zeros(4, F))
    box_shadow: List[BoxShadow] = field(default_factory=list)


def _radius(r) -> np.ndarray:
    return np.full(4, F(r), F)


def _clip_radius(r: np.ndarray, w, h) -> np.ndarray:
    mx = _fmax(F(0), _fmin(f(w), f(h)) / F(2))
    return np.array([F(0) if x < F(0) else mx if x > mx else x for x in r], F)  # f32::clamp keeps a NaN


def _radius_add(r: np.ndarray, d) -> np.ndarray:
    return np.array([_fmax(x + F(d), F(0)) for x in r], F)


def _layout_content(c, index=0):
    if _is_layout(c):
        return ("none",)
    if isinstance(c, InputStream):
        return ("child", index, (f(c.size[0]), f(c.size[1])))
    return ("child", index, (f(c.width), f(c.height)))


def _placeholder(n) -> Nested:
    return Nested(F(0), F(0), F(0), F(0), child_nodes_count=n)


def layout(c, w, h) -> Nested:
    if isinstance(c, View):
        return _view_layout(c, f(w), f(h))
    if isinstance(c, Rescaler):
        return _rescaler_layout(c, f(w), f(h))
    if isinstance(c, Tiles):
        return _tiles_layout(c, f(w), f(h))
    raise TypeError(c)


def _wrap_layout_child(ch, top, left, w, h, rot=F(0)) -> Nested:
    inner = layout(ch, w, h)
    return Nested(top, left, w, h, rotation_degrees=rot, children=[inner], child_nodes_count=inner.child_nodes_count)


def _absolute_child(ch, pos: AbsolutePosition, pw, ph) -> Nested:
    """layout_absolute_position_child, scene/layout.rs:164-239."""
    w = pw if pos.width is None else _external(ch, pos.width, True)
    h = ph if pos.height is None else _external(ch, pos.height, False)
    top = (ph - f(pos.bottom) - h) if pos.bottom is not None else f(pos.top if pos.top is not None else 0.0)
    left = (pw - f(pos.right) - w) if pos.right is not None else f(pos.left if pos.left is not None else 0.0)
    rot = f(pos.rotation_degrees)
    if _is_layout(ch):
        return _wrap_layout_child(ch, top, left, w, h, rot)
    return Nested(top, left, w, h, rotation_degrees=rot, content=_layout_content(ch), child_nodes_count=1)


def _static_children(c: View):
    return [ch for ch in c.children if not (_is_layout(ch) and ch.absolute is not None)]


def _sum_static(c: View):
    acc = F(0)
    for ch in _static_children(c):
        v = _width(ch) if c.direction == "row" else _height(ch)
        acc = acc + (v if v is not None else F(0))
    return acc


def _view_layout(c: View, w, h) -> Nested:
    bw = f(c.border_width)
    cw = _fmax(w - F(2) * bw, F(0))
    chh = _fmax(h - F(2) * bw, F(0))
    radius = _clip_radius(_radius(c.border_radius), w, h)
    # static_child_size (view_component/layout.rs:205-231)
    max_size = (cw - c.padding.horizontal()) if c.direction == "row" else (chh - c.padding.vertical())
    unknown = sum(1 for ch in _static_children(c) if (_width(ch) if c.direction == "row" else _height(ch)) is None)
    static_child_size = F(0) if unknown == 0 else _fmax(F(0), (max_size - _sum_static(c)) / F(unknown))
    mask = None
    scale = F(1)
    if c.overflow in ("hidden", "fit"):
        mask = NMask(_radius_add(radius, -bw), bw, bw, cw, chh)
    if c.overflow == "fit":
        sum_size = _fmax(_sum_static(c), F(0.000000001))
        mx, alt = (cw, chh) if c.direction == "row" else (chh, cw)
        alts = [((_height(ch) if c.direction == "row" else _width(ch)) or F(0)) for ch in _static_children(c)]
        max_alt = _fmax(_fmax(alts) if alts else F(0), F(0.000000001))
        scale = _fmin(F(1), _fmin(mx / sum_size, alt / max_alt))
    static_offset = bw / scale
    parent_bw = bw / scale
    kids = []
    for ch in c.children:
        if _is_layout(ch) and ch.absolute is not None:
            kids.append(_absolute_child(ch, ch.absolute, w, h))
            continue
        cwid, chei = _width(ch), _height(ch)
        if c.direction == "row":
            ww = cwid if cwid is not None else static_child_size
            hh = chei if chei is not None else (chh - c.padding.vertical())
            top = parent_bw + f(c.padding.top)
            left = static_offset + f(c.padding.left)
            static_offset = static_offset + ww
        else:
            hh = chei if chei is not None else static_child_size
            ww = cwid if cwid is not None else (cw - c.padding.horizontal())
            top = static_offset + f(c.padding.top)
            left = parent_bw + f(c.padding.left)
            static_offset = static_offset + hh
        if _is_layout(ch):
            kids.append(_wrap_layout_child(ch, top, left, ww, hh))
        else:
            kids.append(Nested(top, left, ww, hh, content=_layout_content(ch), child_nodes_count=1))
    return Nested(F(0), F(0), w, h, scale_x=scale, scale_y=scale, mask=mask, content=("color", tuple(c.background_color)),
                  children=kids, child_nodes_count=sum(k.child_nodes_count for k in kids), border_width=bw,
                  border_color=tuple(c.border_color), border_radius=radius, box_shadow=list(c.box_shadow))


def _rescaler_layout(c: Rescaler, w, h) -> Nested:
    bw = f(c.border_width)
    cw = _fmax(w - F(2) * bw, F(0))
    chh = _fmax(h - F(2) * bw, F(0))
    child = c.child
    kw, kh = _width(child), _height(child)
    radius = _clip_radius(_radius(c.border_radius), w, h)
    if kw is None and kh is None:
        scale = F(1)
    elif kw is None:
        scale = chh / kh
    elif kh is None:
        scale = cw / kw
    else:
        scale = _fmin(cw / kw, chh / kh) if c.mode == "fit" else _fmax(cw / kw, chh / kh)
    if _is_layout(child):
        inner = layout(child, kw if kw is not None else cw / scale, kh if kh is not None else chh / scale)
        content, children, count = ("none",), [inner], inner.child_nodes_count
    else:
        content, children, count = _layout_content(child), [], 1
    if c.vertical_align == "top" or kh is None:
        top = F(0)
    elif c.vertical_align == "bottom":
        top = chh - kh * scale
    else:
        top = (chh - kh * scale) / F(2)
    if c.horizontal_align == "left" or kw is None:
        left = F(0)
    elif c.horizontal_align == "right":
        left = cw - kw * scale
    else:
        left = (cw - kw * scale) / F(2)
    width = kw * scale if kw is not None else cw
    height = kh * scale if kh is not None else chh
    inner = Nested(top + bw, left + bw, width, height, scale_x=scale, scale_y=scale, content=content, children=children,
                   child_nodes_count=count)
    return Nested(F(0), F(0), cw + bw * F(2), chh + bw * F(2), mask=NMask(_radius_add(radius, -bw), bw, bw, cw, chh),
                  children=[inner], child_nodes_count=count, border_width=bw, border_color=tuple(c.border_color),
                  border_radius=radius, box_shadow=list(c.box_shadow))


def _tile_size(c: Tiles, rows, cols, w, h):
    pad, mar = f(c.padding), f(c.margin)
    x_padding = F(cols) * F(2) * pad
    y_padding = F(rows) * F(2) * pad
    x_margin = (F(cols) + F(1)) * mar
    y_margin = (F(rows) + F(1)) * mar
    xs = _fmax(w - x_padding - x_margin, F(0)) / F(cols) / F(c.tile_aspect_ratio[0])
    ys = _fmax(h - y_padding - y_margin, F(0)) / F(rows) / F(c.tile_aspect_ratio[1])
    s = xs if xs < ys else ys
    return F(c.tile_aspect_ratio[0]) * s, F(c.tile_aspect_ratio[1]) * s


def tiles_positions(c: Tiles, count: int, w, h):
    """TilesComponentParams::tiles, scene/tiles_component/tiles.rs:29-166."""
    w, h = f(w), f(h)
    best = (1, -(-count // 1))
    best_w = F(0)
    for rows in range(1, count + 1):
        cols = -(-count // rows)
        tw, _ = _tile_size(c, rows, cols, w, h)
        if tw > best_w:
            best, best_w = (rows, cols), tw
    rows, cols = best
    tw, th = _tile_size(c, rows, cols, w, h)
    pad, mar = f(c.padding), f(c.margin)
    add_y = h - (th + F(2) * pad) * F(rows) - (mar * (F(rows) + F(1)))
    if c.vertical_align == "top":
        add_top, just_y = F(0), F(0)
    elif c.vertical_align == "center":
        add_top, just_y = add_y / F(2), F(0)
    elif c.vertical_align == "bottom":
        add_top, just_y = add_y, F(0)
    else:
        add_top, just_y = F(0), add_y / (F(rows) + F(1))
    out = []
    top = add_top + just_y + pad + mar
    for row in range(rows):
        in_row = cols if row < rows - 1 else count - (rows - 1) * cols
        add_x = w - (tw + F(2) * pad) * F(in_row) - (mar * (F(in_row) + F(1)))
        if c.horizontal_align == "left":
            add_left, just_x = F(0), F(0)
        elif c.horizontal_align == "right":
            add_left, just_x = add_x, F(0)
        elif c.horizontal_align == "justified":
            add_left, just_x = F(0), add_x / F(in_row + 1)
        else:
            add_left, just_x = add_x / F(2), F(0)
        left = add_left + just_x + mar + pad
        for _ in range(in_row):
            out.append((top, left, tw, th))
            left = left + (tw + mar + pad * F(2) + just_x)
        top = top + (th + mar + pad * F(2) + just_y)
    return out


def _tiles_layout(c: Tiles, w, h) -> Nested:
    state = getattr(c, "tiles_state", None)
    if state is not None:
        # a scene with history (oracle/transition.py): the tile list at this pts — interpolated from the previous layout's, entries of
        # tiles that stay hidden during the transition are None (tiles_component.rs:56-65, layout.rs:44-53)
        scene, node, pts_ns, record = state
        tiles = scene.tiles_for_layout(node, w, h, pts_ns, record)
    else:
        tiles = tiles_positions(c, len(c.children), w, h) if c.children else []
    kids = []
    for ch, tile in zip(c.children, tiles):
        if tile is None:
            kids.append(_placeholder(len(node_children(ch)) if _is_layout(ch) else 1))
            continue
        top, left, tw, th = tile
        if _is_layout(ch):
            kids.append(_wrap_layout_child(ch, top, left, tw, th))
        else:
            # fit_into_tile (tiles_component/layout.rs:114-135)
            kw, kh = _width(ch), _height(ch)
            if kw is not None and kh is not None:
                s = _fmin(tw / kw, th / kh)
                top, left, tw, th = top + (th - s * kh) / F(2), left + (tw - s * kw) / F(2), s * kw, s * kh
            kids.append(Nested(top, left, tw, th, content=_layout_content(ch), child_nodes_count=1))
    return Nested(F(0), F(0), w, h, content=("color", tuple(c.background_color)), children=kids,
                  child_nodes_count=sum(k.child_nodes_count for k in kids))


# ----------------------------------------------------------------------------- flatten
@dataclass
class RL:  # RenderLayout (layout.rs:58-96)
    top: np.float32
    left: np.float32
    width: np.float32
    height: np.float32
    rotation_degrees: np.float32
    border_radius: np.ndarray
    masks: List[NMask]
    kind: str  # color | child | shadow
    color: Tuple[int, int, int, int] = (0, 0, 0, 0)
    border_color: Tuple[int, int, int, int] = (0, 0, 0, 0)
    border_width: np.float32 = F(0)
    index: int = 0
    crop: Tuple = (F(0), F(0), F(0), F(0))
    blur_radius: np.float32 = F(0)


def _child_parent_masks(n: Nested, masks):
    s = _fmin(n.scale_x, n.scale_y)
    return [NMask(m.radius * (F(1) / s), (m.top - n.top) / n.scale_y, (m.left - n.left) / n.scale_x, m.width / n.scale_x,
                  m.height / n.scale_y) for m in masks]


def _parent_parent_masks(n: Nested, masks):
    s = _fmin(n.scale_x, n.scale_y)
    return [NMask(m.radius * s, (m.top * n.scale_y) + n.top, (m.left * n.scale_x) + n.left, m.width * n.scale_x,
                  m.height * n.scale_y) for m in masks]


def _flatten_child(n: Nested, c: RL) -> RL:
    us = _fmin(n.scale_x, n.scale_y)
    bw = c.border_width
    blur = c.blur_radius
    if n.crop is None:
        top, left = n.top + (c.top * n.scale_y), n.left + (c.left * n.scale_x)
        width, height = c.width * n.scale_x, c.height * n.scale_y
        crop = c.crop
        bw = bw * us  # both Color and ChildNode scale the border when there is no crop
        blur = blur * us
    else:
        ct, cl, cwid, chei = n.crop
        cropped_top = _fmax(c.top - ct, F(0))
        cropped_left = _fmax(c.left - cl, F(0))
        cropped_bottom = _fmin(c.top + c.height - ct, chei)
        cropped_right = _fmin(c.left + c.width - cl, cwid)
        cw_, ch_ = cropped_right - cropped_left, cropped_bottom - cropped_top
        top, left = n.top + (cropped_top * n.scale_y), n.left + (cropped_left * n.scale_x)
        width, height = cw_ * n.scale_x, ch_ * n.scale_y
        crop = c.crop
        if c.kind == "child":
            top_diff = _fmax(ct - c.top, F(0))
            left_diff = _fmax(cl - c.left, F(0))
            hs = c.crop[2] / c.width
            vs = c.crop[3] / c.height
            crop = (c.crop[0] + (top_diff * vs), c.crop[1] + (left_diff * hs), cw_ * hs, ch_ * vs)
            # flatten.rs:262-279: ChildNode keeps border_width unscaled under a crop
        else:
            bw = bw * us
        blur = blur * us
    return RL(top, left, width, height, c.rotation_degrees + n.rotation_degrees, c.border_radius * us,
              _parent_parent_masks(n, c.masks), c.kind, c.color, c.border_color, bw, c.index, crop, blur)


def _inner_flatten(n: Nested, offset: int, parent_masks):
    content = n.content
    if content[0] == "child":
        content = ("child", content[1] + offset, content[2])
        offset += 1
    if content[0] == "color":
        me = RL(n.top, n.left, n.width, n.height, n.rotation_degrees, n.border_radius, list(parent_masks), "color",
                color=content[1], border_color=n.border_color, border_width=n.border_width)
    elif content[0] == "child":
        me = RL(n.top, n.left, n.width, n.height, n.rotation_degrees, n.border_radius, list(parent_masks), "child",
                border_color=n.border_color, border_width=n.border_width, index=content[1],
                crop=(F(0), F(0), content[2][0], content[2][1]))
    else:
        me = RL(n.top, n.left, n.width, n.height, n.rotation_degrees, n.border_radius, list(parent_masks), "color",
                color=(0, 0, 0, 0), border_color=n.border_color, border_width=n.border_width)
    shadows = [RL(n.top + f(s.offset_y), n.left + f(s.offset_x), n.width, n.height, n.rotation_degrees,
                  _radius_add(n.border_radius, f(s.blur_radius) / F(2)), list(parent_masks), "shadow", color=tuple(s.color),
                  blur_radius=f(s.blur_radius)) for s in n.box_shadow]
    masks = list(parent_masks) + ([n.mask] if n.mask is not None else [])
    masks = _child_parent_masks(n, masks)
    child_shadows, child_layouts = [], []
    for ch in n.children:
        cnt = ch.child_nodes_count
        s, l = _inner_flatten(ch, offset, masks)
        offset += cnt
        child_shadows.extend(s)
        child_layouts.extend(l)
    child_shadows = [_flatten_child(n, c) for c in child_shadows]
    child_layouts = [_flatten_child(n, c) for c in child_layouts]
    return shadows, [me] + child_shadows + child_layouts


def _should_render(l: RL, input_resolutions, W, H) -> bool:
    if l.width <= 0 or l.height <= 0 or l.top > F(H) or l.left > F(W):
        return False
    if l.kind == "color":
        if l.color[3] == 0:
            return l.border_color[3] != 0 or l.border_width > 0
        return True
    if l.kind == "child":
        size = input_resolutions[l.index] if l.index < len(input_resolutions) else None
        if size is not None and (l.crop[1] > F(size[0]) or l.crop[0] > F(size[1])):
            return False
        if l.crop[0] + l.crop[3] < 0 or l.crop[1] + l.crop[2] < 0:
            return False
        return True
    return l.color[3] != 0


def _fix_final(l: RL) -> RL:
    if l.kind in ("color", "child") and l.border_width < F(1):
        l.border_width = F(0)
    keep = []
    for m in l.masks:
        mt = _fmax(m.radius[0], m.radius[1])
        mb = _fmax(m.radius[3], m.radius[2])
        ml = _fmax(m.radius[0], m.radius[3])
        mr = _fmax(m.radius[1], m.radius[2])
        skip = (m.top + mt <= l.top and m.left + ml <= l.left and m.left + m.width - mr >= l.left + l.width
                and m.top + m.height - mb >= l.top + l.height)
        if not skip:
            keep.append(m)
    l.masks = keep
    return l


def flatten(n: Nested, input_resolutions, W, H) -> List[RL]:
    shadows, layouts = _inner_flatten(n, 0, [])
    return [_fix_final(l) for l in shadows + layouts if _should_render(l, input_resolutions, W, H)]


def to_render_layouts(rls: List[RL], srgb=True) -> List[Layout]:
    """ParamsBindGroups::update conventions (layout/params.rs:223-303) -> the POD the kernels take."""
    out = []
    for l in rls:
        masks = [Mask([float(x) for x in m.radius], float(m.top), float(m.left), float(m.width), float(m.height)) for m in l.masks]
        base = dict(top=float(l.top), left=float(l.left), width=float(l.width), height=float(l.height),
                    rotation_degrees=float(l.rotation_degrees), border_radius=[float(x) for x in l.border_radius], masks=masks)
        if l.kind == "color":
            out.append(Layout(type=1, color=color_to_shader(l.color, srgb), border_color=color_to_shader(l.border_color, srgb),
                              border_width=float(l.border_width), **base))
        elif l.kind == "child":
            out.append(Layout(type=0, source_index=l.index, border_color=color_to_shader(l.border_color, srgb),
                              border_width=float(l.border_width), crop=[float(x) for x in l.crop], **base))
        else:
            out.append(Layout(type=2, color=color_to_shader(l.color, srgb), blur_radius=float(l.blur_radius), **base))
    return out


def scene_layouts(root, W, H, input_resolutions, srgb=True) -> List[Layout]:
    """LayoutNode::render up to the flattened list (transformations/layout.rs:176-184)."""
    update_state(root, list(input_resolutions))
    nested = layout(root, F(W), F(H))
    return to_render_layouts(flatten(nested, list(input_resolutions), W, H), srgb)
