"""Wave B of the hot path on the lane emulator: k_classify_tiles + k_compose_output (smelter_amd/csrc/smr_fused_compose.h) compiled for the
CPU by tests/emu/emu_compose.cpp and launched as smr_render_layouts launches them (the host's own packing of the layout list, the tile
classes, the band list + every tile), against the oracle's apply_layouts + rgba_to_yuv / rgba_to_nv12: EVERY BYTE EQUAL — the compositor
adds no rounding of its own to the tile's bytes, whatever class a tile falls into.  Test infrastructure only: the
product has no CPU path; tests/test_gpu_fused.py and tests/test_gpu_reference_scenes.py hold the kernel itself to the oracle on the device."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
ROOT = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
P8 = C.POINTER(C.c_uint8)
CLASSES = ["clear", "colour", "texture", "full", "skip", "sampled", "select"]


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ to build the emulator with")
    from tests import emu_build
    lib = emu_build.build("smr_emu_compose", "emu_compose.cpp", ("smr_fused_compose.h", "smr_layout_dev.h", "smr_convert_dev.h", "smr_tables.h"))
    h = C.CDLL(lib)
    h.emu_compose.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(P8), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P8, P8, P8, C.POINTER(C.c_int)]
    orc.build()
    if os.environ.get("SMR_EMU_GUARD"):  # the inner run of test_compositor_never_leaves_its_surfaces
        h.emu_set_guard(int(os.environ["SMR_EMU_GUARD"]), 1)
    return h


def _p(a):
    return a.ctypes.data_as(P8)


def compose(emu, layouts, sources, kinds, W, H, out="planar", srgb=True, banded=-1, slices=4, select=1):
    """-> (planes, classes): the emulated launch's output and how many tiles fell into each class."""
    arr = orc.pack_layouts(layouts)
    n = len(sources)
    keep = [None if s is None else np.ascontiguousarray(s, np.uint8) for s in sources]
    px = (P8 * max(n, 1))(*[None if s is None else _p(s) for s in keep])
    ws = (C.c_int * max(n, 1))(*[1 if s is None else s.shape[1] for s in keep])
    hs = (C.c_int * max(n, 1))(*[1 if s is None else s.shape[0] for s in keep])
    ks = (C.c_int * max(n, 1))(*kinds)
    info = (C.c_int * 9)()
    nv = {"planar": 0, "nv12": 1, "rgba": 2}[out]
    if nv == 2:
        o0, o1, o2 = np.zeros((H, W, 4), np.uint8), np.zeros(1, np.uint8), np.zeros(1, np.uint8)
    elif nv == 1:
        o0, o1, o2 = np.zeros((H, W), np.uint8), np.zeros((H // 2, W // 2, 2), np.uint8), np.zeros(1, np.uint8)
    else:
        o0, o1, o2 = np.zeros((H, W), np.uint8), np.zeros((H // 2, W // 2), np.uint8), np.zeros((H // 2, W // 2), np.uint8)
    rc = emu.emu_compose(C.addressof(arr), len(layouts), n, px, ws, hs, ks, W, H, nv, int(srgb), banded, slices, select, _p(o0), _p(o1), _p(o2), info)
    assert rc == 0, rc
    return (o0, o1, o2)[:{0: 3, 1: 2, 2: 1}[nv]], dict(zip(CLASSES + ["listed", "workgroups"], list(info)))


def oracle_output(layouts, sources, W, H, out="planar", srgb=True):
    rgba = orc.apply_layouts(W, H, layouts, sources, srgb=srgb)
    if out == "rgba":
        return (rgba,)
    if out == "nv12":
        return orc.rgba_to_nv12(rgba)
    return orc.rgba_to_planar_yuv(rgba, orc.YUV420)


def assert_same(got, want, what):
    assert len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        w = np.asarray(w).reshape(g.shape)
        assert np.array_equal(g, w), (what, "plane", k, int((g != w).sum()), np.argwhere(g != w)[:5].tolist())


def _video(w, h, seed, alpha=False):
    """An opaque 'resampled tile' (noise on a gradient: every texel differs from its neighbours), or with `alpha` a premultiplied one."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    t = np.stack([(xx * 3 + seed * 40) % 256, (yy * 5 + 90) % 256, (xx + yy * 2) % 256, np.full_like(xx, 255)], -1).astype(np.int64)
    t[..., :3] = np.clip(t[..., :3] + rng.integers(-20, 21, (h, w, 3)), 0, 255)
    if alpha:
        a = rng.integers(0, 256, (h, w))
        t = np.concatenate([t[..., :3] * a[..., None] // 255, a[..., None]], -1)
    return t.astype(np.uint8)


def _tex(i, left, top, w, h, tw=None, th=None, **kw):
    tw, th = tw or int(w), th or int(h)
    return orc.Layout(top=top, left=left, width=w, height=h, type=0, source_index=i, crop=(0.0, 0.0, float(tw), float(th)), **kw)


def _grid_scene(W, H, cols, rows, gap=0, srgb=True, labels=True):
    """BASELINE configs[2] in small: a background colour, a grid of opaque tiles blitted 1:1 (texture / select / colour / clear tiles), and
    translucent rounded labels with a border over the lower left corners of every other one (composited tiles)."""
    tw, th = (W - gap * (cols + 1)) // cols, (H - gap * (rows + 1)) // rows
    layouts = [orc.Layout(top=0.0, left=0.0, width=float(W), height=float(H), type=1, color=orc.color_to_shader((16, 24, 48, 255), srgb))]
    sources, kinds = [], []
    for r in range(rows):
        for c in range(cols):
            if r == rows - 1 and c == cols - 1:
                continue  # the last cell stays background
            i = len(sources)
            sources.append(_video(tw, th, i))
            kinds.append(2)
            left, top = gap + c * (tw + gap), gap + r * (th + gap)
            layouts.append(_tex(i, float(left), float(top), float(tw), float(th)))
            if labels and i % 2 == 0:
                layouts.append(orc.Layout(top=float(top + th - 22), left=float(left + 6), width=70.0, height=16.0, type=1, border_radius=(5.0,) * 4,
                                          color=orc.color_to_shader((0, 0, 0, 150), srgb), border_width=1.0, border_color=orc.color_to_shader((255, 255, 255, 200), srgb)))
    return layouts, sources, kinds


@pytest.mark.parametrize("out", ["planar", "nv12", "rgba"])
@pytest.mark.parametrize("W,H,cols,rows,gap", [(384, 96, 3, 3, 0), (384, 96, 2, 2, 4), (262, 70, 2, 1, 2), (640, 64, 4, 2, 0)])
def test_a_grid_of_tiles_with_labels_is_the_oracle_byte_for_byte(emu, W, H, cols, rows, gap, out):
    layouts, sources, kinds = _grid_scene(W, H, cols, rows, gap)
    got, classes = compose(emu, layouts, sources, kinds, W, H, out)
    assert_same(got, oracle_output(layouts, sources, W, H, out), (W, H, cols, rows, gap, out, classes))
    # (the list holds a tile's BANDS: four under a layer that blends over an area — the labels —, two along opaque layers' edges)
    assert classes["full"] > 0 and 2 * classes["full"] <= classes["listed"] <= 4 * classes["full"]
    if (W, gap) == (384, 0):
        assert classes["texture"] > 0 and classes["colour"] > 0   # whole tiles inside a video tile / inside the empty cell


@pytest.mark.parametrize("banded,slices", [(-1, 8), (0, 4), (1, 4), (1000, 8)])
def test_band_list_predictions_that_miss_and_overshoot(emu, banded, slices):
    """The host sizes the band part of the grid from the list's length when it has come back, from its own prediction otherwise: a
    prediction of nothing (every composited tile done by the workgroup that owns it, band after band), of one tile, of more than exist."""
    W, H = 384, 96
    layouts, sources, kinds = _grid_scene(W, H, 3, 3, 0)
    got, classes = compose(emu, layouts, sources, kinds, W, H, "planar", banded=banded, slices=slices)
    assert_same(got, oracle_output(layouts, sources, W, H), (banded, slices, classes))


@pytest.mark.parametrize("out", ["planar", "rgba"])
@pytest.mark.parametrize("select", [1, 0])
def test_seams_between_tiles_copy_from_the_topmost_layer(emu, out, select):
    """A grid without gaps or labels: the tiles that hold a seam are TC_SELECT (a per-pixel copy from the topmost layer whose box holds the
    pixel), with compose_select off they are composited — the same bytes either way."""
    W, H = 400, 90   # 3 x 2 cells of 133 x 45: seams inside compositor tiles, an uncovered margin on the right
    layouts, sources, kinds = _grid_scene(W, H, 3, 2, 0, labels=False)
    got, classes = compose(emu, layouts, sources, kinds, W, H, out, select=select)
    assert_same(got, oracle_output(layouts, sources, W, H, out), (out, select, classes))
    assert (classes["select"] > 0) == bool(select), classes


@pytest.mark.parametrize("srgb", [True, False])
@pytest.mark.parametrize("out", ["planar", "nv12"])
def test_tiles_in_transition_are_sampled_in_linear_light(emu, out, srgb):
    """A grid in mid-transition: opaque tiles at fractional positions and scales that are not 1:1 — TC_SAMPLED tiles inside them (one
    bilinear layer, column / row halves of the sample positions computed once per block), composited tiles along their edges."""
    W, H = 512, 128
    sources = [_video(240, 60, 1), _video(250, 64, 2), _video(120, 30, 3)]
    kinds = [2, 2, 2]
    layouts = [orc.Layout(top=0.0, left=0.0, width=float(W), height=float(H), type=1, color=orc.color_to_shader((30, 30, 30, 255), srgb)),
               _tex(0, 3.3, 2.7, 390.6, 60.9, 240, 60), _tex(1, 400.25, 1.5, 107.0, 63.25, 250, 64), _tex(2, 100.5, 66.0, 300.0, 59.0, 120, 30)]
    got, classes = compose(emu, layouts, sources, kinds, W, H, out, srgb=srgb)
    assert_same(got, oracle_output(layouts, sources, W, H, out, srgb), (out, srgb, classes))
    assert classes["sampled"] > 0 and classes["full"] > 0, classes


@pytest.mark.parametrize("srgb", [True, False])
def test_tiles_crossing_each_other_start_from_the_topmost_opaque_layer(emu, srgb):
    """A grid whose tiles swap places: opaque tiles at fractional positions OVER one another and over a tile at rest, a translucent label and
    a texture with an alpha channel on top.  In one band a pixel's start value is an opaque colour, a 1:1 texel, a filtered sample of one of
    two moving tiles, or nothing (list compositor, step A); the edges and everything under the label go through the list (steps B, C)."""
    W, H = 384, 70
    sources = [_video(128, 40, 1), _video(120, 36, 2), _video(120, 36, 3), _video(64, 20, 4, alpha=True)]
    kinds = [2, 2, 2, 1]
    layouts = [orc.Layout(top=0.0, left=0.0, width=300.0, height=float(H), type=1, color=orc.color_to_shader((40, 20, 90, 255), srgb)),  # (x >= 300: cleared)
               _tex(0, 16.0, 8.0, 128.0, 40.0),                      # at rest: 1:1 texels
               _tex(1, 90.4, 3.3, 120.0, 36.0),                      # crossing it
               _tex(2, 150.7, 20.6, 120.0, 36.0, border_radius=(6.0,) * 4),  # crossing both, rounded
               orc.Layout(top=30.5, left=60.0, width=200.0, height=18.0, type=1, border_radius=(4.0,) * 4, color=orc.color_to_shader((0, 0, 0, 128), srgb)),
               _tex(3, 250.25, 40.5, 64.0, 20.0)]
    for out in ("planar", "rgba"):
        got, classes = compose(emu, layouts, sources, kinds, W, H, out, srgb=srgb)
        assert_same(got, oracle_output(layouts, sources, W, H, out, srgb), (out, srgb, classes))
    assert classes["full"] > 0, classes


@pytest.mark.parametrize("srgb", [True, False])
@pytest.mark.parametrize("left,top", [(3.3, 2.7), (-40.5, -7.25), (100.0, 20.5)])
def test_a_tile_at_its_own_size_and_a_fractional_position_shares_texels_between_neighbours(emu, left, top, srgb):
    """Sampled tiles whose texel step is exactly one pixel (a tile of a grid whose cells swap places: its own size, a fractional position — here
    also hanging over the frame's corner, where the clamped columns and rows repeat): the 5 x 3 neighbourhood path of the sampled block."""
    W, H = 512, 96
    sources = [_video(440, 90, 5)]
    layouts = [orc.Layout(top=0.0, left=0.0, width=float(W), height=float(H), type=1, color=orc.color_to_shader((30, 30, 30, 255), srgb)),
               _tex(0, left, top, 440.0, 90.0)]
    got, classes = compose(emu, layouts, sources, [2], W, H, "planar", srgb=srgb)
    assert_same(got, oracle_output(layouts, sources, W, H, "planar", srgb), (left, top, srgb, classes))
    assert classes["sampled"] > 0, classes


def _zoo(W, H, rng, n, n_sources, srgb=True):
    layouts = []
    for _ in range(n):
        t = int(rng.integers(0, 3))
        w, h = float(rng.uniform(8, W * 0.7)), float(rng.uniform(6, H * 0.9))
        rmax = min(w, h) / 2
        L = orc.Layout(top=float(rng.uniform(-10, H - 4)), left=float(rng.uniform(-20, W - 8)), width=w, height=h, type=t,
                       rotation_degrees=float(rng.choice([0.0, 0.0, 0.0, 17.0, -33.5, 90.0])),
                       border_radius=tuple(float(rng.uniform(0, rmax)) if rng.random() < 0.5 else 0.0 for _ in range(4)),
                       color=orc.color_to_shader(tuple(int(v) for v in rng.integers(0, 256, 3)) + (int(rng.choice([255, 255, 140, 0])),), srgb),
                       border_color=orc.color_to_shader(tuple(int(v) for v in rng.integers(0, 256, 4)), srgb),
                       border_width=float(rng.choice([0.0, 0.0, 0.5, 1.0, 3.5])), blur_radius=float(rng.choice([0.0, 2.0, 9.0])) if t == 2 else 0.0)
        if t == 0:
            L.source_index = int(rng.integers(0, n_sources + 1))   # (n_sources: out of range = the empty texture)
            L.crop = (float(rng.uniform(0, 6)), float(rng.uniform(0, 9)), float(rng.uniform(20, 90)), float(rng.uniform(10, 50)))
        if rng.random() < 0.4:
            L.masks = [orc.Mask(radius=(float(rng.uniform(0, 12)),) * 4, top=float(rng.uniform(0, H / 2)), left=float(rng.uniform(0, W / 2)),
                                width=float(rng.uniform(W / 4, W)), height=float(rng.uniform(H / 4, H))) for _ in range(int(rng.integers(1, 4)))]
        layouts.append(L)
    return layouts


@pytest.mark.parametrize("seed", range(6))
def test_a_zoo_of_layouts_is_the_oracle_byte_for_byte(emu, seed):
    """Rotations, four radii, borders below and above one pixel, shadows, parent masks, translucent colours, textures with an alpha channel,
    crops, source indices out of range, boxes that leave the frame: random lists on the compositing path, in both rendering modes."""
    rng = np.random.default_rng(100 + seed)
    W, H = int(rng.choice([256, 258, 130])), int(rng.choice([64, 50, 34]))
    srgb = bool(seed % 2 == 0)
    sources = [_video(96, 54, 7, alpha=True), _video(64, 40, 8), None]
    kinds = [1, 2, 0]
    layouts = _zoo(W, H, rng, int(rng.integers(3, 14)), len(sources), srgb)
    out = ["planar", "nv12", "rgba"][seed % 3]
    got, classes = compose(emu, layouts, sources, kinds, W, H, out, srgb=srgb, banded=[-1, 0][seed % 2])
    assert_same(got, oracle_output(layouts, sources, W, H, out, srgb), (seed, W, H, out, srgb, classes))


def test_a_list_longer_than_the_lds_copy_and_an_empty_one(emu):
    """More than 48 layouts / 96 masks: the kernel's build that reads the list in place; no layout at all: transparent black = Y 16, U V 128."""
    rng = np.random.default_rng(5)
    W, H = 256, 48
    sources = [_video(80, 40, 3)]
    layouts = _zoo(W, H, rng, 60, 1)
    got, classes = compose(emu, layouts, sources, [2], W, H, "planar")
    assert_same(got, oracle_output(layouts, sources, W, H), classes)
    got, classes = compose(emu, [], [], [], 130, 18, "planar")
    assert classes["clear"] == sum(classes[c] for c in CLASSES) and (got[0] == 16).all() and (got[1] == 128).all() and (got[2] == 128).all()


def test_compositor_never_leaves_its_surfaces(emu):
    """The memory contract of include/smr.h for wave B: the scenes above once more in child processes with every source texture, output plane,
    layout / mask / class / list array in an allocation of exactly its size — surfaces on the smallest pitch their width allows — ending at
    (mode 1) or starting behind (mode 2) an unmapped page: a 16-byte texel load past a tile's last row, a Y'CbCr store beyond a plane of an
    output whose width is 2 mod 4, a layout record read past the list would kill the child."""
    if os.environ.get("SMR_EMU_GUARD"):
        pytest.skip("this is the inner run")
    children = {mode: subprocess.Popen([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider"],
                                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, SMR_EMU_GUARD=str(mode)), cwd=ROOT)
                for mode in (1, 2)}
    for mode, child in children.items():
        out, err = child.communicate(timeout=1500)
        assert child.returncode == 0, f"guard mode {mode}: rc {child.returncode} (-11 = the kernel left its surfaces)\n{out[-3000:]}\n{err[-2000:]}"
        assert " passed" in out
