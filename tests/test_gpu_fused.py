"""The fused hot path (smr_render_layouts), with the three ingest implementations (SMR_OPT_INGEST_IMPL):
  valu   exact f32 kernel: (1) bit-identical to the pass-per-launch path on the same device,
  mfma   the default (SMR_INGEST_AUTO): exact converter into the node texture + matrix-core resampler (k_ingest_wave on RGBA8):
         (1') within 1 LSB of the pass-per-launch path on every content class — white noise included —, >= 99 % of the bytes identical,
  fused  opt-in (SMR_INGEST_MFMA_F16_FUSED): the matrix-core kernel converts planar 4:2:0 / NV12 on the fly (within one code per stage),
and for valu and mfma (2) within 1 LSB of the CPU oracle's restatement of the reference's pass sequence END TO END (>= 99.5 % identical on
the scene cases, >= 99 % on the random-geometry sweeps), (3) size-independent properties at BASELINE.json's full sizes.  The opt-in fused
conversion is held to the same bound on camera-like content and, on white noise, to <= 1 LSB per stage plus an explicit end-to-end bound
against the oracle (<= 4 codes, <= 1e-6 of the bytes beyond 1): its documented contract (include/smr.h)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import scene as S
from tests import convert_model, refpipe, scenes  # (convert_model: the opt-in fused conversion only)

pytestmark = pytest.mark.gpu


def _impl_or_skip(c, hip, impl):
    """The fused-conversion route (ingest implementation 5) exists in laboratory builds of the library only (-DSMR_LAB): a product build
    refuses it, and the tests of that route skip."""
    try:
        c.set_ingest_impl(impl)
    except hip.SmrError:
        c.close()
        pytest.skip("fused conversion: laboratory builds only (-DSMR_LAB)")


@pytest.fixture(scope="module")
def hip():
    from smelter_amd import hip as h
    return h


def _routes():
    """The routes of wave A a build of the library has: the fused conversion exists in laboratory builds only (smr_build_flags)."""
    from smelter_amd import hip as h
    return ["valu", "mfma"] + (["fused"] if h.lab_build() else [])


@pytest.fixture(scope="module", params=_routes())
def ctx(hip, request):
    c = hip.Context(0)
    c.impl = request.param
    _impl_or_skip(c, hip, {"valu": hip.INGEST_VALU_F32, "mfma": hip.INGEST_AUTO, "fused": hip.INGEST_MFMA_F16_FUSED}[request.param])
    yield c
    c.close()


def _oracle_floor(ctx):
    """Share of bytes that must equal the oracle's (all within 1 LSB)."""
    return 0.995


def _node(ctx, y, u, v, w, h, variant=None):
    """The node texture of a planar 4:2:0 input: the oracle's planar_yuv_to_rgba — for the f32 kernel and for the default route (whose
    converter produces exactly those bytes).  Only the opt-in fused conversion is checked per stage: there it is the kernel's own folded FMA
    chain (tests/convert_model.py; within 1 LSB of the oracle's, > 99.98 % identical — tests/test_convert_model.py) and the resampler is held
    to <= 1 LSB on every content class against the oracle's resample of that texture."""
    if ctx.impl != "fused" or (variant is not None and variant not in (orc.YUV420, orc.YUVJ420)) or w % 2 or h % 2:
        return orc.planar_yuv_to_rgba(y, u, v, w, h) if variant is None else orc.planar_yuv_to_rgba(y, u, v, w, h, variant)
    return convert_model.node_codes(y, u, v, full_range=(variant == orc.YUVJ420))


def _within_one_lsb(a, b):
    """max |a - b| <= 1: BASELINE.json's tolerance for resample / colour-convert, on every content class."""
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return d.max() <= 1


def _assert_matches_unfused(ctx, got, ref, what, identical=0.99, noise=False):
    """valu: bit for bit.  mfma (the default): the tolerance BASELINE.json's north star gives the resampler (<= 1 LSB) on every content class,
    and nearly all bytes equal (`identical`: 0.99 on the scene content; 0.98 on full-range white noise).  fused (opt-in), `noise`: white-noise
    planes — the pass-per-launch path resamples the exact converter's node texture, the fused kernel its own (a code apart in a few texels per
    hundred thousand, _node above), and on white noise one such texel can move a dark output pixel by more than one code: bounded explicitly
    (<= 4 codes; at most 4 bytes or 1e-6 of the bytes beyond 1 — the option's documented contract)."""
    for a, b, pl in zip(got, ref, "YUV"):
        if ctx.impl == "valu":
            assert (a == b).all(), f"{what}: fused and unfused paths differ on the same device (plane {pl})"
        else:
            if ctx.impl == "fused" and noise:
                d = np.abs(a.astype(np.int16) - b.astype(np.int16))
                assert d.max() <= 4 and (d > 1).sum() <= max(4, 1e-6 * d.size), f"{what} plane {pl}: fused conversion max {d.max()}, {(d > 1).sum()} bytes beyond 1 LSB"
            else:
                assert _within_one_lsb(a, b), f"{what} plane {pl}: matrix-core path {refpipe.max_diff(a, b)} LSB off the f32 path"
            assert refpipe.exact_fraction(a, b) >= identical, f"{what} plane {pl}: only {refpipe.exact_fraction(a, b):.4f} identical to the f32 path"


@pytest.fixture(scope="module")
def ctx_unfused(hip):
    c = hip.Context(0)
    c.set_fused_kernels(False)  # SMR_OPT_FUSED_KERNELS = 0: one general kernel per pass of the reference
    yield c
    c.close()


def _inputs(ctx, hip, n, w, h, seed=1234):
    planes, frames = [], []
    for i in range(n):
        y, u, v = scenes.test_input(i, w, h, noise_seed=seed + i)
        planes.append((y, u, v))
        frames.append(ctx.frame(hip.FRAME_PLANAR_YUV420, w, h, [y, u, v]))
    return planes, frames


def _label_surfaces(ctx, n):
    atlas, glyphs = scenes.label_glyphs("CAM 3 LIVE", 3)
    bg = orc.color_to_shader((0, 0, 0, 0), True)
    t = ctx.surface(scenes.LABEL_W, scenes.LABEL_H)
    ctx.blit_glyphs(t, bg, glyphs, atlas)
    # the oracle's picture takes the ORACLE's label node (orc.blit_glyphs), never the kernel's own output: the text pixels of the scene
    # tests are compared with an independent computation (the blit itself is held to the oracle in tests/test_gpu_parity.py::test_blit_glyphs)
    host = orc.blit_glyphs(scenes.LABEL_W, scenes.LABEL_H, bg, glyphs, atlas, True)
    return t, host


def _render(ctx, hip, layouts, sources, W, H):
    out = ctx.frame(hip.FRAME_PLANAR_YUV420, W, H)
    ctx.render_layouts(layouts, sources, W, H, out=out)
    return out.download()


SMALL_CASES = [
    ("cfg2_small", lambda: scenes.cfg2_scene(320, 180, 320, 180, 4), 320, 180, 320, 180),
    ("cfg3_small", lambda: scenes.cfg3_scene(480, 270, 960, 540, 8), 480, 270, 960, 540),
    ("cfg3_odd", lambda: scenes.cfg3_scene(482, 274, 1000, 562, 5), 482, 274, 1000, 562),
    ("upscale", lambda: scenes.cfg2_scene(160, 90, 640, 360, 2), 160, 90, 640, 360),
    ("downscale4", lambda: scenes.cfg2_scene(1280, 720, 320, 180, 4), 1280, 720, 320, 180),   # k = 8 -> box pre-reduce
]


@pytest.mark.parametrize("name,mk,iw,ih,W,H", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
def test_fused_equals_unfused_and_oracle(ctx, ctx_unfused, hip, name, mk, iw, ih, W, H):
    layouts, res = mk()
    n_in = sum(1 for r in res if r == (iw, ih))
    planes, _ = _inputs(ctx, hip, n_in, iw, ih)
    label_t, label_host = _label_surfaces(ctx, 1)

    def sources_for(c):
        srcs, k = [], 0
        lt = c.surface_from(label_host)
        for r in res:
            if r == (iw, ih):
                y, u, v = planes[k]
                k += 1
                srcs.append(c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, [y, u, v]))
            else:
                srcs.append(lt)
        return srcs

    got = _render(ctx, hip, layouts, sources_for(ctx), W, H)
    got_unfused = _render(ctx_unfused, hip, layouts, sources_for(ctx_unfused), W, H)
    _assert_matches_unfused(ctx, got, got_unfused, name)
    # oracle
    nodes, k = [], 0
    for r in res:
        if r == (iw, ih):
            y, u, v = planes[k]
            k += 1
            nodes.append(orc.planar_yuv_to_rgba(y, u, v, iw, ih))
        else:
            nodes.append(label_host)
    want, _ = refpipe.render_yuv420(layouts, nodes, W, H)
    for g, w_, pl in zip(got, want, "YUV"):
        d = refpipe.max_diff(g, w_)
        ex = refpipe.exact_fraction(g, w_)
        assert d <= 1, f"{name} plane {pl}: {d} LSB off the oracle"
        assert ex >= _oracle_floor(ctx), f"{name} plane {pl}: only {ex:.4f} identical"


def test_nv12_and_rgba_outputs(ctx, ctx_unfused, hip):
    layouts, res = scenes.cfg2_scene(320, 180, 640, 360, 4)
    planes, frames = _inputs(ctx, hip, 4, 320, 180)
    nodes = refpipe.nodes_from_yuv420(planes)
    rgba_want = refpipe.layout_node_render(layouts, nodes, 640, 360)
    out = ctx.frame(hip.FRAME_NV12, 640, 360)
    ctx.render_layouts(layouts, frames, 640, 360, out=out)
    y, uv = out.download()
    wy, wuv = orc.rgba_to_nv12(rgba_want)
    assert refpipe.max_diff(y, wy) <= 1 and refpipe.max_diff(uv, wuv) <= 1
    t = ctx.surface(640, 360)
    ctx.render_layouts(layouts, frames, 640, 360, out_rgba=t)
    assert refpipe.max_diff(t.download(), rgba_want) <= 1
    # the formats wave B does not write itself (4:2:2, 4:4:4; an RGBA frame likewise): the same compositor kernel onto an RGBA8 scratch target, then the output converter —
    # not the pass-per-launch compositor (k_apply_layouts)
    for fmt, ov in ((hip.FRAME_PLANAR_YUV422, orc.YUV422), (hip.FRAME_PLANAR_YUV444, orc.YUV444)):
        o = ctx.frame(fmt, 640, 360)
        before = ctx.kernel_launches()
        ctx.render_layouts(layouts, frames, 640, 360, out=o)
        ran = {k: v - before[k] for k, v in ctx.kernel_launches().items()}
        assert ran["compose_output"] == 1 and ran["apply_layouts"] == 0, ran
        for g, w_ in zip(o.download(), orc.rgba_to_planar_yuv(rgba_want, ov)):
            assert refpipe.max_diff(g, w_) <= 1


@pytest.mark.parametrize("name,mk,iw,ih,W,H", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
def test_narrow_strip_variant_of_the_ingest_kernel(ctx, ctx_unfused, hip, monkeypatch, name, mk, iw, ih, W, H):
    """The ingest kernel picks 32-column strips when 64-column ones no longer fit two workgroups per CU (large scale factors);
    pinned here on every geometry: still bit-identical to the pass-per-launch path."""
    if ctx.impl != "valu":
        pytest.skip("strip width is a knob of the f32 kernel")
    ctx.set_strip_width(32)
    layouts, res = mk()
    n_in = sum(1 for r in res if r == (iw, ih))
    planes, _ = _inputs(ctx, hip, n_in, iw, ih)
    _, label_host = _label_surfaces(ctx, 1)

    def sources_for(c):
        srcs, k = [], 0
        lt = c.surface_from(label_host)
        for r in res:
            if r == (iw, ih):
                srcs.append(c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(planes[k]))); k += 1
            else:
                srcs.append(lt)
        return srcs

    try:
        got = _render(ctx, hip, layouts, sources_for(ctx), W, H)
    finally:
        ctx.set_strip_width(0)
    ref = _render(ctx_unfused, hip, layouts, sources_for(ctx_unfused), W, H)
    for a, b in zip(got, ref):
        assert (a == b).all(), name
    # ... and the oracle (the narrow strips against the reference's pass sequence, not only against our other kernel)
    nodes, k = [], 0
    for r in res:
        if r == (iw, ih):
            nodes.append(orc.planar_yuv_to_rgba(*planes[k], iw, ih)); k += 1
        else:
            nodes.append(label_host)
    want, _ = refpipe.render_yuv420(layouts, nodes, W, H)
    for g, w_, pl in zip(got, want, "YUV"):
        assert refpipe.max_diff(g, w_) <= 1 and refpipe.exact_fraction(g, w_) >= _oracle_floor(ctx), f"{name} plane {pl}"


@pytest.mark.parametrize("seed", range(72))
def test_random_geometries_fused_equals_unfused(ctx, ctx_unfused, hip, seed):
    """Random input sizes (even and odd), output sizes, fit / fill (cropping) rescalers, positions and strip widths: the fused
    kernels must reproduce the pass-per-launch path (valu: bit for bit) AND the oracle's restatement of the reference's pass
    sequence (<= 1 LSB) — footprints, crop offsets, edge clamps, ragged last strips."""
    import os
    rng = np.random.default_rng(1000 + seed)
    iw, ih = int(rng.integers(8, 700)), int(rng.integers(8, 500))
    ih = int(np.clip(ih, iw // 3, iw * 3))  # (a `fill` of an extreme aspect ratio asks for a tile beyond the maximum node size)
    if seed % 3:
        iw, ih = iw & ~1, ih & ~1  # the staged 4:2:0 path needs even sizes; odd ones take the direct-sampling branch
    iw, ih = max(iw, 2), max(ih, 2)
    W, H = int(rng.integers(8, 300)) * 4, int(rng.integers(8, 250)) * 2
    n = int(rng.integers(1, 4))
    kids = []
    for i in range(n):
        w, h = float(rng.integers(4, W)), float(rng.integers(4, H))
        kids.append({"type": "rescaler", "mode": str(rng.choice(["fit", "fill"])), "width": w, "height": h,
                     "top": float(rng.integers(0, max(1, H - int(h)))), "left": float(rng.integers(0, max(1, W - int(w)))),
                     "horizontal_align": str(rng.choice(["left", "center", "right"])), "vertical_align": str(rng.choice(["top", "center", "bottom"])),
                     "child": {"type": "input_stream", "input_id": f"in{i}"}})
    scene = {"type": "view", "background_color": "#102030FF", "children": kids}
    from smelter_amd.scene import Scene
    sc = Scene()
    sc.update(scene, W, H)
    layouts = sc.layouts(0, 0, [(iw, ih)] * n)
    planes = [scenes.random_yuv420(iw, ih, 77 + seed * 10 + i) for i in range(n)]
    planes = [(y, u[: ih // 2, : iw // 2], v[: ih // 2, : iw // 2]) for y, u, v in planes]

    def frames(c):
        return [c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(p)) for p in planes]

    ctx.set_strip_width(32 if seed % 2 else 64)
    try:
        got = _render(ctx, hip, layouts, frames(ctx), W, H)
    finally:
        ctx.set_strip_width(0)
    ref = _render(ctx_unfused, hip, layouts, frames(ctx_unfused), W, H)
    _assert_matches_unfused(ctx, got, ref, (seed, iw, ih, W, H, scene), identical=0.98, noise=True)
    # which node texture a layout's tile was resampled from: the fused conversion's (two-pass and single-axis plans), or — for a
    # box-pre-reduced plan, whose first step is downsample.wgsl on the node — the exact converter's
    nodes = []
    for i, (y, u, v) in enumerate(planes):
        tex = [l for l in layouts if l.type == 0 and l.source_index == i]
        boxed = False
        for l in tex:
            dw, dh = max(int(np.floor(l.width + 0.5)), 1), max(int(np.floor(l.height + 0.5)), 1)
            plan = orc.resample_plan(iw, ih, tuple(l.crop), dw, dh)
            boxed = boxed or (plan.kind > 0 and plan.levels != (0, 0))
        nodes.append(orc.planar_yuv_to_rgba(y, u, v, iw, ih) if boxed else _node(ctx, y, u, v, iw, ih))
    want, _ = refpipe.render_yuv420(layouts, nodes, W, H)
    for g, w_, pl in zip(got, want, "YUV"):
        assert _within_one_lsb(g, w_), (seed, pl, iw, ih, W, H, refpipe.max_diff(g, w_))
        assert refpipe.exact_fraction(g, w_) >= 0.98, (seed, pl, refpipe.exact_fraction(g, w_))


INPUT_FORMATS = [
    ("nv12", "FRAME_NV12", orc.YUV420),            # decoder hand-off: Y + interleaved UV (wgpu/texture/nv12.rs)
    ("yuvj420", "FRAME_PLANAR_YUVJ420", orc.YUVJ420),
    ("yuv422", "FRAME_PLANAR_YUV422", orc.YUV422),  # staged path is 4:2:0 only: these take wave A's direct-sampling branch
    ("yuv444", "FRAME_PLANAR_YUV444", orc.YUV444),
]


@pytest.mark.parametrize("name,fmt_name,variant", INPUT_FORMATS, ids=[f[0] for f in INPUT_FORMATS])
@pytest.mark.parametrize("geom", [(480, 270, 640, 360, 4), (322, 182, 500, 282, 3)], ids=["even", "ragged"])
def test_fused_ingest_input_formats(ctx, ctx_unfused, hip, name, fmt_name, variant, geom):
    """Every frame format wave A accepts: fused == pass-per-launch bit for bit, and within 1 LSB of the oracle."""
    iw, ih, W, H, n = geom
    fmt = getattr(hip, fmt_name)
    layouts, res = scenes.cfg2_scene(iw, ih, W, H, n)
    rng = np.random.default_rng(99)
    ch, cw = orc.chroma_shape(iw, ih, variant)
    inputs = []
    for i in range(n):
        y, _, _ = scenes.test_input(i, iw, ih, noise_seed=50 + i)
        u = rng.integers(0, 256, (ch, cw), dtype=np.uint8)
        v = rng.integers(0, 256, (ch, cw), dtype=np.uint8)
        inputs.append((y, u, v))

    def frames(c):
        out = []
        for y, u, v in inputs:
            planes = [y, np.stack([u, v], axis=-1)] if name == "nv12" else [y, u, v]
            out.append(c.frame(fmt, iw, ih, planes))
        return out

    ctx.profile_reset()
    ctx.profile_enable(True)
    got = _render(ctx, hip, layouts, frames(ctx), W, H)
    ctx.sync()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    assert prof["fused_ingest_resample"][1] == 1 and prof["fused_compose_output"][1] == 1, f"{name} did not take the fused kernels: {prof}"
    got_unfused = _render(ctx_unfused, hip, layouts, frames(ctx_unfused), W, H)
    _assert_matches_unfused(ctx, got, got_unfused, name, noise=True)
    nodes = [_node(ctx, y, u, v, iw, ih, orc.YUV420 if name == "nv12" else variant) for y, u, v in inputs]
    want, _ = refpipe.render_yuv420(layouts, nodes, W, H)
    floor = 0.995 if ctx.impl == "valu" else 0.99  # (white-noise chroma planes: the worst case for the f16 weights of pass 2)
    for g, w_, pl in zip(got, want, "YUV"):
        assert refpipe.max_diff(g, w_) <= 1, f"{name} plane {pl}"
        assert refpipe.exact_fraction(g, w_) >= floor, f"{name} plane {pl}"


def test_sharded_path_on_one_gpu_matches_the_fused_path(hip):
    """The multi-GPU driver with world_size 1 (every input on the root, no exchange): smr_ingest_resample per input into torch
    tensors wrapped as surfaces + compose from the tiles must equal smr_render_layouts on the raw frames, bit for bit.
    (world_size > 1 adds only the tile send/recv, covered by tests/test_dist.py over gloo.)"""
    import torch
    from smelter_amd import dist as smr_dist
    torch.cuda.set_device(0)
    side = torch.cuda.Stream()  # the renderer enqueues on torch's (non-default) current stream, as bench.py does under torchrun
    torch.cuda.set_stream(side)
    c = hip.Context(0, stream=side.cuda_stream)
    iw, ih, W, H, n = 480, 270, 960, 540, 8
    layouts, res = scenes.cfg3_scene(iw, ih, W, H, n)
    planes, frames = _inputs(c, hip, n, iw, ih)
    label_t, _ = _label_surfaces(c, 1)
    slots = [i for i, r in enumerate(res) if r == (iw, ih)]
    srcs, k = [], 0
    for r in res:
        if r == (iw, ih):
            srcs.append(frames[k]); k += 1
        else:
            srcs.append(label_t)
    want = _render(c, hip, layouts, srcs, W, H)
    plan = smr_dist.ShardPlan(n_inputs=n, world=1)
    sharded = smr_dist.ShardedCompositor(c, hip, plan, 0, layouts, res, slots, label_t, torch, None)
    out = c.frame(hip.FRAME_PLANAR_YUV420, W, H)
    sharded.step({i: frames[i] for i in range(n)}, out)
    c.sync()
    for a, b in zip(out.download(), want):
        assert (a == b).all()
    # pipelined mode: frame k is resampled while frame k-1 is composed; two tile sets, one output frame per frame in flight
    _, frames_b = _inputs(c, hip, n, iw, ih, seed=4321)
    srcs_b = [frames_b[slots.index(i)] if i in slots else label_t for i in range(len(res))]
    want_b = _render(c, hip, layouts, srcs_b, W, H)
    outs = [c.frame(hip.FRAME_PLANAR_YUV420, W, H) for _ in range(3)]
    rows = [{i: frames[i] for i in range(n)}, {i: frames_b[i] for i in range(n)}, {i: frames[i] for i in range(n)}]
    for k in range(3):
        sharded.step_pipelined(rows[k], outs[k])
    sharded.flush()
    c.sync()
    for k, w_ in enumerate([want, want_b, want]):
        for a, b in zip(outs[k].download(), w_):
            assert (a == b).all(), k
    # animated scene: a new layout list per frame (tiles move by fractions of a pixel, their sizes stay), set before the frame
    # whose exchange then overlaps the previous frame's compose — each frame must come out with its own list
    from dataclasses import replace
    moved = [[replace(L, left=L.left + 0.37 * (j + 1), top=L.top + 1.5 * j) if L.type == 0 and L.source_index in slots else L
              for L in layouts] for j in range(3)]
    wants = [_render(c, hip, moved[j], srcs_b if j == 1 else srcs, W, H) for j in range(3)]
    for j in range(3):
        sharded.set_layouts(moved[j])
        sharded.step_pipelined(rows[j], outs[j])
    sharded.flush()
    c.sync()
    for j in range(3):
        for a, b in zip(outs[j].download(), wants[j]):
            assert (a == b).all(), j
    assert not (wants[0][0] == want[0]).all()
    c.close()
    torch.cuda.set_stream(torch.cuda.default_stream())


def test_long_layout_lists_and_odd_output_sizes(ctx, ctx_unfused, hip):
    """More layouts than the fused compose kernel keeps in LDS (48): still one fused launch, the list read where it lies in memory.
    An even output width that is not a multiple of 4 (642; 854 and 1366 are common ones): still the fused kernels, the last
    block of a row two pixels wide.  Both match the pass-per-launch path and the oracle."""
    iw, ih = 160, 90
    for (W, H, n) in [(1280, 720, 12), (642, 362, 3)]:
        layouts, res = scenes.cfg3_scene(iw, ih, W, H, n)
        assert n < 12 or 48 < len(layouts) <= 100  # (beyond max_layouts_count = 100 the reference drops layouts, params.rs:176-182)
        planes, _ = _inputs(ctx, hip, n, iw, ih)
        _, label_host = _label_surfaces(ctx, 1)

        def sources_for(c):
            srcs, k = [], 0
            lt = c.surface_from(label_host)
            for r in res:
                if r == (iw, ih):
                    srcs.append(c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(planes[k]))); k += 1
                else:
                    srcs.append(lt)
            return srcs

        if W % 2 == 0:
            ctx.profile_reset()
            ctx.profile_enable(True)
            got = _render(ctx, hip, layouts, sources_for(ctx), W, H)
            ctx.sync()
            prof = ctx.profile_read()
            ctx.profile_enable(False)
            assert prof["fused_compose_output"][1] == 1 and prof["layouts"][1] == 0, f"{len(layouts)} layouts left the fused compositor: {prof}"
            ref = _render(ctx_unfused, hip, layouts, sources_for(ctx_unfused), W, H)
            _assert_matches_unfused(ctx, got, ref, (W, H, n))
        rgba = ctx.surface(W, H)
        ctx.render_layouts(layouts, sources_for(ctx), W, H, out_rgba=rgba)
        nodes, k = [], 0
        for r in res:
            if r == (iw, ih):
                nodes.append(orc.planar_yuv_to_rgba(*planes[k], iw, ih)); k += 1
            else:
                nodes.append(label_host)
        want = refpipe.layout_node_render(layouts, nodes, W, H)
        g = rgba.download()
        assert refpipe.max_diff(g, want) <= 1 and refpipe.exact_fraction(g, want) >= _oracle_floor(ctx)


def test_missing_input_renders_like_the_reference(ctx, hip):
    # stale / missing input: node texture None -> InputStream size 0x0 -> layout culled (scene/layout.rs:109-115)
    root = S.Tiles(children=[S.InputStream(0), S.InputStream(1)], background_color=(10, 20, 30, 255))
    res = [(320, 180), None]
    layouts = S.scene_layouts(root, 640, 360, res)
    planes, frames = _inputs(ctx, hip, 1, 320, 180)
    got = _render(ctx, hip, layouts, [frames[0], None], 640, 360)
    want, _ = refpipe.render_yuv420(layouts, [refpipe.nodes_from_yuv420(planes)[0], None], 640, 360)
    for g, w_ in zip(got, want):
        assert refpipe.max_diff(g, w_) <= 1


def test_non_finite_layouts_draw_nothing(ctx, ctx_unfused, hip):
    """A quad with a NaN or infinite corner rasterises to nothing on the reference's GPU (a transition evaluated outside its
    contract can produce one: transition.rs:88-101 divides by 1 - state_offset).  Such layouts must neither draw nor reach the
    float -> int conversions of the tile binning (which once cost a 2^31-iteration host loop per frame)."""
    import time
    from dataclasses import replace
    layouts = [orc.Layout(0.0, 0.0, 640.0, 360.0, 1, color=orc.color_to_shader((10, 20, 30, 255))),
               orc.Layout(10.0, 20.0, 100.0, 60.0, 1, color=orc.color_to_shader((200, 40, 40, 255)), border_radius=(8.0, 8.0, 8.0, 8.0)),
               orc.Layout(100.5, 200.25, 80.0, 80.0, 1, color=orc.color_to_shader((40, 200, 40, 255)))]
    nan, inf = float("nan"), float("inf")
    bad = [replace(layouts[1], left=nan), replace(layouts[1], top=inf), replace(layouts[2], width=nan), replace(layouts[2], left=-inf),
           replace(layouts[1], rotation_degrees=nan), replace(layouts[2], left=nan, top=nan, width=nan, height=nan)]
    for c in (ctx, ctx_unfused):
        want = _render(c, hip, [layouts[0]], [], 640, 360)
        both = _render(c, hip, layouts, [], 640, 360)
        assert not (both[0] == want[0]).all()
        for b in bad:
            t0 = time.perf_counter()
            got = _render(c, hip, [layouts[0], b], [], 640, 360)
            assert time.perf_counter() - t0 < 0.5
            for g, w_ in zip(got, want):
                assert (g == w_).all()


def test_empty_scene_is_transparent_black(ctx, hip):
    out = ctx.frame(hip.FRAME_PLANAR_YUV420, 64, 36)
    ctx.render_layouts([], [], 64, 36, out=out)
    y, u, v = out.download()
    # cleared transparent target -> RGBA (0,0,0,0) -> Y=16, U=V=128
    assert (y == 16).all() and (u == 128).all() and (v == 128).all()


@pytest.mark.parametrize("name,mk,iw,ih,W,H,n", [
    ("configs1", lambda: scenes.cfg2_scene(1920, 1080, 1920, 1080, 4), 1920, 1080, 1920, 1080, 4),   # 2x, 13 taps, 32-column strips
    ("configs3_one_gpu", lambda: scenes.cfg3_scene(3840, 2160, 3840, 2160, 8), 3840, 2160, 3840, 2160, 8),  # 3x, 19 taps
], ids=["configs1", "configs3_one_gpu"])
def test_other_baseline_configs_at_full_size_match_the_oracle(ctx, hip, name, mk, iw, ih, W, H, n):
    """BASELINE.json configs[1] and configs[3] (on one GPU) at their full sizes against the oracle's pass sequence."""
    layouts, res = mk()
    planes, frames = _inputs(ctx, hip, n, iw, ih)
    label_t, label_host = _label_surfaces(ctx, 1)
    srcs, nodes, k = [], [], 0
    for r in res:
        if r == (iw, ih):
            srcs.append(frames[k])
            nodes.append(orc.planar_yuv_to_rgba(*planes[k], iw, ih, omp=True))
            k += 1
        else:
            srcs.append(label_t)
            nodes.append(label_host)
    got = _render(ctx, hip, layouts, srcs, W, H)
    want, _ = refpipe.render_yuv420(layouts, nodes, W, H, omp=True)
    for g, w_ in zip(got, want):
        assert refpipe.max_diff(g, w_) <= 1 and refpipe.exact_fraction(g, w_) >= _oracle_floor(ctx)


# ---- BASELINE.json full sizes: properties that do not need the oracle at 4K ----------------------
def test_full_size_properties(ctx, ctx_unfused, hip):
    iw, ih, W, H, n = 1920, 1080, 3840, 2160, 8
    layouts, res = scenes.cfg3_scene(iw, ih, W, H, n)
    label_t, label_host = _label_surfaces(ctx, 1)
    planes, frames = _inputs(ctx, hip, n, iw, ih)

    def srcs(c, fr):
        lt = c.surface_from(label_host)
        out, k = [], 0
        for r in res:
            if r == (iw, ih):
                out.append(fr[k]); k += 1
            else:
                out.append(lt)
        return out

    a = _render(ctx, hip, layouts, srcs(ctx, frames), W, H)
    # idempotence: same inputs, same bytes
    b = _render(ctx, hip, layouts, srcs(ctx, frames), W, H)
    for p, q in zip(a, b):
        assert (p == q).all()
    # fused == unfused at full size
    fr_u = [ctx_unfused.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(p)) for p in planes]
    c = _render(ctx_unfused, hip, layouts, srcs(ctx_unfused, fr_u), W, H)
    _assert_matches_unfused(ctx, a, c, "configs[2] at full size")
    # tile independence: permuting which input feeds which tile permutes the tiles (linearity of placement)
    perm = [3, 0, 1, 2, 7, 4, 5, 6]
    d = _render(ctx, hip, layouts, srcs(ctx, [frames[i] for i in perm]), W, H)
    tiles = S.tiles_positions(S.Tiles(), n, W, H)
    for slot, src_idx in enumerate(perm):
        t0, l0, tw, th = [int(round(float(v))) for v in tiles[slot]]
        t1, l1, _, _ = [int(round(float(v))) for v in tiles[src_idx]]
        # interior of the tile (away from labels/borders which are identical anyway)
        ys, xs = slice(t0 + 64, t0 + 256), slice(l0 + 400, l0 + 900)
        ys1, xs1 = slice(t1 + 64, t1 + 256), slice(l1 + 400, l1 + 900)
        assert (d[0][ys, xs] == a[0][ys1, xs1]).all()
    # oracle check of one full-resolution tile crop (cheap: a 2-input sub-scene at the same geometry)
    sub_layouts, sub_res = scenes.cfg3_scene(iw, ih, 2560, 720, 2)
    sub = _render(ctx, hip, sub_layouts, srcs(ctx, frames)[:4], 2560, 720)
    nodes = [refpipe.nodes_from_yuv420([planes[0]], omp=True)[0], label_host, refpipe.nodes_from_yuv420([planes[1]], omp=True)[0], label_host]
    want, _ = refpipe.render_yuv420(sub_layouts, nodes, 2560, 720, omp=True)
    for g, w_ in zip(sub, want):
        assert refpipe.max_diff(g, w_) <= 1
    # and the whole frame of the benchmark workload against the oracle's pass sequence (OpenMP build: about a second)
    nodes_full, k = [], 0
    for r in res:
        if r == (iw, ih):
            nodes_full.append(orc.planar_yuv_to_rgba(*planes[k], iw, ih, omp=True)); k += 1
        else:
            nodes_full.append(label_host)
    want_full, _ = refpipe.render_yuv420(layouts, nodes_full, W, H, omp=True)
    for g, w_ in zip(a, want_full):
        assert refpipe.max_diff(g, w_) <= 1 and refpipe.exact_fraction(g, w_) >= _oracle_floor(ctx)


DIRECT_CASES = [
    ("cfg2_small", lambda: scenes.cfg2_scene(320, 180, 640, 360, 4), 320, 180, 640, 360),
    ("cfg3_small", lambda: scenes.cfg3_scene(480, 270, 960, 540, 8), 480, 270, 960, 540),
    ("cfg3_odd_positions", lambda: scenes.cfg3_scene(482, 274, 1000, 562, 5), 482, 274, 1000, 562),
    ("cfg3_1080p", lambda: scenes.cfg3_scene(960, 540, 1920, 1080, 8), 960, 540, 1920, 1080),
]


@pytest.mark.parametrize("route", ["rgb12_node", "rgba8_node"] + (["fused_conversion"] if "fused" in _routes() else []))
@pytest.mark.parametrize("fmt_name", ["planar", "nv12"])
@pytest.mark.parametrize("name,mk,iw,ih,W,H", DIRECT_CASES, ids=[c[0] for c in DIRECT_CASES])
def test_direct_output_of_a_scene_at_rest_is_bit_identical(hip, name, mk, iw, ih, W, H, fmt_name, route):
    """SMR_OPT_DIRECT_OUTPUT: from the second frame of an unchanged layout list on, the resampling kernel writes the Y'CbCr of the
    compositor's copy tiles itself and their RGBA8 texels are never stored.  Every frame must equal the first one (rendered through
    the RGBA8 tile) and the frames of a context with the option off, byte for byte; the output frames start out poisoned, so a
    tile neither kernel wrote would show."""
    fmt = hip.FRAME_PLANAR_YUV420 if fmt_name == "planar" else hip.FRAME_NV12
    layouts, res = mk()
    c_on, c_off = hip.Context(0), hip.Context(0)
    try:
        for c in (c_on, c_off):  # (direct output is a build of the matrix-core kernel: one per source kind)
            _impl_or_skip(c, hip, hip.INGEST_MFMA_F16_FUSED if route == "fused_conversion" else hip.INGEST_AUTO)
            c.set_compact_nodes(route == "rgb12_node")
        c_on.set_direct_output(True)
        c_off.set_direct_output(False)
        n_in = sum(1 for r in res if r == (iw, ih))
        planes, _ = _inputs(c_on, hip, n_in, iw, ih)
        _, label_host = _label_surfaces(c_on, 1)

        def frames_of(c):
            srcs, k = [], 0
            lt = c.surface_from(label_host)
            for r in res:
                if r == (iw, ih):
                    srcs.append(c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(planes[k])))
                    k += 1
                else:
                    srcs.append(lt)
            return srcs

        def render(c, srcs, poison):
            out = c.frame(fmt, W, H)
            out.upload([np.full(p.shape, poison, np.uint8) for p in out.download()])
            c.render_layouts(layouts, srcs, W, H, out=out)
            return out.download()

        s_on, s_off = frames_of(c_on), frames_of(c_off)
        ref = render(c_off, s_off, 0x11)
        for i in range(4):
            got = render(c_on, s_on, 0x33 + i)
            for g, w_, pl in zip(got, ref, "YUV"):
                assert np.array_equal(g, w_), f"{name} frame {i} plane {pl}: {int((g != w_).sum())} bytes differ"
        # new input content under the same layout list: still direct, still equal
        planes2, _ = _inputs(c_on, hip, n_in, iw, ih, seed=99)
        for f_on, f_off, p in zip([s for s, r in zip(s_on, res) if r == (iw, ih)], [s for s, r in zip(s_off, res) if r == (iw, ih)], planes2):
            f_on.upload(list(p))
            f_off.upload(list(p))
        ref2, got2 = render(c_off, s_off, 0x55), render(c_on, s_on, 0x77)
        for g, w_, pl in zip(got2, ref2, "YUV"):
            assert np.array_equal(g, w_), f"{name} new content plane {pl}: {int((g != w_).sum())} bytes differ"
        assert not all(np.array_equal(a, b) for a, b in zip(ref, ref2))
    finally:
        c_on.close()
        c_off.close()


@pytest.mark.parametrize("fmt_name", ["planar", "nv12", "rgba"])
@pytest.mark.parametrize("geom", [(640, 360, 1920, 1080, 16), (482, 274, 1000, 562, 5), (320, 180, 3840, 2160, 9)], ids=["4x4", "ragged", "3x3_4k"])
def test_seam_tiles_copy_from_the_topmost_layer(hip, monkeypatch, geom, fmt_name):
    """A grid of video tiles that abut: the seams run through the compositor's 128 x 16 tiles, where every pixel still is a plain copy
    from one layer (TC_SELECT).  Bit for bit what the compositing path makes of those tiles (SMR_OPT_COMPOSE_SELECT = 0), on every output route."""
    iw, ih, W, H, n = geom
    layouts, res = scenes.cfg2_scene(iw, ih, W, H, n)
    fmt = {"planar": hip.FRAME_PLANAR_YUV420, "nv12": hip.FRAME_NV12, "rgba": hip.FRAME_RGBA}[fmt_name]
    c_off = hip.Context(0)
    c_off.set_compose_select(False)  # SMR_OPT_COMPOSE_SELECT = 0: the seam tiles through the compositing path
    c_on = hip.Context(0)
    try:
        planes, _ = _inputs(c_on, hip, n, iw, ih)
        outs = []
        for c in (c_on, c_off):
            srcs = [c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(p)) for p in planes]
            frames = []
            for rep in range(3):  # (the class records are kept while the list repeats: the cached path too)
                out = c.frame(fmt, W, H)
                out.upload([np.full(p.shape, 0x40 + rep, np.uint8) for p in out.download()])
                c.render_layouts(layouts, srcs, W, H, out=out)
                frames.append(out.download())
            for f in frames[1:]:
                assert all(np.array_equal(a, b) for a, b in zip(f, frames[0]))
            outs.append(frames[0])
        for a, b, pl in zip(outs[0], outs[1], "YUV"):
            assert np.array_equal(a, b), f"plane {pl}: {int((a != b).sum())} bytes differ"
    finally:
        c_on.close()
        c_off.close()


@pytest.mark.parametrize("geom", [(7680, 4320, 1920, 1080), (7680, 4320, 3840, 2160), (7680, 2, 3840, 2)], ids=["8k_to_1080p", "8k_to_4k", "8k_x_2"])
def test_maximum_node_resolution(hip, geom):
    """The reference's largest node (MAX_NODE_RESOLUTION, types.rs:146-149: 7682 x 4320): an 8K 4:2:0 frame through the default route.  The
    converter's node texture is the oracle's byte for byte; the tile (a factor-4 plan with a box pre-reduction, a factor-2 two-pass plan, a
    two-row sliver) is within 1 LSB of the oracle's resample on every byte; the f32 kernels agree."""
    iw, ih, dw, dh = geom
    rng = np.random.default_rng(iw * 31 + dh)
    xx, yy = np.meshgrid(np.arange(iw), np.arange(ih))
    y = ((xx * 5 + yy * 3) % 220 + 16 + rng.integers(0, 12, (ih, iw))).astype(np.uint8)
    u = rng.integers(0, 256, (ih // 2, iw // 2), dtype=np.uint8)
    v = ((xx[::2, ::2] // 3 + yy[::2, ::2]) % 256).astype(np.uint8)
    crop = (0.0, 0.0, float(iw), float(ih))
    node_o = orc.planar_yuv_to_rgba(y, u, v, iw, ih, omp=True)
    _, want = orc.resample(node_o, crop, dw, dh, omp=True)
    c = hip.Context(0)
    try:
        f = c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, [y, u, v])
        assert np.array_equal(c.frame_to_rgba(f).download(), node_o), "the converter's 8K node texture is not the oracle's"
        for impl in (hip.INGEST_AUTO, hip.INGEST_VALU_F32):
            c.set_ingest_impl(impl)
            t = c.surface(dw, dh)
            c.ingest_resample(f, crop, t)
            de = np.abs(t.download().astype(np.int16) - want.astype(np.int16))
            assert de.max() <= 1, f"impl {impl}: max {de.max()}, {(de > 1).sum()} bytes off by more than 1"
            assert (de == 0).mean() >= 0.999, f"impl {impl}: {(de == 0).mean():.5f} identical"
    finally:
        c.close()


def test_full_size_white_noise_within_one_lsb(hip):
    """White noise at the benchmark geometry (1920x1080 -> 1280x720) is the worst case for every approximation: dark output pixels that
    are cancelling sums of bright texels.  The default route (SMR_INGEST_AUTO — what bench.py measures) and the f32 kernel are within 1 LSB
    of the oracle's planar_yuv_to_rgba -> resample END TO END, on every byte.  The opt-in fused conversion is held stage by stage: its node
    texture (folded FMA chain, tests/convert_model.py) is within 1 LSB of the oracle's with > 99.99 % of the codes identical, its tile within
    1 LSB — on every byte — of the oracle's resample of that node texture, and end to end a handful of bytes in 7.4 million sit 2-4 codes
    from the oracle (each a dark pixel next to a bright texel whose code differs by one between the two converters): bounded explicitly."""
    from tests import convert_model
    rng = np.random.default_rng(50)
    iw, ih, dw, dh = 1920, 1080, 1280, 720
    y = rng.integers(0, 256, (ih, iw), dtype=np.uint8)
    u = rng.integers(0, 256, (ih // 2, iw // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (ih // 2, iw // 2), dtype=np.uint8)
    crop = (0.0, 0.0, float(iw), float(ih))
    node_o = orc.planar_yuv_to_rgba(y, u, v, iw, ih, omp=True)
    _, want_o = orc.resample(node_o, crop, dw, dh, omp=True)
    c = hip.Context(0)
    try:
        f = c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, [y, u, v])
        assert np.array_equal(c.frame_to_rgba(f).download(), node_o), "the input converter's node texture is not the oracle's"
        for impl, ident in ((hip.INGEST_AUTO, 0.9995), (hip.INGEST_VALU_F32, 0.9999)):
            c.set_ingest_impl(impl)
            t = c.surface(dw, dh)
            c.ingest_resample(f, crop, t)
            de = np.abs(t.download().astype(np.int16) - want_o.astype(np.int16))
            assert de.max() <= 1, f"impl {impl}: max {de.max()}, {(de > 1).sum()} bytes off by more than 1 END TO END"
            assert (de == 0).mean() >= ident, f"impl {impl}: {(de == 0).mean():.5f} identical"
        if not c.lab_build():
            return  # (what follows holds the fused-conversion route — ingest implementation 5, laboratory builds only — to its own statement)
        # the fused conversion: per stage, and an explicit end-to-end bound against the oracle
        node_k = convert_model.node_codes(y, u, v)
        dn = np.abs(node_k.astype(np.int16) - node_o.astype(np.int16))
        assert dn.max() <= 1 and (dn == 0).mean() >= 0.9999, f"node texture: max {dn.max()}, {(dn == 0).mean():.6f} identical"
        _, want_k = orc.resample(node_k, crop, dw, dh, omp=True)
        c.set_ingest_impl(hip.INGEST_MFMA_F16_FUSED)
        t = c.surface(dw, dh)
        c.ingest_resample(f, crop, t)
        got = t.download()
        d = np.abs(got.astype(np.int16) - want_k.astype(np.int16))
        assert d.max() <= 1 and (d == 0).mean() >= 0.9995, f"fused conversion, per stage: max {d.max()}, {(d == 0).mean():.5f} identical"
        de = np.abs(got.astype(np.int16) - want_o.astype(np.int16))
        assert de.max() <= 4 and (de > 1).mean() <= 1e-6 and (de == 0).mean() >= 0.999, ((de > 1).sum(), de.max(), (de == 0).mean())
    finally:
        c.close()


def test_tile_class_cache_survives_alternating_and_evicted_layout_lists(hip):
    """The compositor keeps the tile classes of the last four layout lists per context (smr_fused.hip).  Lists that alternate (a
    nested node and its root), come back after being evicted, or differ only in where a tile sits must each be composited with
    their own classes: every frame equals the frame of a fresh context that has never seen another list."""
    iw, ih, W, H = 320, 180, 640, 360
    c = hip.Context(0)
    try:
        planes, _ = _inputs(c, hip, 4, iw, ih)
        _, label_host = _label_surfaces(c, 1)
        variants = []
        for k in range(6):  # six lists: more than the cache holds
            layouts, res = scenes.cfg3_scene(iw, ih, W, H, 2 + (k % 3))
            if k >= 3:  # same structure, shifted: different copy-tile pointers for the same tile sizes
                layouts = [Layout_shift(l, 8 * (k - 2), 4 * (k - 2)) for l in layouts]
            variants.append((layouts, res))

        def sources(ctx_, res):
            srcs, k = [], 0
            lt = ctx_.surface_from(label_host)
            for r in res:
                if r == (iw, ih):
                    srcs.append(ctx_.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(planes[k])))
                    k += 1
                else:
                    srcs.append(lt)
            return srcs

        want = []
        for layouts, res in variants:
            f = hip.Context(0)
            try:
                want.append(_render(f, hip, layouts, sources(f, res), W, H))
            finally:
                f.close()
        srcs = [sources(c, res) for _, res in variants]
        order = [0, 1, 0, 1, 0, 2, 3, 4, 5, 0, 5, 1, 4, 2, 3, 3, 0]
        for step, v in enumerate(order):
            got = _render(c, hip, variants[v][0], srcs[v], W, H)
            for g, w_, pl in zip(got, want[v], "YUV"):
                assert np.array_equal(g, w_), f"step {step} list {v} plane {pl}: {int((g != w_).sum())} bytes differ"
    finally:
        c.close()


def test_a_scene_at_rest_reuses_its_parameter_pack_with_new_pixels(hip, monkeypatch):
    """smr_pack_commit queues no copy when the packed layout list equals the previous frame's byte for byte (a scene that does not
    move): the device copy of that frame is read again.  The pixels are new every frame — same device frames, new uploads — and a
    different list in between must not be served from the stale copy.  Every frame equals the frame of a context that never gets to reuse (another list is rendered in front of every frame)."""
    iw, ih, W, H = 480, 270, 960, 540
    lay_a, res_a = scenes.cfg3_scene(iw, ih, W, H, 4)
    lay_b = [Layout_shift(l, 16, 8) for l in lay_a]
    frames_px = [[scenes.test_input(i, iw, ih, noise_seed=100 * t + i) for i in range(4)] for t in range(4)]
    sequence = [(lay_a, 0), (lay_a, 1), (lay_a, 2), (lay_b, 2), (lay_b, 3), (lay_a, 3), (lay_a, 0)]

    lay_other = [Layout_shift(l, 3, 5) for l in lay_a]  # a third list: rendered in between, the next frame's pack never equals its predecessor's

    def run(c, never_reuse=False):
        _, label_host = _label_surfaces(c, 1)
        lt = c.surface_from(label_host)
        dev = [c.frame(hip.FRAME_PLANAR_YUV420, iw, ih) for _ in range(4)]
        outs = []
        for lay, t in sequence:
            for i in range(4):
                dev[i].upload(list(frames_px[t][i]))
            srcs, k = [], 0
            for r in res_a:
                if r == (iw, ih):
                    srcs.append(dev[k])
                    k += 1
                else:
                    srcs.append(lt)
            if never_reuse:
                _render(c, hip, lay_other, srcs, W, H)
            outs.append(_render(c, hip, lay, srcs, W, H))
        return outs

    c = hip.Context(0)
    try:
        got = run(c)
    finally:
        c.close()
    f = hip.Context(0)
    try:
        want = run(f, never_reuse=True)
    finally:
        f.close()
    for step, (g, w_) in enumerate(zip(got, want)):
        for a, b, pl in zip(g, w_, "YUV"):
            assert np.array_equal(a, b), f"frame {step} plane {pl}: {int((a != b).sum())} bytes differ"
    assert not np.array_equal(got[0][0], got[1][0]) and not np.array_equal(got[2][0], got[3][0])  # (the frames do differ)


def Layout_shift(l, dx, dy):
    import copy
    m = copy.deepcopy(l)
    m.left += dx
    m.top += dy
    for k in m.masks:
        k.left += dx
        k.top += dy
    return m


@pytest.mark.parametrize("impl", ["auto"] + (["fused"] if "fused" in _routes() else []))
@pytest.mark.parametrize("fmt_name", ["planar", "nv12"])
@pytest.mark.parametrize("geom", [(1920, 1080, 1279, 719), (640, 360, 427, 239), (322, 182, 255, 143)], ids=["1080p", "360p", "ragged"])
def test_vertical_first_plans_run_on_the_transposed_frame(hip, geom, fmt_name, impl):
    """The reference filters the axis with the stronger shrink first; for aspect-preserving fits the order hangs on the rounding of the
    tile size.  The matrix-core kernel filters horizontally first, so a vertical-first plan is run on the transposed node texture (the
    default route; the transposed planes with the opt-in fused conversion), tile transposed back: still the matrix-core kernel, and as close
    to the oracle as a horizontal-first plan."""
    iw, ih, dw, dh = geom
    crop = (0.0, 0.0, float(iw), float(ih))
    plan = orc.resample_plan(iw, ih, crop, dw, dh)
    assert tuple(plan.axis[:2]) == (1, 0), "the geometry is meant to give a vertical-first plan"
    y, u, v = scenes.test_input(3, iw, ih, noise_seed=77)
    rng = np.random.default_rng(5)
    u = (u.astype(np.int16) + rng.integers(-20, 21, u.shape)).clip(0, 255).astype(np.uint8)  # (textured chroma: a transposition slip would show)
    v = (v.astype(np.int16) + rng.integers(-20, 21, v.shape)).clip(0, 255).astype(np.uint8)
    _, want = orc.resample(orc.planar_yuv_to_rgba(y, u, v, iw, ih), crop, dw, dh, omp=True)
    c = hip.Context(0)
    try:
        _impl_or_skip(c, hip, hip.INGEST_AUTO if impl == "auto" else hip.INGEST_MFMA_F16_FUSED)
        f = c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, [y, u, v]) if fmt_name == "planar" else c.frame(hip.FRAME_NV12, iw, ih, [y, np.stack([u, v], axis=-1)])
        t = c.surface(dw, dh)
        c.profile_reset()
        c.profile_enable(True)
        c.ingest_resample(f, crop, t)
        c.sync()
        prof = c.profile_read()
        c.profile_enable(False)
        assert prof["fused_ingest_resample"][1] == 1 and prof["resample"][1] == 0 and prof["ingest"][1] == (1 if impl == "auto" else 0), f"left the matrix-core kernel: {prof}"
        d = np.abs(t.download().astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1, f"max {d.max()}"
        assert (d == 0).mean() >= 0.998, f"{(d == 0).mean():.5f} identical"  # (noisy chroma; measured 0.9988)
    finally:
        c.close()


@pytest.mark.parametrize("fmt_name", ["planar", "nv12", "j420"])
@pytest.mark.parametrize("geom", [(1920, 1080, 1280, 720), (640, 360, 426, 240), (1280, 720, 640, 360), (328, 182, 250, 140), (5760, 1080, 1280, 240)],
                         ids=["1080p", "360p", "half", "ragged", "wider_than_an_rgb12_row"])
def test_compact_node_textures_give_the_same_tiles(hip, geom, fmt_name):
    """SMR_OPT_COMPACT_NODES (default on): the node texture of a 4:2:0 frame that only the matrix-core resampler reads is RGB12 (12 bytes
    per four pixels) instead of RGBA8 — the same codes, so the tile is the same bit for bit; and both are within 1 LSB of the oracle's
    converter + resampler on white noise.  (A frame whose RGB12 rows would exceed the surface limit keeps the RGBA8 node.)"""
    iw, ih, dw, dh = geom
    crop = (0.0, 0.0, float(iw), float(ih))
    rng = np.random.default_rng(iw + dh)
    y = rng.integers(0, 256, (ih, iw), dtype=np.uint8)
    u = rng.integers(0, 256, (ih // 2, iw // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (ih // 2, iw // 2), dtype=np.uint8)
    if fmt_name == "nv12":
        node = orc.nv12_to_rgba(y, np.stack([u, v], axis=-1), iw, ih)
    else:
        node = orc.planar_yuv_to_rgba(y, u, v, iw, ih, orc.YUVJ420 if fmt_name == "j420" else orc.YUV420, omp=True)
    _, want = orc.resample(node, crop, dw, dh, omp=True)
    c = hip.Context(0)
    try:
        if fmt_name == "nv12":
            f = c.frame(hip.FRAME_NV12, iw, ih, [y, np.stack([u, v], axis=-1)])
        else:
            f = c.frame(hip.FRAME_PLANAR_YUVJ420 if fmt_name == "j420" else hip.FRAME_PLANAR_YUV420, iw, ih, [y, u, v])
        tiles = []
        for on in (True, False):
            c.set_compact_nodes(on)
            t = c.surface(dw, dh)
            before = c.kernel_launches()
            c.ingest_resample(f, crop, t)
            ran = {k: v_ - before[k] for k, v_ in c.kernel_launches().items()}
            assert ran["frame_to_rgba"] == 1 and ran["ingest_wave_rgba"] + ran["resample_general"] >= 1 and ran["ingest_wave"] == 0, ran
            tiles.append(t.download())
        assert np.array_equal(tiles[0], tiles[1]), int((tiles[0] != tiles[1]).sum())
        d = np.abs(tiles[0].astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1 and (d == 0).mean() >= 0.999, (int(d.max()), float((d == 0).mean()))
    finally:
        c.close()


@pytest.mark.parametrize("fmt_name", ["planar", "nv12", "j420"])
@pytest.mark.parametrize("geom", [(1920, 1080, 1280, 720), (640, 360, 426, 240), (1280, 720, 640, 360), (3840, 2160, 1280, 720), (328, 182, 216, 120)],
                         ids=["1080p", "360p", "half", "4k", "ragged"])
def test_plane_source_route_converts_exactly_inside_the_resampler(hip, geom, fmt_name):
    """SMR_OPT_PLANE_SOURCE (opt-in): k_ingest_wave reads the frame's planes and converts each chunk of its window in the wave with the exact
    converter's block arithmetic — no converter launch, no node texture in memory.  The virtual node is the oracle's bit for bit, so on white
    noise the tile is within 1 LSB of the oracle's converter + resampler, and equal to the node route's tile except where the two chunk grids
    sum pass 2 in another order (never more than one code)."""
    iw, ih, dw, dh = geom
    crop = (0.0, 0.0, float(iw), float(ih))
    rng = np.random.default_rng(iw * 3 + dh)
    y = rng.integers(0, 256, (ih, iw), dtype=np.uint8)
    u = rng.integers(0, 256, (ih // 2, iw // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (ih // 2, iw // 2), dtype=np.uint8)
    if fmt_name == "nv12":
        node = orc.nv12_to_rgba(y, np.stack([u, v], axis=-1), iw, ih)
    else:
        node = orc.planar_yuv_to_rgba(y, u, v, iw, ih, orc.YUVJ420 if fmt_name == "j420" else orc.YUV420, omp=True)
    _, want = orc.resample(node, crop, dw, dh, omp=True)
    c = hip.Context(0)
    try:
        if fmt_name == "nv12":
            f = c.frame(hip.FRAME_NV12, iw, ih, [y, np.stack([u, v], axis=-1)])
        else:
            f = c.frame(hip.FRAME_PLANAR_YUVJ420 if fmt_name == "j420" else hip.FRAME_PLANAR_YUV420, iw, ih, [y, u, v])
        tiles = []
        for on in (True, False):
            c.set_plane_source(on)
            t = c.surface(dw, dh)
            before = c.kernel_launches()
            c.ingest_resample(f, crop, t)
            ran = {k: v_ - before[k] for k, v_ in c.kernel_launches().items()}
            if on:
                assert ran["ingest_wave"] == 1 and ran["frame_to_rgba"] == 0 and ran["ingest_wave_rgba"] == 0, ran  # one launch, no converter
            else:
                assert ran["ingest_wave"] == 0 and ran["frame_to_rgba"] == 1, ran
            tiles.append(t.download())
        for t in tiles:
            d = np.abs(t.astype(np.int16) - want.astype(np.int16))
            assert d.max() <= 1 and (d == 0).mean() >= 0.999, (int(d.max()), float((d == 0).mean()))
        dr = np.abs(tiles[0].astype(np.int16) - tiles[1].astype(np.int16))
        assert dr.max() <= 1 and (dr == 0).mean() >= 0.9999, (int(dr.max()), float((dr == 0).mean()))
    finally:
        c.close()


def test_one_call_with_many_frames_of_assorted_sizes(hip):
    """Twenty inputs of different sizes, formats and ranges in ONE smr_render_layouts call: the converter's launches (sixteen frames, then four;
    planar and NV12 apart) deal bands of each frame's block rows to the XCDs and equal shares to their waves — frames of a few rows, widths that
    are no multiple of 256, heights of 2 mod 4, a 4K frame beside an 8 x 2 one.  Every tile within 1 LSB of the oracle's converter + resampler +
    compositor, and the same picture as the pass-per-launch kernels make."""
    from oracle import scene as S
    sizes = [(1920, 1080), (8, 2), (12, 6), (260, 38), (516, 26), (64, 130), (20, 6), (1280, 720), (1924, 1082), (3840, 2160),
             (256, 16), (252, 18), (1024, 4), (4, 1024), (640, 362), (332, 250), (960, 540), (8, 8), (2048, 858), (1440, 1080)]
    sizes = [(max(w, 8), h) for w, h in sizes]  # (the block converter takes widths from 8; narrower frames take the pass kernels)
    W, H = 1920, 1080
    root = S.Tiles(children=[S.InputStream(i) for i in range(len(sizes))], background_color=(8, 8, 8, 255))
    layouts = S.scene_layouts(root, W, H, sizes)
    rng = np.random.default_rng(77)
    planes, kinds = [], []
    for i, (w, h) in enumerate(sizes):
        y, u, v = scenes.test_input(i, w, h, noise_seed=900 + i)
        kind = ("nv12", "j420", "420")[i % 3]
        planes.append((y, u, v)); kinds.append(kind)
    del rng

    def frames_of(c):
        out = []
        for (w, h), (y, u, v), kind in zip(sizes, planes, kinds):
            if kind == "nv12":
                out.append(c.frame(hip.FRAME_NV12, w, h, [y, np.stack([u, v], axis=-1)]))
            else:
                out.append(c.frame(hip.FRAME_PLANAR_YUVJ420 if kind == "j420" else hip.FRAME_PLANAR_YUV420, w, h, [y, u, v]))
        return out
    c, g = hip.Context(0), hip.Context(0)
    g.set_fused_kernels(False)
    try:
        got = _render(c, hip, layouts, frames_of(c), W, H)
        ref = _render(g, hip, layouts, frames_of(g), W, H)
        nodes = []
        for (w, h), (y, u, v), kind in zip(sizes, planes, kinds):
            nodes.append(orc.nv12_to_rgba(y, np.stack([u, v], axis=-1), w, h) if kind == "nv12"
                         else orc.planar_yuv_to_rgba(y, u, v, w, h, orc.YUVJ420 if kind == "j420" else orc.YUV420))
        # the node textures themselves, frame by frame (single-frame launches of the same kernel): the oracle's bytes
        for f, node in zip(frames_of(c), nodes):
            assert np.array_equal(c.frame_to_rgba(f).download(), node)
        want, _ = refpipe.render_yuv420(layouts, nodes, W, H, omp=True)
        for a, b, w_, pl in zip(got, ref, want, "YUV"):
            assert refpipe.max_diff(a, w_) <= 1, f"plane {pl}: {refpipe.max_diff(a, w_)} LSB off the oracle"
            assert refpipe.max_diff(a, b) <= 1 and refpipe.exact_fraction(a, b) >= 0.99, f"plane {pl}: fused vs pass-per-launch"
    finally:
        c.close()
        g.close()
