"""smr_renderer_* — the reference's `Renderer` (state.rs:96-252) end to end on the GPU: scene JSON + frame sets in, output
frames out.  Checked against the same work done pass by pass through the lower-level C ABI (bit for bit) and against the
oracle pipeline (<= 1 LSB)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import refpipe, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from smelter_amd import hip as h
    return h


@pytest.fixture(scope="module")
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


@pytest.fixture()
def renderer(ctx):
    from smelter_amd.renderer import Renderer
    r = Renderer(ctx)
    yield r
    r.close()


def _frames(ctx, hip, n, w, h, seed=7):
    planes = [scenes.test_input(i, w, h, noise_seed=seed + i) for i in range(n)]
    return planes, {f"in{i}": ctx.frame(hip.FRAME_PLANAR_YUV420, w, h, list(p)) for i, p in enumerate(planes)}


def _set_labels(r, out_id, nodes):
    from smelter_amd import _ffi
    atlas, glyphs = scenes.label_glyphs("CAM 3 LIVE", 3)
    for n in nodes:
        if n.kind == _ffi.NODE_TEXT:
            r.set_text(out_id, n.index, glyphs, atlas)


def test_cfg3_scene_matches_the_layout_list_path_and_the_oracle(ctx, hip, renderer):
    iw, ih, W, H, n = 480, 270, 960, 540, 8
    planes, frames = _frames(ctx, hip, n, iw, ih)
    for k in frames:
        renderer.register_input(k)
    nodes = renderer.update_scene("out", W, H, scenes.cfg3_scene_json(n))
    _set_labels(renderer, "out", nodes)
    got = renderer.render(0.0, frames)["out"].download()
    # the same scene as a flattened layout list through smr_render_layouts
    layouts, res = scenes.cfg3_scene(iw, ih, W, H, n)
    atlas, glyphs = scenes.label_glyphs("CAM 3 LIVE", 3)
    label = ctx.surface(scenes.LABEL_W, scenes.LABEL_H)
    ctx.blit_glyphs(label, (0.0, 0.0, 0.0, 0.0), glyphs, atlas)
    srcs, k = [], 0
    for r_ in res:
        if r_ == (iw, ih):
            srcs.append(frames[f"in{k}"]); k += 1
        else:
            srcs.append(label)
    out = ctx.frame(hip.FRAME_PLANAR_YUV420, W, H)
    ctx.render_layouts(layouts, srcs, W, H, out=out)
    for a, b in zip(got, out.download()):
        assert (a == b).all()
    # and the oracle's restatement of the reference's pass sequence
    label_host = orc.blit_glyphs(scenes.LABEL_W, scenes.LABEL_H, orc.color_to_shader((0, 0, 0, 0), True), glyphs, atlas, True)  # (the oracle's own label node)
    nodes_o, k = [], 0
    for r_ in res:
        if r_ == (iw, ih):
            nodes_o.append(orc.planar_yuv_to_rgba(*planes[k], iw, ih)); k += 1
        else:
            nodes_o.append(label_host)
    want, _ = refpipe.render_yuv420(layouts, nodes_o, W, H)
    for g, w_ in zip(got, want):
        assert refpipe.max_diff(g, w_) <= 1 and refpipe.exact_fraction(g, w_) >= 0.995


def test_stale_and_unregistered_inputs_are_dropped(ctx, hip, renderer):
    """populate_inputs (render_loop.rs:19-42): no frame, a frame older than stream_fallback_timeout, or an input that was never
    registered all leave the node without a texture — the layout then samples the transparent 1x1 (layout.rs:204-212)."""
    iw, ih, W, H = 320, 180, 640, 360
    _, frames = _frames(ctx, hip, 2, iw, ih)
    renderer.register_input("in0")
    renderer.register_input("in1")
    renderer.update_scene("out", W, H, scenes.cfg2_scene_json(2))
    both = renderer.render(10.0, frames, {"in0": 10.0, "in1": 10.0})["out"].download()
    only0 = renderer.render(10.0, {"in0": frames["in0"]}, {"in0": 10.0})["out"].download()
    stale = renderer.render(10.0, frames, {"in0": 10.0, "in1": 9.4})["out"].download()      # 0.6 s old > 0.5 s
    fresh = renderer.render(10.0, frames, {"in0": 10.0, "in1": 9.6})["out"].download()      # 0.4 s old
    renderer.unregister_input("in1")
    unreg = renderer.render(10.0, frames, {"in0": 10.0, "in1": 10.0})["out"].download()
    assert not (both[0] == only0[0]).all()
    for a, b, c, d in zip(only0, stale, unreg, both):
        assert (a == b).all() and (a == c).all()
    for a, b in zip(fresh, both):
        assert (a == b).all()


def test_shader_node_with_nested_layout_and_image(ctx, hip, renderer):
    """Render graph with every node kind: root View { Shader(blur){ View{ InputStream } }, Image }."""
    iw, ih, W, H = 320, 180, 640, 360
    _, frames = _frames(ctx, hip, 1, iw, ih)
    rng = np.random.default_rng(3)
    logo = rng.integers(0, 256, (40, 60, 4), dtype=np.uint8)
    renderer.register_input("in0")
    renderer.register_image("logo", logo)
    renderer.register_shader("soften")
    scene = {"type": "view", "background_color": "#204060FF", "children": [
        {"type": "shader", "shader_id": "soften", "resolution": {"width": 320, "height": 180}, "shader_param": {"type": "f32", "value": 2.5},
         "children": [{"type": "view", "width": 320, "height": 180, "background_color": "#FF0000FF",
                       "children": [{"type": "rescaler", "child": {"type": "input_stream", "input_id": "in0"}, "border_radius": 20}]}]},
        {"type": "image", "image_id": "logo"},
    ]}
    nodes = renderer.update_scene("out", W, H, scene)
    from smelter_amd import _ffi
    assert [n.kind for n in nodes] == [_ffi.NODE_LAYOUT, _ffi.NODE_SHADER, _ffi.NODE_LAYOUT, _ffi.NODE_INPUT_STREAM, _ffi.NODE_IMAGE]
    got = renderer.render(0.0, frames)["out"].download()
    # the same graph by hand through the lower-level entry points
    from smelter_amd.scene import Scene
    sc = Scene()
    sc.register_image("logo", 60, 40)
    sc.update(scene, W, H)
    inner_layouts = sc.layouts(2, 0, [(iw, ih)])
    inner = ctx.surface(320, 180)
    ctx.render_layouts(inner_layouts, [frames["in0"]], 320, 180, out_rgba=inner)
    blurred = ctx.gaussian_blur(inner, 2.5)
    raw = ctx.surface_from(logo)
    image = ctx.add_premultiplied_alpha(raw)
    root_layouts = sc.layouts(0, 0, [(320, 180), (60, 40)])
    out = ctx.frame(hip.FRAME_PLANAR_YUV420, W, H)
    ctx.render_layouts(root_layouts, [blurred, image], W, H, out=out)
    for a, b in zip(got, out.download()):
        assert (a == b).all()
    assert got[0].std() > 5  # not a blank frame


@pytest.mark.parametrize("opaque", [True, False])
def test_an_opaque_image_is_composited_as_an_opaque_layer_to_the_same_bytes(ctx, hip, renderer, opaque):
    """The renderer knows an image whose every alpha byte is 255 to be opaque and names it SMR_SOURCE_OPAQUE_SURFACE (copy / select tiles where it
    lies 1:1, start values where something blends over it); the same list by hand with the image as a plain surface blends it everywhere: the
    same bytes.  With one translucent pixel in the image both go the blending way."""
    W, H = 640, 360
    rng = np.random.default_rng(11)
    pic = rng.integers(0, 256, (128, 256, 4), dtype=np.uint8)
    pic[..., 3] = 255
    if not opaque:
        pic[40, 100, 3] = 90
    renderer.register_image("pic", pic)
    scene = {"type": "view", "background_color": "#203040FF", "children": [
        {"type": "view", "top": 40, "left": 64, "width": 256, "height": 128, "children": [{"type": "image", "image_id": "pic"}]},
        {"type": "view", "top": 100, "left": 200, "width": 300, "height": 120, "background_color": "#FFFFFF60", "border_radius": 12},
    ]}
    renderer.update_scene("out", W, H, scene)
    got = renderer.render(0.0, {})["out"].download()
    from smelter_amd.scene import Scene
    sc = Scene()
    sc.register_image("pic", 256, 128)
    sc.update(scene, W, H)
    image = ctx.add_premultiplied_alpha(ctx.surface_from(pic))
    out = ctx.frame(hip.FRAME_PLANAR_YUV420, W, H)
    ctx.render_layouts(sc.layouts(0, 0, [(256, 128)]), [image], W, H, out=out)
    for a, b in zip(got, out.download()):
        assert (a == b).all()
    assert got[0].std() > 5


def test_transition_animates_between_updates(ctx, hip, renderer):
    iw, ih, W, H = 320, 180, 640, 360
    _, frames = _frames(ctx, hip, 1, iw, ih)
    renderer.register_input("in0")

    def scene(width, tr=None):
        r = {"type": "rescaler", "id": "pip", "width": width, "height": width * 9 / 16, "top": 20, "left": 20,
             "child": {"type": "input_stream", "input_id": "in0"}}
        if tr:
            r["transition"] = tr
        return {"type": "view", "background_color": "#000000FF", "children": [r]}

    renderer.update_scene("out", W, H, scene(160.0))
    a = renderer.render(1.0, frames, {"in0": 1.0})["out"].download()
    renderer.update_scene("out", W, H, scene(480.0, {"duration_ms": 1000}))
    mid = renderer.render(1.5, frames, {"in0": 1.5})["out"].download()
    end = renderer.render(2.5, frames, {"in0": 2.5})["out"].download()
    # reference points: static scenes of the interpolated sizes in a fresh renderer
    from smelter_amd.renderer import Renderer
    fresh = Renderer(ctx)
    fresh.register_input("in0")
    fresh.update_scene("o", W, H, scene(320.0))
    want_mid = fresh.render(0.0, frames)["o"].download()
    fresh.update_scene("o", W, H, scene(480.0))
    want_end = fresh.render(0.0, frames)["o"].download()
    fresh.close()
    for g, w_ in zip(mid, want_mid):
        assert (g == w_).all()
    for g, w_ in zip(end, want_end):
        assert (g == w_).all()
    assert not (a[0] == mid[0]).all()


@pytest.mark.parametrize("size", ["quarter", "full"])
def test_animated_grid_with_blur_layer_matches_the_oracle_mid_transition(ctx, hip, renderer, size):
    """BASELINE configs[4] — at a quarter of its size and at FULL size (16 x 1920x1080 -> 3840x2160, 960x540 blur layer: what bench.py
    --config 4 runs): 16 inputs in a Tiles grid that is 40 % through a transition (tiles at fractional positions: bilinear-sampled, not
    copy tiles) plus a layer through the gaussian-blur shader.  The frame is compared with the oracle's pass sequence (OpenMP build at
    full size) run on the layout lists the scene engine produces for the same pts: <= 1 LSB."""
    from smelter_amd import synth
    from smelter_amd.scene import Scene
    iw, ih, W, H, n, lw, lh = (480, 270, 960, 540, 16, 240, 136) if size == "quarter" else (1920, 1080, 3840, 2160, 16, 960, 540)
    omp = size == "full"
    planes, frames = _frames(ctx, hip, n, iw, ih)
    frames = {f"input_{i}": frames[f"in{i}"] for i in range(n)}
    for k in frames:
        renderer.register_input(k)
    renderer.register_shader("soften")
    sc = Scene()
    res = [(iw, ih)] * n + [(lw, lh)]
    renderer.update_scene("out", W, H, synth.animated_grid_scene(n, 0, lw, lh))
    sc.update(synth.animated_grid_scene(n, 0, lw, lh), W, H)
    first = renderer.render(0.0, frames, {k: 0.0 for k in frames})["out"].download()
    sc.layouts(0, 0, res)                               # the same first frame at pts 0 for the bare engine
    renderer.update_scene("out", W, H, synth.animated_grid_scene(n, 3, lw, lh))
    sc.update(synth.animated_grid_scene(n, 3, lw, lh), W, H)
    t = 0.2                                             # 40 % of the 500 ms transition
    got = renderer.render(t, frames, {k: t for k in frames})["out"].download()
    assert not (got[0] == first[0]).all()
    # the oracle on the engine's layout lists: inner node (View{Rescaler{input_0}}) -> blur -> root
    nodes = sc.nodes()
    root_kids = list(nodes[0].children)
    shader = root_kids[-1]
    inner = list(nodes[shader].children)[0]
    rgba = [orc.planar_yuv_to_rgba(*p, iw, ih, omp=omp) for p in planes]
    inner_px = refpipe.layout_node_render(sc.layouts(inner, int(t * 1e9), [(iw, ih)]), [rgba[0]], lw, lh, omp=omp)
    layer = orc.gaussian_blur(inner_px, 3.0)
    root_layouts = sc.layouts(0, int(t * 1e9), res)
    moving = [L for L in root_layouts if L.type == 0 and (L.left != round(L.left) or L.top != round(L.top))]
    assert len(moving) >= 8                              # really mid-flight
    order = [(i + 3) % n for i in range(n)]              # child k of the grid shows input order[k]
    want, _ = refpipe.render_yuv420(root_layouts, [rgba[i] for i in order] + [layer], W, H, omp=omp)
    for g, w_ in zip(got, want):
        assert refpipe.max_diff(g, w_) <= 1 and refpipe.exact_fraction(g, w_) >= 0.99
    for f in frames.values():
        f.destroy()


def test_a_long_transition_reclassifies_every_frame_and_stays_byte_equal_to_a_fresh_renderer(ctx, hip, renderer):
    """150 frames of a grid in motion (quarter-size configs[4]): every frame has a new root layout list — a new parameter pack, a new
    classification, a new band list — while the blur layer's node repeats its own.  Frames 70 and 149 — after the ring of list counters
    (64) wrapped twice and the ring of parameter packs came round past the slot the static node keeps reusing — must equal, byte for
    byte, what a fresh renderer that jumps straight to the same presentation time renders."""
    from smelter_amd import synth
    from smelter_amd.renderer import Renderer
    iw, ih, W, H, n, lw, lh = 480, 270, 960, 540, 16, 240, 136
    _, frames = _frames(ctx, hip, n, iw, ih)
    frames = {f"input_{i}": frames[f"in{i}"] for i in range(n)}
    fresh = Renderer(ctx)
    for r in (renderer, fresh):
        for k in frames:
            r.register_input(k)
        r.register_shader("soften")
        r.update_scene("out", W, H, synth.animated_grid_scene(n, 0, lw, lh))
        r.render(0.0, frames, {k: 0.0 for k in frames})
        r.update_scene("out", W, H, synth.animated_grid_scene(n, 5, lw, lh))
    ts = [0.003 * (k + 1) for k in range(150)]           # 3 ms apart: all inside the 500 ms transition
    kept = {}
    for k, t in enumerate(ts):
        out = renderer.render(t, frames, {q: t for q in frames})["out"]
        if k in (0, 70, 149):
            kept[k] = out.download()
    assert not (kept[70][0] == kept[149][0]).all()       # really moving
    for k in (0, 70, 149):
        want = fresh.render(ts[k], frames, {q: ts[k] for q in frames})["out"].download()
        for g, w_ in zip(kept[k], want):
            assert (g == w_).all(), k
    fresh.close()
    for f in frames.values():
        f.destroy()


def test_many_nested_nodes_at_rest_under_a_root_in_motion(ctx, hip, renderer):
    """Ten shader nodes, each over its own nested layout node that never changes, under a root whose rescaler is in a transition: every frame
    stages eleven parameter packs of which ten repeat — more resident packs than the ring's eight slots (it grows instead of recycling one of
    them and waiting for the stream every frame).  The last of 40 frames equals a fresh renderer's."""
    from smelter_amd.renderer import Renderer
    iw, ih, W, H = 320, 180, 640, 360
    _, frames = _frames(ctx, hip, 1, iw, ih)

    def scene(width, tr=None):
        kids = [{"type": "view", "top": 8 + 34 * (k // 5), "left": 8 + 70 * (k % 5), "width": 64, "height": 30,
                 "children": [{"type": "shader", "shader_id": "soften", "resolution": {"width": 64, "height": 30}, "shader_param": {"type": "f32", "value": 1.0 + 0.2 * k},
                               "children": [{"type": "view", "width": 64, "height": 30, "background_color": "#%02X4080FF" % (20 * k),
                                             "children": [{"type": "rescaler", "child": {"type": "input_stream", "input_id": "in0"}}]}]}]}
                for k in range(10)]
        pip = {"type": "rescaler", "id": "pip", "width": width, "height": width * 9 / 16, "top": 100, "left": 40, "child": {"type": "input_stream", "input_id": "in0"}}
        if tr:
            pip["transition"] = tr
        return {"type": "view", "background_color": "#101010FF", "children": kids + [pip]}

    fresh = Renderer(ctx)
    for r in (renderer, fresh):
        r.register_input("in0")
        r.register_shader("soften")
        r.update_scene("out", W, H, scene(200.0))
        r.render(0.0, frames, {"in0": 0.0})
        r.update_scene("out", W, H, scene(420.0, {"duration_ms": 1000}))
    ts = [0.02 * (k + 1) for k in range(40)]
    for t in ts:
        got = renderer.render(t, frames, {"in0": t})["out"].download()
    fresh.render(ts[0], frames, {"in0": ts[0]})
    want = fresh.render(ts[-1], frames, {"in0": ts[-1]})["out"].download()
    fresh.close()
    for g, w_ in zip(got, want):
        assert (g == w_).all()
    assert got[0].std() > 5


def test_configs0_two_720p_packed_rgba_inputs_static_view_cpu_optimized_is_bit_exact(hip):
    """BASELINE configs[0] — the integration-tests plumbing scene (pixel_input_format_tests.rs:30-150 at 720p): two 1280x720 inputs handed
    over as FrameData::Bgra / FrameData::Argb bytes, a static View, RenderingMode::CpuOptimized, RGBA output.  Integer work end to end
    (byte swizzle, 1:1 texel-aligned blit over a transparent background, RGBA texture out): every byte equal to the oracle's pass sequence,
    and the visible part of either input equal to the reference's swizzle of its bytes ([3, 2, 1, 4] / [4, 1, 2, 3] for [1, 2, 3, 4])."""
    from oracle import scene as S
    from smelter_amd.renderer import Renderer
    from tests import scene_json
    w, h, W, H = 1280, 720, 1280, 720
    rng = np.random.default_rng(720)
    data = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(2)]
    data[0][: h // 2, :, 3] = 255  # (BGRA: alpha is byte 3 — the upper half opaque, the rest arbitrary, taken as premultiplied: bgra_to_rgba.wgsl:26-27)
    data[1][: h // 2, :, 2] = 255  # (ARGB: the reference's swizzle is [x3, x0, x1, x2] — byte 2 ends up as alpha, pixel_input_format_tests.rs:133-135)
    scene = {"type": "view", "children": [
        {"type": "view", "width": 640.0, "children": [{"type": "input_stream", "input_id": "bgra"}]},
        {"type": "view", "width": 640.0, "children": [{"type": "input_stream", "input_id": "argb"}]}]}
    c = hip.Context(0, mode=hip.MODE_CPU_OPTIMIZED)
    r = Renderer(c)
    try:
        frames = {"bgra": c.frame(hip.FRAME_BGRA, w, h, [data[0]]), "argb": c.frame(hip.FRAME_ARGB, w, h, [data[1]])}
        for k in frames:
            r.register_input(k)
        r.update_scene("out", W, H, scene, output_format=hip.FRAME_RGBA)
        got = r.render(0.0, frames)["out"].download()
        got = got[0] if isinstance(got, (list, tuple)) else got
        got = np.asarray(got).reshape(H, W, 4)
        nodes = [orc.swizzle_to_rgba(data[0], w, h, 0), orc.swizzle_to_rgba(data[1], w, h, 1)]
        assert np.array_equal(nodes[0][0, 0], data[0][0, 0][[2, 1, 0, 3]]) and np.array_equal(nodes[1][0, 0], data[1][0, 0][[3, 0, 1, 2]])
        root, ids = scene_json.to_oracle(scene)
        assert ids == ["bgra", "argb"]
        layouts = S.scene_layouts(root, W, H, [(w, h), (w, h)], srgb=False)
        want = refpipe.layout_node_render(layouts, nodes, W, H, srgb=False, omp=True)
        assert np.array_equal(got, want), int((got != want).sum())
        # ... and the blit is the swizzled bytes themselves where a texel is opaque (premultiplied OVER a transparent background adds nothing)
        for half, node in ((slice(0, 640), nodes[0]), (slice(640, 1280), nodes[1])):
            vis = node[:, 0:640]
            opaque = vis[..., 3] == 255
            assert np.array_equal(got[:, half][opaque], vis[opaque])
    finally:
        r.close()
        c.close()


def test_frames_in_flight_share_one_scene_state(ctx, hip, renderer):
    """A renderer with three lanes (smr_renderer_add_lane) against the plain one-lane renderer on the same call sequence —
    scene updates with transitions, a text run, a shader node: every frame comes out bit-identical, whichever lane rendered it,
    and a frame stays valid while the other lanes render."""
    from smelter_amd import synth
    from smelter_amd.renderer import Renderer
    iw, ih, W, H, n, lw, lh = 320, 180, 640, 360, 6, 160, 90
    _, frames = _frames(ctx, hip, n, iw, ih)
    frames = {f"input_{i}": frames[f"in{i}"] for i in range(n)}
    extra = [hip.Context(0), hip.Context(0)]
    piped = Renderer(ctx, lanes=extra)
    for r in (renderer, piped):
        for k in frames:
            r.register_input(k)
        r.register_shader("soften")
    held = []
    for step in range(14):
        if step == 10:
            # a scene with text runs: the glyph surfaces are rendered once (on the first lane's stream) and read by every lane
            for r in (renderer, piped):
                for i in range(n):
                    r.register_input(f"in{i}")
                _set_labels(r, "out", r.update_scene("out", W, H, scenes.cfg3_scene_json(n)))
            frames = {**frames, **{f"in{i}": frames[f"input_{i}"] for i in range(n)}}
        elif step % 5 == 0:
            for r in (renderer, piped):
                r.update_scene("out", W, H, synth.animated_grid_scene(n, step // 5, lw, lh, transition_ms=100))
        t = step / 60
        pts = {k: t for k in frames}
        want = renderer.render(t, frames, pts)["out"].download()
        got_frame = piped.render(t, frames, pts)["out"]
        held.append((got_frame, want))
        if len(held) == 3:  # read a frame back two render calls later: its lane has not been reused yet
            f, w_ = held.pop(0)
            for a, b in zip(f.download(), w_):
                assert (a == b).all(), step
    piped.sync()
    for f, w_ in held:
        for a, b in zip(f.download(), w_):
            assert (a == b).all()
    piped.close()
    for c in extra:
        c.close()


def test_resample_targets_are_reused_across_sizes(ctx, hip, renderer):
    """The resample target of an animated rescaler changes size on every frame; the scratch surface behind it is re-described
    in place while it fits its allocation and regrown otherwise.  Shrinking, growing past the first allocation and coming back
    all give the frame a fresh renderer gives for the same static scene."""
    from smelter_amd.renderer import Renderer
    iw, ih, W, H = 320, 180, 640, 360
    _, frames = _frames(ctx, hip, 1, iw, ih)
    renderer.register_input("in0")

    def scene(width):
        return {"type": "view", "background_color": "#102030FF", "children": [
            {"type": "rescaler", "width": width, "height": round(width * 9 / 16), "top": 8, "left": 8,
             "child": {"type": "input_stream", "input_id": "in0"}}]}

    for width in (400, 250, 96, 97, 401, 520, 600, 130, 600, 17):
        renderer.update_scene("out", W, H, scene(width))
        got = renderer.render(0.0, frames)["out"].download()
        ctx2 = hip.Context(0)                  # scratch surfaces are cached per context: a new one has nothing cached
        _, frames2 = _frames(ctx2, hip, 1, iw, ih)
        fresh = Renderer(ctx2)
        fresh.register_input("in0")
        fresh.update_scene("o", W, H, scene(width))
        want = fresh.render(0.0, frames2)["o"].download()
        fresh.close()
        ctx2.close()
        for g, w_ in zip(got, want):
            assert (g == w_).all(), width


def test_non_layout_root_and_empty_output(ctx, hip, renderer):
    iw, ih = 320, 180
    planes, frames = _frames(ctx, hip, 1, iw, ih)
    renderer.register_input("in0")
    renderer.update_scene("direct", iw, ih, {"type": "input_stream", "input_id": "in0"})
    outs = renderer.render(0.0, frames)
    node = ctx.frame_to_rgba(frames["in0"])
    want = ctx.rgba_to_frame(node, hip.FRAME_PLANAR_YUV420).download()
    for a, b in zip(outs["direct"].download(), want):
        assert (a == b).all()
    # no frame for the input: the root node is empty -> black (render_loop.rs:127-139)
    black = renderer.render(0.0, {})["direct"].download()
    assert (black[0] == 16).all() and (black[1] == 128).all() and (black[2] == 128).all()
    # two outputs at once, NV12 for the second
    renderer.update_scene("second", 640, 360, scenes.cfg2_scene_json(1), output_format=hip.FRAME_NV12)
    outs = renderer.render(0.0, frames)
    assert set(outs) == {"direct", "second"}
    y, uv = outs["second"].download()
    assert y.shape == (360, 640) and uv.shape == (180, 320, 2)
    renderer.unregister_output("second")
    assert set(renderer.render(0.0, frames)) == {"direct"}


def test_cpu_optimized_mode_renderer(hip):
    """RenderingMode::CpuOptimized (types.rs:8-18): plain unorm node textures, no resampler (bilinear layout sampling only),
    colours converted without the sRGB decode — end to end through the renderer, against the oracle in the same mode."""
    from smelter_amd.renderer import Renderer
    c = hip.Context(0, mode=hip.MODE_CPU_OPTIMIZED)
    r = Renderer(c)
    iw, ih, W, H, n = 320, 180, 640, 360, 4
    planes, frames = _frames(c, hip, n, iw, ih)
    for k in frames:
        r.register_input(k)
    r.update_scene("out", W, H, scenes.cfg2_scene_json(n))
    got = r.render(0.0, frames)["out"].download()
    from oracle import scene as S
    root = S.Tiles(children=[S.InputStream(i) for i in range(n)], background_color=(0, 0, 0, 255))
    layouts = S.scene_layouts(root, W, H, [(iw, ih)] * n, srgb=False)
    nodes = [orc.planar_yuv_to_rgba(*planes[k], iw, ih) for k in range(n)]
    want, _ = refpipe.render_yuv420(layouts, nodes, W, H, srgb=False)
    for g, w_ in zip(got, want):
        assert refpipe.max_diff(g, w_) <= 1 and refpipe.exact_fraction(g, w_) >= 0.99
    r.close()
    c.close()


def test_pinned_async_transfers_round_trip(ctx, hip):
    """smr_host_alloc + smr_frame_upload_async / _download_async: stream-ordered, byte-exact, every plane shape."""
    rng = np.random.default_rng(5)
    for fmt, w, h in [(hip.FRAME_PLANAR_YUV420, 322, 182), (hip.FRAME_NV12, 640, 360), (hip.FRAME_PLANAR_YUV444, 65, 33)]:
        f = ctx.frame(fmt, w, h)
        src, dst = f.pinned_planes(), f.pinned_planes()
        for p in src:
            p[...] = rng.integers(0, 256, p.shape, dtype=np.uint8)
        f.upload_async(src)
        f.download_async(dst)
        ctx.sync()
        for a, b in zip(src, dst):
            assert (a == b).all()
        for a, b in zip(f.download(), src):
            assert (a == b).all()


def test_scene_errors_surface_through_the_renderer(renderer):
    from smelter_amd.scene import SceneError
    with pytest.raises(SceneError) as e:
        renderer.update_scene("out", 64, 64, {"type": "view", "top": 1, "bottom": 2, "left": 0})
    assert "mutually exclusive" in str(e.value)
    with pytest.raises(SceneError) as e:
        renderer.update_scene("out", 64, 64, {"type": "image", "image_id": "missing"})
    assert "does not exist" in str(e.value)


def test_unknown_shader_is_rejected_and_the_previous_scene_keeps_rendering(ctx, hip, renderer):
    """ShaderComponent::stateful_component (shader_component.rs:44-52): an unregistered shader_id is SceneError::ShaderNotFound at
    update time — update_scene fails, the active scene and its surfaces are untouched, later frames still come out."""
    from smelter_amd.scene import SceneError
    iw, ih, W, H = 160, 90, 320, 180
    _, frames = _frames(ctx, hip, 1, iw, ih)
    renderer.register_input("in0")
    good = {"type": "view", "background_color": "#203040FF", "children": [{"type": "rescaler", "child": {"type": "input_stream", "input_id": "in0"}}]}
    renderer.update_scene("out", W, H, good)
    before = renderer.render(0.0, frames)["out"].download()
    bad = {"type": "view", "children": [{"type": "shader", "shader_id": "nope", "resolution": {"width": 64, "height": 64},
                                         "children": [{"type": "input_stream", "input_id": "in0"}]}]}
    with pytest.raises(SceneError) as e:
        renderer.update_scene("out", W, H, bad)
    assert 'Shader "nope" does not exist' in str(e.value)
    after = renderer.render(1.0 / 60, frames)["out"].download()
    for a, b in zip(before, after):
        assert (a == b).all()
    # a first update that fails leaves no output behind
    with pytest.raises(SceneError):
        renderer.update_scene("other", W, H, bad)
    assert "other" not in renderer.render(2.0 / 60, frames)


def test_frames_with_mismatched_planes_are_rejected(ctx, hip):
    """smr_frame is a caller-filled struct (planes may be wrapped memory): a plane whose size or pixel format does not fit the
    frame's format must be SMR_ERR_INVALID at the entry point, not an out-of-bounds access in a kernel."""
    f = ctx.frame(hip.FRAME_PLANAR_YUV420, 64, 36)
    small = ctx.surface(8, 8, hip.PX_R8)
    node = ctx.surface(64, 36)
    out = ctx.frame(hip.FRAME_PLANAR_YUV420, 64, 36)
    keep = f.c.planes[1]
    f.c.planes[1] = small.handle
    try:
        with pytest.raises(hip.SmrError) as e:
            ctx.frame_to_rgba(f, node)
        assert "plane 1" in str(e.value)
        with pytest.raises(hip.SmrError):
            ctx.ingest_resample(f, (0.0, 0.0, 64.0, 36.0), ctx.surface(32, 18))
        with pytest.raises(hip.SmrError):
            ctx.render_layouts([], [f], 64, 36, out=out)
        with pytest.raises(hip.SmrError):
            ctx.render_layouts([], [], 64, 36, out=f)
    finally:
        f.c.planes[1] = keep
    ctx.frame_to_rgba(f, node)  # intact again


def test_c_example_runs(tmp_path):
    """examples/render_scene.c: the renderer driven from plain C."""
    import subprocess
    from tests.test_abi import _build_c_example
    exe = _build_c_example(tmp_path)
    out = tmp_path / "out.yuv"
    p = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "rendered 1280x720 from 4 inputs" in p.stdout
    if os.path.isdir("/usr/share/fonts/truetype"):  # the labels: laid out, rasterised and drawn by the library itself — no Python, no caller-side shaper
        assert "4 text nodes drawn" in p.stdout and "R\u00e9gie" in p.stdout
    data = np.fromfile(out, np.uint8)
    assert data.size == 1280 * 720 * 3 // 2 and data[: 1280 * 720].std() > 10


def test_fitted_text_is_sized_by_the_shaper_and_drawn(ctx, hip, renderer):
    """TextDimensions::Fitted end to end: the node is sized by get_text_resolution from the caller's shaper (tests/text_twin.py over a
    system TrueType font), the shaper's glyph run is blitted by smr_blit_glyphs, and the output equals the oracle's blit."""
    import json

    from smelter_amd import _ffi
    from tests import text_twin as T
    try:
        book = T.FontBook.system()
    except FileNotFoundError:
        pytest.skip("no TrueType fonts on this machine")
    sh = T.Shaper(book)
    renderer.set_text_measurer(sh.measurer)
    W, H, fs, lh = 640, 360, 34.0, 40.0
    txt = "Fitted text\nsized by the shaper"
    scene = {"type": "view", "background_color": "#00000000",
             "children": [{"type": "text", "text": txt, "font_size": fs, "line_height": lh, "align": "center", "color": "#FFCC33FF"}]}
    nodes = renderer.update_scene("out", W, H, json.dumps(scene), output_format=hip.FRAME_RGBA)
    tn = [n for n in nodes if n.kind == _ffi.NODE_TEXT][0]
    font = book.match("Verdana")
    assert (tn.width, tn.height) == T.text_resolution(T.layout(font, txt, fs), fs, lh)
    color = orc.color_to_shader((0xFF, 0xCC, 0x33, 0xFF), True)
    glyphs, atlas = sh.rasterise(txt, tn.width, tn.height, fs, lh, align="Center", color=color)
    renderer.set_text("out", tn.index, glyphs, atlas)
    got = np.asarray(renderer.render(0.0, {})["out"].download()[0]).reshape(H, W, 4)
    want = orc.blit_glyphs(tn.width, tn.height, (0.0, 0.0, 0.0, 0.0), glyphs, atlas, True)
    assert np.abs(got[:tn.height, :tn.width].astype(int) - want.astype(int)).max() <= 1
    assert (got[:tn.height, :tn.width] == want).mean() > 0.999
    assert not got[tn.height:].any() and not got[:, tn.width:].any()
    assert want[..., 3].max() == 255 and (want[..., 3] > 0).mean() > 0.05  # the text is really there


def test_renderer_with_a_font_book_draws_its_text_nodes_itself(ctx, hip, renderer):
    """smr_renderer_set_fontbook = the reference's TextRendererCtx: update_scene measures fitted Text nodes with the book (get_text_resolution)
    and lays out, rasterises and draws EVERY Text node — colour from the component, background through convert_to_shader_color — once per
    update (text_renderer.rs:72-167, 282-368).  The node's pixels equal the oracle's blit of the same run (the run itself is held byte for
    byte to the Python twin in tests/test_text_capi.py); a second update re-draws; detaching the book restores the caller-supplied path."""
    import json
    import math

    from smelter_amd import _ffi
    from tests import text_twin as T
    try:
        book = T.NativeFontBook.system()
    except FileNotFoundError:
        pytest.skip("no TrueType fonts on this machine")
    renderer.set_fontbook(book)
    W, H, fs, lh = 640, 360, 34.0, 40.0
    txt = "Fitted text\nsized and drawn in C++"

    def scene(t, colour, bg):
        return {"type": "view", "background_color": "#00000000",
                "children": [{"type": "text", "text": t, "font_size": fs, "line_height": lh, "align": "right", "color": colour, "background_color": bg}]}
    for t, colour, bg, rgba, bg_rgba in [(txt, "#FFCC33FF", "#00000000", (0xFF, 0xCC, 0x33, 0xFF), (0, 0, 0, 0)),
                                         ("second update", "#40FF80C0", "#20304080", (0x40, 0xFF, 0x80, 0xC0), (0x20, 0x30, 0x40, 0x80))]:
        nodes = renderer.update_scene("out", W, H, json.dumps(scene(t, colour, bg)), output_format=hip.FRAME_RGBA)
        tn = [n for n in nodes if n.kind == _ffi.NODE_TEXT][0]
        widest, lines = book.measure(t, fs)
        assert (tn.width, tn.height) == (math.ceil(widest), int(lines * math.ceil(lh) + fs / 5.0))
        got = np.asarray(renderer.render(0.0, {})["out"].download()[0]).reshape(H, W, 4)
        glyphs, atlas = book.rasterise(t, tn.width, tn.height, fs, lh, align="Right", color=tuple(c / 255.0 for c in rgba))
        want = orc.blit_glyphs(tn.width, tn.height, orc.color_to_shader(bg_rgba, True), glyphs, atlas, True)
        # the node sits at the view's top-left at 1:1: the layout pass copies it (premultiplied OVER a transparent target)
        assert np.abs(got[:tn.height, :tn.width].astype(int) - want.astype(int)).max() <= 1
        assert (got[:tn.height, :tn.width] == want).mean() > 0.999
        assert not got[tn.height:].any() and not got[:, tn.width:].any()
        assert (want[..., 3] > bg_rgba[3]).mean() > 0.05  # the text is really there
    renderer.set_fontbook(None)
    with pytest.raises(Exception):  # no shaper any more: a fitted Text node is refused
        renderer.update_scene("out", W, H, json.dumps(scene(txt, "#FFFFFFFF", "#00000000")), output_format=hip.FRAME_RGBA)
    book.close()


def test_an_update_whose_text_cannot_be_drawn_leaves_the_previous_scene_active(ctx, hip, renderer):
    """smr_renderer_update_scene drew its Text nodes after the new scene had been committed, so an update that failed there
    (an empty font book) reported an error with the new scene active.  The reference fails an update as a whole (state.rs:177-189): what the
    scene itself can get wrong is now found before the commit (smr_fontbook_measure on every Text node of the converted tree) — the previous
    scene keeps rendering, and a failed FIRST update leaves no output behind."""
    import json

    from smelter_amd import fontbook
    empty = fontbook.NativeFontBook()  # no font in it
    renderer.set_fontbook(empty)
    W, H = 64, 32
    text_scene = {"type": "view", "children": [{"type": "text", "text": "x", "font_size": 12.0, "width": 40.0, "height": 16.0}]}
    with pytest.raises(Exception, match="font book is empty"):
        renderer.update_scene("out", W, H, json.dumps(text_scene), output_format=hip.FRAME_RGBA)
    assert "out" not in renderer.render(0.0, {})  # the failed first update left no output
    red = {"type": "view", "background_color": "#FF0000FF"}
    renderer.update_scene("out", W, H, json.dumps(red), output_format=hip.FRAME_RGBA)
    before = np.asarray(renderer.render(0.0, {})["out"].download()[0]).copy()
    with pytest.raises(Exception, match="font book is empty"):
        renderer.update_scene("out", W, H, json.dumps(text_scene), output_format=hip.FRAME_RGBA)
    after = np.asarray(renderer.render(1.0, {})["out"].download()[0])
    assert before.reshape(H, W, 4)[0, 0].tolist() == [255, 0, 0, 255] and np.array_equal(before, after)  # still the red view
    renderer.set_fontbook(None)
    empty.close()
