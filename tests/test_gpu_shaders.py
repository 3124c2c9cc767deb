"""Built-in ports of the reference's in-tree WGSL shaders (include/smr.h smr_builtin_shader_id) through the HIP path, against the
oracle's forward rasterisation of the same planes (oracle/smr_oracle.c orc_builtin_shader) and against the reference's own golden
bytes for the gradient shader (integration-tests/src/render_tests/yuv_tests.rs:31-87)."""
import json

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

GRADIENT_RGB_EXPECTED = [71, 0, 0, 255, 120, 0, 0, 255, 152, 0, 0, 255, 177, 0, 0, 255, 198, 0, 0, 255, 216, 0, 0, 255, 233, 0, 0, 255,
                         248, 0, 0, 255] * 2
GRADIENT_YUV_EXPECTED = [89, 0, 0, 255, 100, 5, 3, 255, 160, 0, 0, 255, 165, 2, 0, 255, 204, 0, 0, 255, 207, 1, 0, 255, 239, 0, 0, 255,
                         241, 1, 0, 255] * 2


@pytest.fixture(scope="module")
def hip():
    from smelter_amd import hip as h
    return h


@pytest.fixture(scope="module")
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def _textures(n, w, h, seed=3):
    out = []
    for i in range(n):
        rng = np.random.default_rng(seed + i)
        yy, xx = np.mgrid[0:h, 0:w]
        t = np.zeros((h, w, 4), np.uint8)
        t[..., 0] = (xx * 255 // max(1, w - 1)) ^ (i * 37)
        t[..., 1] = (yy * 255 // max(1, h - 1))
        t[..., 2] = rng.integers(0, 256, (h, w), dtype=np.uint8)
        a = rng.integers(0, 256, (h, w), dtype=np.uint8) if i % 2 else np.full((h, w), 255, np.uint8)
        t[..., 3] = a
        t[..., :3] = (t[..., :3].astype(np.uint16) * a[..., None] // 255).astype(np.uint8)  # premultiplied node textures
        out.append(t)
    return out


def _run(ctx, sid, textures, W, H, params=b"", time_s=0.0):
    srcs = [ctx.surface_from(t) for t in textures]
    dst = ctx.surface(W, H)
    dst.upload(np.full((H, W, 4), 77, np.uint8))  # stale contents must not show through the clear
    ctx.builtin_shader(sid, srcs, dst, params, time_s)
    return dst.download()


def _check(got, ref, what, identical=0.995):
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1, f"{what}: max diff {d.max()} at {np.unravel_index(d.argmax(), d.shape)}"
    assert (d == 0).mean() >= identical, f"{what}: {(d == 0).mean():.4f} identical"


def test_gradient_matches_the_reference_golden_bytes(ctx, hip):
    """yuv_tests.rs yuv_test_gradient: an 8x2 Shader node with no children; the RGBA texture output bit for bit, the 4:2:0
    output after the harness's BT.601 read-back within the reference's own tolerance of 2."""
    got = _run(ctx, hip.SHADER_GRADIENT, [], 8, 2)
    assert got.reshape(-1).tolist() == GRADIENT_RGB_EXPECTED
    y, u, v = ctx.rgba_to_frame(ctx.surface_from(got), hip.FRAME_PLANAR_YUV420).download()
    back = orc.harness_yuv420_to_rgba(y, u, v, 8, 2)
    assert np.abs(back.reshape(-1).astype(int) - np.array(GRADIENT_YUV_EXPECTED)).max() <= 2


def test_gradient_scene_through_the_renderer(ctx, hip):
    from smelter_amd.renderer import Renderer
    r = Renderer(ctx)
    try:
        r.register_shader("example_shader", hip.SHADER_GRADIENT)
        scene = {"type": "shader", "shader_id": "example_shader", "resolution": {"width": 8, "height": 2}}
        r.update_scene("out", 8, 2, json.dumps(scene), output_format=hip.FRAME_RGBA)
        out = r.render(0.0, {})["out"].download()
        assert np.asarray(out[0]).reshape(-1).tolist() == GRADIENT_RGB_EXPECTED
        r.update_scene("out", 8, 2, json.dumps(scene))
        y, u, v = [np.asarray(p) for p in r.render(0.0, {})["out"].download()]
        back = orc.harness_yuv420_to_rgba(y, u, v, 8, 2)
        assert np.abs(back.reshape(-1).astype(int) - np.array(GRADIENT_YUV_EXPECTED)).max() <= 2
    finally:
        r.close()


@pytest.mark.parametrize("n_src", [0, 1, 2, 3])
def test_color_by_texture_count(ctx, hip, n_src):
    tex = _textures(n_src, 32, 18)
    got = _run(ctx, hip.SHADER_COLOR_BY_TEXTURE_COUNT, tex, 64, 36)
    ref = orc.builtin_shader(orc.SHADER_COLOR_BY_TEXTURE_COUNT, tex, 64, 36)
    assert np.array_equal(got, ref)
    assert got[0, 0].tolist() == [[255, 0, 0, 255], [0, 255, 0, 255], [0, 0, 255, 255], [0, 0, 255, 255]][n_src]


@pytest.mark.parametrize("size", [(640, 360), (333, 201)])
def test_red_border(ctx, hip, size):
    W, H = size
    tex = _textures(1, 160, 90)
    got = _run(ctx, hip.SHADER_RED_BORDER, tex, W, H)
    _check(got, orc.builtin_shader(orc.SHADER_RED_BORDER, tex, W, H), "red_border")
    assert got[10, 10].tolist() == [255, 0, 0, 255] and got[H // 2, W // 2].tolist() != [255, 0, 0, 255]


@pytest.mark.parametrize("n_src", [0, 1, 2, 4, 5])
def test_layout_planes(ctx, hip, n_src):
    tex = _textures(n_src, 200, 120)
    got = _run(ctx, hip.SHADER_LAYOUT_PLANES, tex, 640, 360)
    _check(got, orc.builtin_shader(orc.SHADER_LAYOUT_PLANES, tex, 640, 360), f"layout_planes n={n_src}")
    if n_src == 0:
        assert got[100, 100].tolist() == [255, 0, 0, 255]
    if n_src == 2:
        assert not got[300, 100].any()  # the lower half was only cleared


@pytest.mark.parametrize("t", [0.0, 0.7, 1.9, 4.0])
def test_fade_to_ball(ctx, hip, t):
    tex = _textures(1, 320, 180)
    got = _run(ctx, hip.SHADER_FADE_TO_BALL, tex, 640, 360, time_s=t)
    _check(got, orc.builtin_shader(orc.SHADER_FADE_TO_BALL, tex, 640, 360, time=t), f"fade_to_ball t={t}", identical=0.98)


@pytest.mark.parametrize("t", [0.0, 0.4, 1.3])
def test_silly(ctx, hip, t):
    tex = _textures(1, 320, 180)
    got = _run(ctx, hip.SHADER_SILLY, tex, 640, 360, time_s=t)
    ref = orc.builtin_shader(orc.SHADER_SILLY, tex, 640, 360, time=t)
    # sin / cos / atan2 of the device library against libm: a sample position may move by an ulp, noise texels follow
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    assert (d <= 1).mean() >= 0.999 and (d == 0).mean() >= 0.97, f"silly t={t}: {(d <= 1).mean():.5f} / {(d == 0).mean():.4f}"
    none = _run(ctx, hip.SHADER_SILLY, [], 64, 36)
    assert not none.any()


def test_circle_layout_overlapping_planes_blend_in_order(ctx, hip):
    tex = _textures(3, 200, 200)
    circles = [(10, 20, 300, 300, (0.0, 0.0, 1.0, 1.0)), (200, 50, 250, 200, (0.0, 0.25, 0.0, 0.5)), (400, 100, 240, 260, (0.0, 0.0, 0.0, 0.0))]
    params = orc.circle_layout_params(circles)
    got = _run(ctx, hip.SHADER_CIRCLE_LAYOUT, tex, 640, 360, params)
    ref = orc.builtin_shader(orc.SHADER_CIRCLE_LAYOUT, tex, 640, 360, params=params)
    _check(got, ref, "circle_layout", identical=0.99)
    assert not got[5, 5].any() and got[20, 10].tolist() == [0, 0, 255, 255]


def test_circle_layout_needs_its_parameters(ctx, hip):
    tex = _textures(2, 32, 32)
    with pytest.raises(Exception) as e:
        _run(ctx, hip.SHADER_CIRCLE_LAYOUT, tex, 64, 64, params=b"\0" * 32)
    assert "circle_layout" in str(e.value)
    with pytest.raises(Exception):
        _run(ctx, 99, tex, 64, 64)


def test_circle_layout_scene_with_struct_params_through_the_renderer(ctx, hip):
    """ShaderParam::to_bytes over a list of structs (shader/node.rs:95-112), children rendered to node textures first."""
    from smelter_amd.renderer import Renderer
    from tests import scenes
    r = Renderer(ctx)
    try:
        iw, ih, W, H = 320, 180, 640, 360
        planes = [scenes.test_input(i, iw, ih, noise_seed=11 + i) for i in range(2)]
        frames = {f"in{i}": ctx.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(p)) for i, p in enumerate(planes)}
        for k in frames:
            r.register_input(k)
        r.register_shader("circles", hip.SHADER_CIRCLE_LAYOUT)
        circles = [(20, 30, 280, 280, (0.0, 0.0, 1.0, 1.0)), (330, 40, 300, 300, (0.0, 0.0, 0.0, 0.0))]

        def entry(c):
            l, t, w, h, bg = c
            color = {"type": "list", "value": [{"type": "f32", "value": x} for x in bg]}
            return {"type": "struct", "value": [
                {"field_name": "left_px", "type": "u32", "value": l}, {"field_name": "top_px", "type": "u32", "value": t},
                {"field_name": "width_px", "type": "u32", "value": w}, {"field_name": "height_px", "type": "u32", "value": h},
                dict(field_name="background_color", **color)]}
        scene = {"type": "shader", "shader_id": "circles", "resolution": {"width": W, "height": H},
                 "shader_param": {"type": "list", "value": [entry(c) for c in circles]},
                 "children": [{"type": "input_stream", "input_id": "in0"}, {"type": "input_stream", "input_id": "in1"}]}
        r.update_scene("out", W, H, json.dumps(scene), output_format=hip.FRAME_RGBA)
        got = np.asarray(r.render(0.0, frames)["out"].download()[0]).reshape(H, W, 4)
        tex = [orc.planar_yuv_to_rgba(*p, iw, ih) for p in planes]
        ref = orc.builtin_shader(orc.SHADER_CIRCLE_LAYOUT, tex, W, H, params=orc.circle_layout_params(circles))
        # the RGBA output is a clone of the root node's texture (render_loop.rs:81-103)
        _check(got, ref, "circle_layout scene", identical=0.99)
    finally:
        r.close()
