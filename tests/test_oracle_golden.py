"""Pins the CPU oracle against every known-answer vector the reference keeps in-tree.

Sources (all under /root/reference, values copied as data — they are test vectors):
  integration-tests/src/render_tests/yuv_tests.rs:30-131        (+-2 per byte, :21)
  integration-tests/src/render_tests/pixel_input_format_tests.rs:30-150   (exact, :21)
  smelter-render/src/transformations/layout/resampler.rs:402-468 (pass planning)
"""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle.oracle import Layout, Mask


def _close(actual, expected, tol):
    a = np.asarray(actual, np.int32).ravel()
    e = np.asarray(expected, np.int32).ravel()
    assert a.shape == e.shape
    assert np.abs(a - e).max() <= tol, f"actual={a.tolist()} expected={e.tolist()}"


# ---- yuv_tests.rs -----------------------------------------------------------------
UNIFORM_YUV_EXPECTED = [49, 0, 0, 255] * 16   # yuv_tests.rs:117-123
UNIFORM_RGB_EXPECTED = [50, 0, 0, 255] * 16   # yuv_tests.rs:126-130
GRADIENT_YUV_EXPECTED = [89, 0, 0, 255, 100, 5, 3, 255, 160, 0, 0, 255, 165, 2, 0, 255,
                         204, 0, 0, 255, 207, 1, 0, 255, 239, 0, 0, 255, 241, 1, 0, 255] * 2  # :69-70
GRADIENT_RGB_EXPECTED = [71, 0, 0, 255, 120, 0, 0, 255, 152, 0, 0, 255, 177, 0, 0, 255,
                         198, 0, 0, 255, 216, 0, 0, 255, 233, 0, 0, 255, 248, 0, 0, 255] * 2  # :76-77


def _uniform_scene():
    # View{background RGBAColor(50,0,0,255)} at 8x2, GpuOptimized: one colour layout.
    col = orc.color_to_shader((50, 0, 0, 255), srgb=True)
    return orc.apply_layouts(8, 2, [Layout(top=0, left=0, width=8, height=2, type=1, color=col)], [], srgb=True)


def test_yuv_uniform_color_rgba_output():
    _close(_uniform_scene(), UNIFORM_RGB_EXPECTED, 2)
    # the reference value is in fact hit exactly
    _close(_uniform_scene(), UNIFORM_RGB_EXPECTED, 0)


def test_yuv_uniform_color_yuv_output():
    rgba = _uniform_scene()
    y, u, v = orc.rgba_to_planar_yuv(rgba, orc.YUV420)
    # SURVEY Appendix B worked example: Y=25, U=123, V=150
    assert int(y[0, 0]) == 25 and int(u[0, 0]) == 123 and int(v[0, 0]) == 150
    back = orc.harness_yuv420_to_rgba(y, u, v, 8, 2)
    _close(back, UNIFORM_YUV_EXPECTED, 2)


def _gradient_node_texture():
    # gradient.wgsl: fragment returns (tex_coords.x, 0, 0, 1) in linear light into an
    # sRGB render target (ShaderNode target = node texture srgb view), 8x2.
    tex = np.zeros((2, 8, 4), np.uint8)
    for x in range(8):
        lin = np.float32((x + 0.5) / 8.0)
        tex[:, x] = [orc.srgb_encode8(lin), 0, 0, 255]
    return tex


def test_yuv_gradient_rgba_output():
    _close(_gradient_node_texture(), GRADIENT_RGB_EXPECTED, 2)
    _close(_gradient_node_texture(), GRADIENT_RGB_EXPECTED, 0)


def test_yuv_gradient_through_the_shader_node_port():
    # the same test driven through the ShaderNode restatement (one plane, plane_id -1, cleared target)
    tex = orc.builtin_shader(orc.SHADER_GRADIENT, [], 8, 2)
    assert np.array_equal(tex, _gradient_node_texture())
    _close(tex, GRADIENT_RGB_EXPECTED, 0)


def test_shader_ports_plane_rules():
    """shader/pipeline.rs:81-141: plane_id -1 when the node has no sources, one plane per source otherwise, premultiplied OVER."""
    rng = np.random.default_rng(5)
    tex = [rng.integers(0, 256, (36, 64, 4), dtype=np.uint8) for _ in range(4)]
    for t in tex:
        t[..., 3] = 255
    assert orc.builtin_shader(orc.SHADER_LAYOUT_PLANES, [], 64, 36)[7, 9].tolist() == [255, 0, 0, 255]
    quad = orc.builtin_shader(orc.SHADER_LAYOUT_PLANES, tex, 128, 72)
    # every quadrant is the source texel for texel (scale 1, sample positions on texel centres)
    assert np.array_equal(quad[:36, :64], tex[0]) and np.array_equal(quad[:36, 64:], tex[1])
    assert np.array_equal(quad[36:, :64], tex[2]) and np.array_equal(quad[36:, 64:], tex[3])
    for n, want in ((0, [255, 0, 0, 255]), (1, [0, 255, 0, 255]), (2, [0, 0, 255, 255])):
        assert orc.builtin_shader(orc.SHADER_COLOR_BY_TEXTURE_COUNT, tex[:n], 4, 4)[0, 0].tolist() == want
    # silly.wgsl returns transparent unless it has exactly one texture; fade_to_ball at t = 0 keeps a disc of radius < eps
    assert not orc.builtin_shader(orc.SHADER_SILLY, tex[:2], 16, 16).any()
    ball = orc.builtin_shader(orc.SHADER_FADE_TO_BALL, tex[:1], 64, 36, time=0.0)
    assert not ball[0, 0].any() and ball[18, 32, 3] > 0
    # two half-transparent planes over each other: 0.5 + 0.5 * (1 - 0.5)
    half = orc.circle_layout_params([(0, 0, 8, 8, (0.0, 0.0, 0.0, 0.5))] * 2)
    sq = np.zeros((8, 8, 4), np.uint8)
    out = orc.builtin_shader(orc.SHADER_CIRCLE_LAYOUT, [sq, sq], 8, 8, params=half)
    assert out[0, 0, 3] in (191, 192) and out[4, 4, 3] == 0  # corners are outside the circle, the centre samples the empty texture


def test_yuv_gradient_yuv_output():
    tex = _gradient_node_texture()
    y, u, v = orc.rgba_to_planar_yuv(tex, orc.YUV420)
    back = orc.harness_yuv420_to_rgba(y, u, v, 8, 2)
    _close(back, GRADIENT_YUV_EXPECTED, 2)


# ---- pixel_input_format_tests.rs --------------------------------------------------
INPUT_BYTES = list(range(1, 65))
BGRA_EXPECTED = [3, 2, 1, 4, 7, 6, 5, 8, 11, 10, 9, 12, 15, 14, 13, 16, 19, 18, 17, 20, 23, 22, 21, 24,
                 27, 26, 25, 28, 31, 30, 29, 32, 35, 34, 33, 36, 39, 38, 37, 40, 43, 42, 41, 44, 47, 46, 45, 48,
                 51, 50, 49, 52, 55, 54, 53, 56, 59, 58, 57, 60, 63, 62, 61, 64]
ARGB_EXPECTED = [4, 1, 2, 3, 8, 5, 6, 7, 12, 9, 10, 11, 16, 13, 14, 15, 20, 17, 18, 19, 24, 21, 22, 23,
                 28, 25, 26, 27, 32, 29, 30, 31, 36, 33, 34, 35, 40, 37, 38, 39, 44, 41, 42, 43, 48, 45, 46, 47,
                 52, 49, 50, 51, 56, 53, 54, 55, 60, 57, 58, 59, 64, 61, 62, 63]


def _view_with_input(node):
    # View{ children: [InputStream] } default View (transparent bg, overflow hidden):
    # flatten gives [colour layout (culled: alpha 0, no border), texture layout 8x2 crop whole].
    h, w = node.shape[:2]
    layouts = [Layout(top=0, left=0, width=w, height=h, type=0, source_index=0, crop=(0, 0, w, h))]
    return orc.apply_layouts(w, h, layouts, [node], srgb=True)


@pytest.mark.parametrize("kind,expected", [(0, BGRA_EXPECTED), (1, ARGB_EXPECTED)])
def test_pixel_format_inputs_exact(kind, expected):
    data = np.array(INPUT_BYTES, np.uint8).reshape(2, 8, 4)
    node = orc.swizzle_to_rgba(data, 8, 2, kind)
    _close(node, expected, 0)
    # ... and the whole View -> sRGB-view sample -> blend -> sRGB encode path keeps the bytes exact
    _close(_view_with_input(node), expected, 0)


# ---- resampler.rs unit tests ------------------------------------------------------
def _plan(left, top, width, height, dst):
    return orc.resample_plan(4096, 4096, (top, left, width, height), dst[0], dst[1])


def test_plans_a_pass_for_every_non_direct_axis():
    assert _plan(0.0, 0.0, 640.0, 360.0, (640, 360)).kind == 0
    assert _plan(100.0, 40.0, 640.0, 360.0, (640, 360)).kind == 0
    p = _plan(100.0, 0.0, 640.0, 360.0, (640, 300))
    assert p.kind == 1 and (p.axis[0], p.perp_offset[0]) == (1, 100)
    p = _plan(0.0, 42.0, 640.0, 360.0, (320, 360))
    assert p.kind == 1 and (p.axis[0], p.perp_offset[0]) == (0, 42)
    assert _plan(100.5, 0.0, 640.0, 360.0, (640, 300)).kind == 2
    p = _plan(0.0, 0.0, 1920.0, 1080.0, (960, 270))
    assert p.kind == 2 and (p.axis[0], p.axis[1]) == (1, 0)


def test_predecimation_levels():
    # scale 9 -> ceil(log2(9/4)) = 2 levels (factor 4), residual 2.25
    p = orc.resample_plan(5760, 3240, (0, 0, 5760, 3240), 640, 360)
    assert p.levels == (2, 2) and p.reduced == (1440, 810)
    # scale exactly 4 stays on the kernel alone
    p = orc.resample_plan(3840, 2160, (0, 0, 3840, 2160), 960, 540)
    assert p.levels == (0, 0)
    # 3:1 multiscale-grid case (rescaler.rs:813-859) needs no box pass
    p = orc.resample_plan(5760, 3240, (0, 0, 5760, 3240), 1920, 1080)
    assert p.levels == (0, 0) and p.kind == 2


# ---- scalar conventions -------------------------------------------------------------
def test_srgb_roundtrip_is_identity():
    dec = orc.srgb_decode_table()
    for i in range(256):
        assert orc.srgb_encode8(float(dec[i])) == i


def test_f16_conversion_matches_numpy():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(2000).astype(np.float32),
                         (rng.standard_normal(500) * 1e-6).astype(np.float32),
                         np.array([0.0, -0.0, 1.0, 65504.0, 65520.0, 1e-8, 6.1e-5, 5.96e-8], np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([orc.f32_to_f16_bits(float(x)) for x in xs], np.uint16)
    assert (want == got).all()


def test_black_fallback_yuv():
    # render_loop.rs:127-139 -> RGBColor::BLACK.to_yuv(): Y=16, U=V=128
    assert orc.rgb_to_yuv_bytes(0, 0, 0) == (16, 128, 128)


def test_c_frame_loop_equals_the_pass_by_pass_pipeline():
    """orc_render_frame_yuv420 (the all-C frame loop timed as bench.py's cpu_baseline) is the same pass sequence as the
    pass-by-pass Python pipeline the parity tests compare the GPU with: identical bytes."""
    from tests import refpipe, scenes
    iw, ih, W, H, n = 96, 54, 192, 108, 3
    layouts, res = scenes.cfg3_scene(iw, ih, W, H, n)
    planes = [scenes.test_input(i, iw, ih, noise_seed=5 + i) for i in range(n)]
    rng = np.random.default_rng(3)
    label = rng.integers(0, 256, (scenes.LABEL_H, scenes.LABEL_W, 4), dtype=np.uint8)
    label[..., :3] = np.minimum(label[..., :3], label[..., 3:])  # premultiplied
    nodes, sources, k = [], [], 0
    for r in res:
        if r == (iw, ih):
            nodes.append(orc.planar_yuv_to_rgba(*planes[k], iw, ih)); sources.append(k); k += 1
        else:
            nodes.append(label); sources.append(label)
    want, _ = refpipe.render_yuv420(layouts, nodes, W, H)
    got = orc.render_frame_yuv420(planes, layouts, sources, W, H)
    for g, w_ in zip(got, want):
        assert (g == w_).all()


@pytest.mark.parametrize("sw,dw", [(192, 128), (180, 80), (100, 100 * 3 // 7), (64, 160)])
def test_lanczos_pass_agrees_with_pillow_on_linear_planes(sw, dw):
    """A third party's Lanczos3 (Pillow's Image.resize(..., LANCZOS) on mode-F planes: same kernel, a = 3, stretched by the scale
    factor when shrinking, weights normalised) against the oracle's restatement of resample.wgsl:31-87 — the part of the oracle
    no reference-held vector pins.  One axis at a time (the oracle rounds its intermediate to f16 as the reference does, Pillow
    keeps f32), interior pixels only (edges: clamp-to-edge vs Pillow's truncated window).  Tolerance: the f16 rounding of the
    oracle's input and output (2^-10 relative) plus 1e-4 absolute."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    rng = np.random.default_rng(sw * 1000 + dw)
    sh = 24
    plane = (0.5 + 0.35 * np.sin(np.arange(sw) / 7.0)[None, :] * np.cos(np.arange(sh) / 5.0)[:, None] + rng.uniform(-0.12, 0.12, (sh, sw))).astype(np.float16)
    src = np.repeat(plane[:, :, None], 4, axis=2).view(np.uint16)  # RGBA16F, every channel the same plane
    scale = sw / dw
    got = orc.resample_pass(src, orc.PX_RGBA16F, 0, scale, 0.0, 0, orc.PX_RGBA16F, dw, sh).view(np.float16)[:, :, 0].astype(np.float64)
    ref = np.asarray(Image.fromarray(plane.astype(np.float32), mode="F").resize((dw, sh), Image.LANCZOS, box=(0, 0, sw, sh)), dtype=np.float64)
    # Pillow resizes both axes even when one is unchanged (identity there: size equal) — compare the interior columns
    m = int(np.ceil(3 * max(scale, 1.0) / scale)) + 1
    d = np.abs(got[:, m:-m] - ref[:, m:-m])
    tol = 2.0 ** -10 * np.abs(ref[:, m:-m]) + 1e-4
    assert (d <= tol).all(), f"max excess {(d - tol).max():.2e}"
    # vertical pass: the same through a transposed plane
    src_t = np.ascontiguousarray(np.repeat(plane.T[:, :, None], 4, axis=2)).view(np.uint16)
    got_v = orc.resample_pass(src_t, orc.PX_RGBA16F, 1, scale, 0.0, 0, orc.PX_RGBA16F, sh, dw).view(np.float16)[:, :, 0].astype(np.float64)
    d = np.abs(got_v[m:-m, :] - ref.T[m:-m, :])
    assert (d <= 2.0 ** -10 * np.abs(ref.T[m:-m, :]) + 1e-4).all()


def test_byte_over_255_by_one_fma_is_the_ieee_quotient():
    """smr_convert_dev.h unorm_of_byte: RN(a * RN(1/255) + a * -2^-33) == RN(a / 255) for every byte (exact rational arithmetic,
    rounded to nearest-even f32 once — what v_fma_f32 does)."""
    from fractions import Fraction

    def rn_f32(q: Fraction) -> float:
        if q == 0:
            return 0.0
        e = 0
        while Fraction(2) ** (e + 1) <= q:
            e += 1
        while Fraction(2) ** e > q:
            e -= 1
        ulp = Fraction(2) ** (e - 23)
        n, rem = divmod(q, ulp)
        n = int(n)
        if rem * 2 > ulp or (rem * 2 == ulp and n % 2 == 1):
            n += 1
        return float(Fraction(n) * ulp)

    rb = Fraction(float(np.float32(1.0) / np.float32(255.0)))
    c = -Fraction(1, 2 ** 33)
    for a in range(256):
        got = np.float32(rn_f32(a * rb + a * c))
        want = np.float32(a) / np.float32(255.0)
        assert got == want, a
