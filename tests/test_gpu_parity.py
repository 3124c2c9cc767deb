"""GPU parity tests proper: every C-ABI pass against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): bit-exact for integer blit/copy paths; <= 1 LSB per channel for
colour conversion / resampling / blending (tolerance written in each test).  All calls go
through libsmr_hip.so (ctypes -> extern "C")."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle.oracle import Glyph, Layout, Mask
from tests import refpipe, scenes

pytestmark = pytest.mark.gpu

TOL = 1  # LSB per channel, float paths


@pytest.fixture(scope="module")
def hip():
    from smelter_amd import hip as h
    return h


@pytest.fixture(scope="module")
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_cpu(hip):
    c = hip.Context(0, mode=hip.MODE_CPU_OPTIMIZED)
    yield c
    c.close()


def check(got, want, tol, min_exact=0.0, what=""):
    d = refpipe.max_diff(got, want)
    ex = refpipe.exact_fraction(got, want)
    assert d <= tol, f"{what}: max |diff| = {d} LSB (> {tol}); exact fraction {ex:.5f}"
    assert ex >= min_exact, f"{what}: only {ex:.5f} of bytes identical (< {min_exact})"
    return d, ex


# ------------------------------------------------------------------ a3 converters
@pytest.mark.parametrize("w,h", [(64, 36), (640, 360), (66, 38), (37, 21), (2, 2), (1922, 1082)])
@pytest.mark.parametrize("variant", ["420", "422", "444", "j420"])
def test_planar_yuv_to_rgba(ctx, hip, w, h, variant):
    fmt = {"420": hip.FRAME_PLANAR_YUV420, "422": hip.FRAME_PLANAR_YUV422, "444": hip.FRAME_PLANAR_YUV444,
           "j420": hip.FRAME_PLANAR_YUVJ420}[variant]
    ov = {"420": orc.YUV420, "422": orc.YUV422, "444": orc.YUV444, "j420": orc.YUVJ420}[variant]
    rng = np.random.default_rng(hash((w, h, variant)) % 2**32)
    ch, cw = orc.chroma_shape(w, h, ov)
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    u = rng.integers(0, 256, (ch, cw), dtype=np.uint8)
    v = rng.integers(0, 256, (ch, cw), dtype=np.uint8)
    f = ctx.frame(fmt, w, h, [y, u, v])
    got = ctx.frame_to_rgba(f).download()
    want = orc.planar_yuv_to_rgba(y, u, v, w, h, ov)
    check(got, want, TOL, 0.999, f"planar {variant} {w}x{h}")
    assert (got[..., 3] == 255).all()


@pytest.mark.parametrize("w,h", [(64, 36), (66, 38), (2, 2), (6, 4), (8, 2), (12, 6), (260, 10), (258, 10), (1920, 1080), (1924, 1082), (1922, 1082), (3840, 2160), (38, 21), (37, 21)])
@pytest.mark.parametrize("variant", ["420", "j420", "nv12", "422", "444"])
def test_batched_converter_equals_the_general_kernel(ctx, hip, w, h, variant):
    """The block converters — k_yuv420_to_rgba (4:2:0 planar / NV12: a thread per 4 x 4 block; the default) and k_yuv_to_rgba_batch (4 x 2
    blocks: 4:2:2 / 4:4:4, and 4:2:0 with SMR_CONVERT_BLOCK_4X2) — against k_yuv_to_rgba (the WGSL pass as it is written,
    SMR_CONVERT_GENERAL): every byte equal, on white noise and on the extremes; and the 4:2:0 results equal to the oracle's bytes."""
    rng = np.random.default_rng(hash((w, h, variant)) % 2**32)
    if (variant in ("420", "j420", "nv12") and (w % 2 or h % 2)) or (variant == "422" and w % 2):
        pytest.skip("odd size along a subsampled axis: the general kernel's case")
    cw, ch = (w if variant == "444" else w // 2), (h if variant in ("422", "444") else h // 2)
    for content in ("noise", "extremes"):
        if content == "noise":
            y = rng.integers(0, 256, (h, w), dtype=np.uint8)
            c = rng.integers(0, 256, (ch, cw, 2), dtype=np.uint8)
        else:
            y = rng.choice(np.array([0, 1, 15, 16, 17, 234, 235, 236, 254, 255], np.uint8), (h, w))
            c = rng.choice(np.array([0, 15, 16, 17, 127, 128, 129, 239, 240, 241, 255], np.uint8), (ch, cw, 2))
        if variant == "nv12":
            f = ctx.frame(hip.FRAME_NV12, w, h, [y, c])
        else:
            fmt = {"420": hip.FRAME_PLANAR_YUV420, "j420": hip.FRAME_PLANAR_YUVJ420, "422": hip.FRAME_PLANAR_YUV422, "444": hip.FRAME_PLANAR_YUV444}[variant]
            f = ctx.frame(fmt, w, h, [y, np.ascontiguousarray(c[..., 0]), np.ascontiguousarray(c[..., 1])])
        try:
            fast = ctx.frame_to_rgba(f).download()
            ctx.set_convert_impl(hip.CONVERT_BLOCK_4X2)
            block42 = ctx.frame_to_rgba(f).download()
            ctx.set_convert_impl(hip.CONVERT_GENERAL)
            general = ctx.frame_to_rgba(f).download()
        finally:
            ctx.set_convert_impl(hip.CONVERT_AUTO)
        assert np.array_equal(fast, general), (variant, w, h, content, int((fast != general).sum()))
        assert np.array_equal(block42, general), (variant, w, h, content, int((block42 != general).sum()))
        if variant in ("420", "j420"):
            want = orc.planar_yuv_to_rgba(y, np.ascontiguousarray(c[..., 0]), np.ascontiguousarray(c[..., 1]), w, h, orc.YUVJ420 if variant == "j420" else orc.YUV420)
            assert np.array_equal(fast, want), (variant, w, h, content, int((fast != want).sum()))
        elif variant == "nv12":
            assert np.array_equal(fast, orc.nv12_to_rgba(y, c, w, h)), (variant, w, h, content)


@pytest.mark.parametrize("variant,w,h", [("nv12", 1280, 720), ("nv12", 3840, 16), ("nv12", 256, 18), ("420", 2048, 16), ("420", 512, 34), ("j420", 1024, 6), ("420", 4096, 8)])
def test_chroma_rows_that_fill_their_pitch_take_the_block_converter(ctx, hip, variant, w, h):
    """k_yuv420_to_rgba reads (and ignores) up to a dword past the last block's chroma window.  Where a chroma row's bytes are a multiple of
    256 — 720p and 4K NV12, planar frames 512 / 1024 / 2048 / 4096 wide — that dword lies past the row's pitch: in the next row, or, for
    the last row, in the spare bytes every allocation of the library ends with (SMR_SURFACE_TAIL).  Such frames used to fall back to the
    general converter; a plane the library allocated now takes the block converter (its launch shows in SMR_KERNEL_FRAME_TO_RGBA_420) and
    writes the oracle's bytes.  (Found by running the emulated kernel under AddressSanitizer: tests/test_emu_asan.py.)"""
    rng = np.random.default_rng(w * 31 + h)
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    c = rng.integers(0, 256, (h // 2, w // 2, 2), dtype=np.uint8)
    if variant == "nv12":
        f = ctx.frame(hip.FRAME_NV12, w, h, [y, c])
        want = orc.nv12_to_rgba(y, c, w, h)
    else:
        u, v = np.ascontiguousarray(c[..., 0]), np.ascontiguousarray(c[..., 1])
        f = ctx.frame(hip.FRAME_PLANAR_YUVJ420 if variant == "j420" else hip.FRAME_PLANAR_YUV420, w, h, [y, u, v])
        want = orc.planar_yuv_to_rgba(y, u, v, w, h, orc.YUVJ420 if variant == "j420" else orc.YUV420)
    before = ctx.kernel_launches()
    got = ctx.frame_to_rgba(f).download()
    ran = {k: n - before[k] for k, n in ctx.kernel_launches().items()}
    assert ran["frame_to_rgba_420"] == 1 and ran["frame_to_rgba"] == 1, ran
    assert np.array_equal(got, want), int((got != want).sum())


def test_nv12_to_rgba(ctx, hip):
    w, h = 642, 362
    rng = np.random.default_rng(5)
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    uv = rng.integers(0, 256, (h // 2, w // 2, 2), dtype=np.uint8)
    got = ctx.frame_to_rgba(ctx.frame(hip.FRAME_NV12, w, h, [y, uv])).download()
    check(got, orc.nv12_to_rgba(y, uv, w, h), TOL, 0.999, "nv12")


@pytest.mark.parametrize("order", [0, 1])
def test_interleaved422_to_rgba(ctx, hip, order):
    w, h = 640, 48
    data = np.random.default_rng(6 + order).integers(0, 256, (h, w // 2, 4), dtype=np.uint8)
    fmt = hip.FRAME_UYVY422 if order == 0 else hip.FRAME_YUYV422
    got = ctx.frame_to_rgba(ctx.frame(fmt, w, h, [data])).download()
    check(got, orc.interleaved422_to_rgba(data, w, h, order), TOL, 0.999, "interleaved422")


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("w,h", [(8, 2), (10, 3), (640, 48), (1920, 1080), (1922, 1081), (6, 4)])
def test_batched_packed422_converter_equals_the_general_kernel(ctx, hip, order, w, h):
    """UYVY / YUYV through k_yuv_to_rgba_batch's packed mode against k_interleaved422_to_rgba (SMR_CONVERT_GENERAL=1): every byte equal
    (widths below 8 stay on the general kernel: there both runs are the same kernel)."""
    data = np.random.default_rng(60 + order + w).integers(0, 256, (h, w // 2, 4), dtype=np.uint8)
    f = ctx.frame(hip.FRAME_UYVY422 if order == 0 else hip.FRAME_YUYV422, w, h, [data])
    fast = ctx.frame_to_rgba(f).download()
    ctx.set_convert_impl(hip.CONVERT_GENERAL)
    try:
        general = ctx.frame_to_rgba(f).download()
    finally:
        ctx.set_convert_impl(hip.CONVERT_AUTO)
    assert np.array_equal(fast, general), (order, w, h, int((fast != general).sum()))
    check(fast, orc.interleaved422_to_rgba(data, w, h, order), TOL, 0.999, "interleaved422")


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("w,h", [(8, 2), (641, 33)])
def test_swizzles_bit_exact(ctx, hip, kind, w, h):
    data = np.random.default_rng(7).integers(0, 256, (h, w, 4), dtype=np.uint8)
    fmt = hip.FRAME_BGRA if kind == 0 else hip.FRAME_ARGB
    got = ctx.frame_to_rgba(ctx.frame(fmt, w, h, [data])).download()
    assert (got == orc.swizzle_to_rgba(data, w, h, kind)).all()


def test_pixel_format_golden_through_gpu(ctx, hip):
    # integration-tests/src/render_tests/pixel_input_format_tests.rs, end to end on the GPU, exact
    from tests.test_oracle_golden import ARGB_EXPECTED, BGRA_EXPECTED, INPUT_BYTES
    data = np.array(INPUT_BYTES, np.uint8).reshape(2, 8, 4)
    for fmt, exp in ((hip.FRAME_BGRA, BGRA_EXPECTED), (hip.FRAME_ARGB, ARGB_EXPECTED)):
        f = ctx.frame(fmt, 8, 2, [data])
        out = ctx.surface(8, 2)
        ctx.render_layouts([Layout(top=0, left=0, width=8, height=2, type=0, source_index=0, crop=(0, 0, 8, 2))], [f], 8, 2,
                           out_rgba=out)
        assert out.download().ravel().tolist() == exp


def test_yuv_golden_through_gpu(ctx, hip):
    # integration-tests/src/render_tests/yuv_tests.rs:84-131 (uniform colour), +-2 in the reference, exact here
    from tests.test_oracle_golden import UNIFORM_RGB_EXPECTED, UNIFORM_YUV_EXPECTED
    col = orc.color_to_shader((50, 0, 0, 255), True)
    lay = [Layout(top=0, left=0, width=8, height=2, type=1, color=col)]
    out_rgba = ctx.surface(8, 2)
    ctx.render_layouts(lay, [], 8, 2, out_rgba=out_rgba)
    assert out_rgba.download().ravel().tolist() == UNIFORM_RGB_EXPECTED
    out = ctx.frame(hip.FRAME_PLANAR_YUV420, 8, 2)
    ctx.render_layouts(lay, [], 8, 2, out=out)
    y, u, v = out.download()
    assert refpipe.max_diff(orc.harness_yuv420_to_rgba(y, u, v, 8, 2).ravel(), UNIFORM_YUV_EXPECTED) <= 2
    assert (y == 25).all() and (u == 123).all() and (v == 150).all()


@pytest.mark.parametrize("srgb", [True, False])
def test_premultiply(ctx, ctx_cpu, hip, srgb):
    c = ctx if srgb else ctx_cpu
    data = np.random.default_rng(8).integers(0, 256, (37, 65, 4), dtype=np.uint8)
    got = c.add_premultiplied_alpha(c.surface_from(data)).download()
    check(got, orc.add_premultiplied_alpha(data, srgb), TOL, 0.999, "add premult")
    got = c.remove_premultiplied_alpha(c.surface_from(data)).download()
    check(got, orc.remove_premultiplied_alpha(data), TOL, 0.999, "remove premult")


# ------------------------------------------------------------------ a11 outputs
@pytest.mark.parametrize("w,h", [(64, 36), (640, 360), (65, 37), (1920, 1080)])
@pytest.mark.parametrize("variant", ["420", "422", "444", "nv12"])
def test_rgba_to_frame(ctx, hip, w, h, variant):
    rgba = np.random.default_rng(9).integers(0, 256, (h, w, 4), dtype=np.uint8)
    node = ctx.surface_from(rgba)
    if variant == "nv12":
        got = ctx.rgba_to_frame(node, hip.FRAME_NV12).download()
        want = orc.rgba_to_nv12(rgba)
    else:
        fmt = {"420": hip.FRAME_PLANAR_YUV420, "422": hip.FRAME_PLANAR_YUV422, "444": hip.FRAME_PLANAR_YUV444}[variant]
        ov = {"420": orc.YUV420, "422": orc.YUV422, "444": orc.YUV444}[variant]
        got = ctx.rgba_to_frame(node, fmt).download()
        want = orc.rgba_to_planar_yuv(rgba, ov)
    for g, w_ in zip(got, want):
        check(g, w_, TOL, 0.999, f"rgba->{variant}")


@pytest.mark.parametrize("w,h", [(64, 36), (66, 38), (2, 2), (6, 4), (258, 11), (37, 21), (1920, 1080), (3840, 2160)])
@pytest.mark.parametrize("variant", ["420", "422", "444", "nv12"])
def test_one_launch_output_converter_equals_the_three_passes(ctx, hip, w, h, variant):
    """k_rgba_to_planes (one launch, a 4 x 2 pixel block per thread) against k_rgba_to_y + k_rgba_to_chroma (rgba_to_yuv.wgsl's passes as
    they are written, SMR_CONVERT_GENERAL): every byte of every plane equal."""
    if (variant in ("420", "nv12") and (w % 2 or h % 2)) or (variant == "422" and w % 2):
        pytest.skip("odd size along a subsampled axis: the three-pass kernels' case")
    rng = np.random.default_rng(hash((w, h, variant)) % 2**32)
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgba[: h // 2, :, :3] = rng.choice(np.array([0, 1, 127, 128, 254, 255], np.uint8), (h // 2, w, 3))
    node = ctx.surface_from(rgba)
    fmt = {"420": hip.FRAME_PLANAR_YUV420, "422": hip.FRAME_PLANAR_YUV422, "444": hip.FRAME_PLANAR_YUV444, "nv12": hip.FRAME_NV12}[variant]
    fast = ctx.rgba_to_frame(node, fmt).download()
    ctx.set_convert_impl(hip.CONVERT_GENERAL)
    try:
        general = ctx.rgba_to_frame(node, fmt).download()
    finally:
        ctx.set_convert_impl(hip.CONVERT_AUTO)
    for a, b, pl in zip(fast, general, "YUV"):
        assert np.array_equal(a, b), (variant, w, h, pl, int((a != b).sum()))


def test_black_fallback(ctx, hip):
    out = ctx.frame(hip.FRAME_PLANAR_YUV420, 66, 34)
    ctx.fill_black(out)
    y, u, v = out.download()
    assert (y == 16).all() and (u == 128).all() and (v == 128).all()


# ------------------------------------------------------------------ a7/a8 resampler
RESAMPLE_CASES = [
    # (src_w, src_h, crop(top,left,w,h), dst_w, dst_h)
    (640, 360, (0, 0, 640, 360), 320, 180),        # 2:1 separable, H first (equal scales)
    (640, 360, (0, 0, 640, 360), 427, 240),        # 1.5:1
    (640, 360, (0, 0, 640, 360), 960, 540),        # upscale
    (640, 360, (0, 0, 640, 360), 422, 238),        # vertical-first plan (stronger vertical shrink)
    (640, 360, (0, 100, 640 - 100, 360), 540, 300),  # single vertical pass, perp offset 100
    (640, 360, (42, 0, 640, 360 - 42), 320, 318),  # single horizontal pass, perp offset 42
    (640, 360, (10.5, 20.25, 300.5, 200.0), 97, 311),  # fractional crop, mixed up/down
    (1920, 1080, (0, 0, 1920, 1080), 213, 120),    # 9:1 -> box pre-reduction (levels 2)
    (1920, 1080, (0, 0, 1920, 1080), 480, 270),    # exactly 4:1: kernel alone, 25 taps
    (333, 77, (0, 0, 333, 77), 100, 200),          # odd sizes
]


@pytest.mark.parametrize("case", RESAMPLE_CASES, ids=[f"{c[0]}x{c[1]}->{c[3]}x{c[4]}" for c in RESAMPLE_CASES])
def test_resample_vs_oracle(ctx, hip, case):
    sw, sh, crop, dw, dh = case
    rng = np.random.default_rng(sw * 31 + dw)
    y, u, v = scenes.test_input(3, sw - sw % 2, sh - sh % 2, noise_seed=1)
    src = orc.planar_yuv_to_rgba(y, u, v, sw - sw % 2, sh - sh % 2)
    if src.shape[:2] != (sh, sw):
        src = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
    src[..., 3] = rng.integers(0, 256, (sh, sw), dtype=np.uint8)  # exercise the alpha channel too
    src[..., :3] = (src[..., :3].astype(np.uint16) * src[..., 3:4] // 255).astype(np.uint8)  # premultiplied
    s = ctx.surface_from(src)
    d = ctx.surface(dw, dh)
    kind = ctx.resample(s, crop, d)
    okind, want = orc.resample(src, crop, dw, dh)
    assert kind == okind and kind > 0
    check(d.download(), want, TOL, 0.995, f"resample {case}")


def test_resample_direct_is_a_noop(ctx, hip):
    s = ctx.surface(640, 360)
    d = ctx.surface(640, 360)
    assert ctx.resample(s, (40, 100, 640, 360), d) == 0


def test_single_passes_and_downsample(ctx, hip):
    src = np.random.default_rng(11).integers(0, 256, (90, 160, 4), dtype=np.uint8)
    s = ctx.surface_from(src)
    red = ctx.surface(40, 45, hip.PX_RGBA16F)
    ctx.downsample(s, 4, 2, red)
    want = orc.downsample(src, orc.PX_RGBA8_SRGB, 4, 2)
    got = red.download()
    # f16 bit patterns: identical up to 1 ulp of f16 where the f32 mean lands on a rounding boundary
    assert np.abs(got.view(np.int16).astype(np.int32) - want.view(np.int16).astype(np.int32)).max() <= 1
    mid = ctx.surface(100, 45, hip.PX_RGBA16F)
    ctx.resample_pass(red, 0, 0.4, 0.0, 0, mid)
    want_mid = orc.resample_pass(want, orc.PX_RGBA16F, 0, 0.4, 0.0, 0, orc.PX_RGBA16F, 100, 45)
    assert np.abs(mid.download().view(np.int16).astype(np.int32) - want_mid.view(np.int16).astype(np.int32)).max() <= 2


@pytest.mark.parametrize("dw,dh", [(1, 1), (2, 1), (4, 2), (7, 5), (16, 9)])
def test_large_box_reductions_keep_the_reference_sum_order(ctx, hip, dw, dh):
    """A 1080p child laid out into a few pixels reduces by up to 512 x 512 before the Lanczos passes (predecimate_levels,
    resampler.rs:60-66).  The one-wave-per-pixel box kernel fetches in parallel but adds in the reference's order: the reduced
    surface equals the oracle's sequential sum bit for bit (f16 patterns), and the whole resample stays within 1 LSB — in
    milliseconds, not tens of them."""
    import time
    sw, sh = 1920, 1080
    y, u, v = scenes.test_input(5, sw, sh, noise_seed=3)
    src = orc.planar_yuv_to_rgba(y, u, v, sw, sh)
    s = ctx.surface_from(src)
    plan = orc.resample_plan(sw, sh, (0.0, 0.0, float(sw), float(sh)), dw, dh)
    fx, fy = 1 << plan.levels[0], 1 << plan.levels[1]
    assert fx * fy >= 256
    red = ctx.surface((sw + fx - 1) // fx, (sh + fy - 1) // fy, hip.PX_RGBA16F)
    ctx.downsample(s, fx, fy, red)
    want_red = orc.downsample(src, orc.PX_RGBA8_SRGB, fx, fy)
    assert (red.download().view(np.int16) == want_red.view(np.int16)).all()
    d = ctx.surface(dw, dh)
    ctx.resample(s, (0.0, 0.0, float(sw), float(sh)), d)
    ctx.sync()
    t0 = time.perf_counter()
    kind = ctx.resample(s, (0.0, 0.0, float(sw), float(sh)), d)
    ctx.sync()
    assert time.perf_counter() - t0 < 0.01
    okind, want = orc.resample(src, (0.0, 0.0, float(sw), float(sh)), dw, dh)
    assert kind == okind
    assert np.abs(d.download().astype(np.int32) - want.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("srgb", [True, False])
def test_rescale_bilinear(ctx, ctx_cpu, hip, srgb):
    c = ctx if srgb else ctx_cpu
    src = np.random.default_rng(12).integers(0, 256, (90, 160, 4), dtype=np.uint8)
    d = c.surface(123, 77)
    c.rescale_bilinear(c.surface_from(src), d)
    check(d.download(), orc.rescale_bilinear(src, 123, 77, orc.PX_RGBA8_SRGB if srgb else orc.PX_RGBA8_UNORM), TOL, 0.99, "rescale")


# ------------------------------------------------------------------ a15 FramePreProcessor::process_to_bytes, chained
@pytest.mark.parametrize("fmt_name,size", [("420", None), ("420", (123, 77)), ("nv12", (200, 120)), ("j420", (64, 36)), ("uyvy", (96, 54)),
                                           ("bgra", None), ("bgra", (50, 30)), ("444", (161, 91))])
def test_frame_preprocess_chain(ctx, hip, fmt_name, size):
    """smr_frame_preprocess = upload_and_convert_to_node_texture -> rescale_node_texture (optional) -> download
    (state/frame_pre_processor.rs:84-132) in one call, against the oracle's converter followed by its bilinear rescale."""
    w, h = 160, 90
    rng = np.random.default_rng(7)
    if fmt_name in ("420", "j420", "444"):
        ov = {"420": orc.YUV420, "j420": orc.YUVJ420, "444": orc.YUV444}[fmt_name]
        ch, cw = orc.chroma_shape(w, h, ov)
        y, u, v = (rng.integers(0, 256, s, dtype=np.uint8) for s in ((h, w), (ch, cw), (ch, cw)))
        f = ctx.frame({"420": hip.FRAME_PLANAR_YUV420, "j420": hip.FRAME_PLANAR_YUVJ420, "444": hip.FRAME_PLANAR_YUV444}[fmt_name], w, h, [y, u, v])
        node = orc.planar_yuv_to_rgba(y, u, v, w, h, ov)
    elif fmt_name == "nv12":
        y = rng.integers(0, 256, (h, w), dtype=np.uint8)
        uv = rng.integers(0, 256, (h // 2, w // 2, 2), dtype=np.uint8)
        f = ctx.frame(hip.FRAME_NV12, w, h, [y, uv])
        node = orc.nv12_to_rgba(y, uv, w, h)
    elif fmt_name == "uyvy":
        data = rng.integers(0, 256, (h, w // 2, 4), dtype=np.uint8)
        f = ctx.frame(hip.FRAME_UYVY422, w, h, [data])
        node = orc.interleaved422_to_rgba(data, w, h, 0)
    else:
        data = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        f = ctx.frame(hip.FRAME_BGRA, w, h, [data])
        node = orc.swizzle_to_rgba(data, w, h, 0)
    got = ctx.frame_preprocess(f, size)
    want = node if size is None else orc.rescale_bilinear(node, size[0], size[1], orc.PX_RGBA8_SRGB)
    assert got.shape == want.shape
    check(got, want, TOL, 0.99, f"preprocess {fmt_name} -> {size}")
    # the same bytes as the three public passes one by one
    n = ctx.frame_to_rgba(f)
    if size is not None:
        d = ctx.surface(size[0], size[1])
        ctx.rescale_bilinear(n, d)
        n = d
    assert (n.download() == got).all()
    with pytest.raises(hip.SmrError):
        ctx.frame_preprocess(f, (0, 5))


# ------------------------------------------------------------------ a9/a10 compositor
def _layout_zoo(W, H):
    col = lambda c: orc.color_to_shader(c, True)
    m1 = Mask((30, 30, 30, 30), 20, 30, W - 80, H - 60)
    m2 = Mask((0, 12, 24, 36), 40.5, 50.25, W / 2, H / 2)
    return [
        Layout(0, 0, W, H, type=1, color=col((32, 32, 48, 255))),
        Layout(10, 12, 200, 120, type=1, color=col((200, 30, 30, 255)), border_radius=(20, 20, 20, 20)),
        Layout(30.5, 180.25, 150.5, 90.75, type=1, color=col((30, 200, 30, 180)), border_radius=(0, 10, 20, 30),
               border_width=6.0, border_color=col((255, 255, 255, 255))),
        Layout(100, 60, 220, 140, type=2, color=col((0, 0, 0, 200)), border_radius=(18, 18, 18, 18), blur_radius=20.0),
        Layout(90, 50, 220, 140, type=0, source_index=0, crop=(0, 0, 220, 140), border_radius=(10, 10, 10, 10)),
        Layout(150, 300, 180, 100, type=0, source_index=1, crop=(10, 20, 300, 150), border_width=3.0,
               border_color=col((255, 128, 0, 255)), border_radius=(12, 12, 12, 12), rotation_degrees=17.0),
        Layout(60, 330, 200, 150, type=0, source_index=0, crop=(0, 0, 220, 140), masks=[m1, m2]),
        Layout(200, 40, 120, 80, type=1, color=col((80, 80, 255, 128)), rotation_degrees=-33.0, masks=[m1]),
        Layout(250, 250, 90, 60, type=2, color=col((255, 0, 0, 128)), blur_radius=0.0),
        Layout(5, 400, 120, 80, type=0, source_index=7, crop=(0, 0, 10, 10)),   # missing source -> transparent
        Layout(-20, -30, 100, 90, type=1, color=col((255, 255, 0, 200)), border_radius=(45, 45, 45, 45)),  # partly off-screen
        Layout(H - 40, W - 70, 120, 90, type=0, source_index=1, crop=(0, 0, 330, 170), border_width=1.0,
               border_color=col((0, 255, 255, 255))),
    ]


@pytest.mark.parametrize("srgb", [True, False])
def test_apply_layouts_zoo(ctx, ctx_cpu, hip, srgb):
    c = ctx if srgb else ctx_cpu
    W, H = 560, 330
    rng = np.random.default_rng(13)
    srcs = []
    for (w, h) in [(220, 140), (330, 170)]:
        a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        a[..., :3] = (a[..., :3].astype(np.uint16) * a[..., 3:4] // 255).astype(np.uint8)
        srcs.append(a)
    layouts = _layout_zoo(W, H)
    if not srgb:
        for L in layouts:
            pass  # colours were converted for sRGB mode; the arithmetic under test is mode-independent
    target = c.surface(W, H)
    c.apply_layouts(target, layouts, [c.surface_from(s) for s in srcs])
    want = orc.apply_layouts(W, H, layouts, srcs, srgb=srgb)
    check(target.download(), want, TOL, 0.999, f"layout zoo srgb={srgb}")


def test_apply_layouts_integer_blit_is_bit_exact(ctx, hip):
    # pixel-aligned opaque + translucent texture blits and colour fills: integer blit/copy class -> exact
    W, H = 320, 200
    rng = np.random.default_rng(14)
    a = rng.integers(0, 256, (100, 160, 4), dtype=np.uint8)
    a[..., 3] = 255
    col = orc.color_to_shader((12, 200, 99, 255), True)
    layouts = [Layout(0, 0, W, H, type=1, color=col), Layout(20, 40, 160, 100, type=0, source_index=0, crop=(0, 0, 160, 100)),
               Layout(100, 10, 80, 50, type=0, source_index=0, crop=(25, 40, 80, 50))]
    t = ctx.surface(W, H)
    ctx.apply_layouts(t, layouts, [ctx.surface_from(a)])
    got = t.download()
    want = orc.apply_layouts(W, H, layouts, [a], srgb=True)
    assert (got == want).all()
    assert (got[20:100, 40:200] == a[:80]).all()  # the blit is a byte copy


def test_max_layouts_and_masks_limits(hip):
    c = hip.Context(0, max_layouts=3)
    col = orc.color_to_shader((255, 0, 0, 255), True)
    layouts = [Layout(0, 0, 16, 16, type=1, color=orc.color_to_shader((0, 0, i * 50, 255), True)) for i in range(5)]
    t = c.surface(16, 16)
    c.apply_layouts(t, layouts, [])
    want = orc.apply_layouts(16, 16, layouts[:3], [], srgb=True)  # params.rs:176-182: extra layouts skipped
    assert (t.download() == want).all()
    c.close()


# ------------------------------------------------------------------ a12 / a13
def test_blit_glyphs(ctx, hip):
    atlas, glyphs = scenes.label_glyphs("CAM 3 LIVE", 3)
    bg = orc.color_to_shader((0, 0, 0, 0), True)
    t = ctx.surface(scenes.LABEL_W, scenes.LABEL_H)
    ctx.blit_glyphs(t, bg, glyphs, atlas)
    want = orc.blit_glyphs(scenes.LABEL_W, scenes.LABEL_H, bg, glyphs, atlas, srgb=True)
    check(t.download(), want, TOL, 0.999, "glyph blit")
    # overlapping glyphs + coloured background
    g2 = [Glyph(2, 2, 15, 21, 0, 0, (1.0, 0.2, 0.1, 0.7)), Glyph(8, 4, 15, 21, 15, 0, (0.1, 0.9, 0.3, 1.0))]
    bg = orc.color_to_shader((20, 40, 90, 200), True)
    t = ctx.surface(40, 30)
    ctx.blit_glyphs(t, bg, g2, atlas)
    check(t.download(), orc.blit_glyphs(40, 30, bg, g2, atlas, srgb=True), TOL, 0.995, "glyph blit overlap")


@pytest.mark.parametrize("sigma", [0.0, 1.5, 4.0, 10.0, 20.0, 64.0])  # (radius 30 / 60 / 192: the other block shapes of the column pass)
def test_gaussian_blur(ctx, hip, sigma):
    src = np.random.default_rng(15).integers(0, 256, (70, 130, 4), dtype=np.uint8)
    got = ctx.gaussian_blur(ctx.surface_from(src), sigma).download()
    check(got, orc.gaussian_blur(src, sigma), TOL, 0.99, f"gaussian blur {sigma}")


@pytest.mark.parametrize("sigma", [0.0, 0.7, 3.0, 10.0, 64.0])
def test_gaussian_blur_of_an_opaque_texture_is_opaque(ctx, hip, sigma):
    """What the renderer relies on when it hands a blurred opaque node to the compositor as an opaque layer (renderer.cpp: the layer is then
    copied or sampled, not blended): both passes accumulate alpha in the order they accumulate the weights' sum, so it is exactly 1."""
    src = np.random.default_rng(16).integers(0, 256, (67, 131, 4), dtype=np.uint8)
    src[..., 3] = 255
    got = ctx.gaussian_blur(ctx.surface_from(src), sigma).download()
    assert (got[..., 3] == 255).all()
    assert (np.asarray(orc.gaussian_blur(src, sigma))[..., 3] == 255).all()


# ------------------------------------------------------------------ error behaviour
def test_errors_are_reported_not_swallowed(ctx, hip):
    with pytest.raises(hip.SmrError, match="Validation"):
        ctx.surface(0, 10)
    f = ctx.frame(hip.FRAME_PLANAR_YUV420, 64, 36)
    with pytest.raises(hip.SmrError, match="Validation"):
        ctx.frame_to_rgba(f, ctx.surface(32, 36))
    with pytest.raises(hip.SmrError, match="Validation"):
        ctx.rgba_to_frame(ctx.surface(64, 36), hip.FRAME_BGRA)
    with pytest.raises(hip.SmrError, match="Validation"):
        ctx.render_layouts([], [], 64, 36)  # no output given
