"""Which kernel a (resample plan x input format) combination runs on, asserted from smr_debug_kernel_launches, and its parity with
the oracle on that path.  The table below is the map of the fast paths and of what is still outside them:

* `wave_rgba`  k_ingest_wave on the RGBA8 node texture the exact converter wrote (every Y'CbCr format: planar 4:2:0 limited / full range,
               NV12, 4:2:2, 4:4:4, packed UYVY / YUYV) or on an opaque surface: two-pass Lanczos plans, either pass order (a vertical-first
               plan runs on the transposed node), up- and down-scaling.  THE DEFAULT ROUTE: within 1 LSB of the oracle end to end.
* `wave`       opt-in (SMR_INGEST_MFMA_F16_FUSED): the same kernel converting planar 4:2:0 / NV12 on the fly — no node texture; checked per
               stage (its conversion is within one code of the oracle's, not equal to it)
* `wave_box`   box-pre-reduced plans (shrink factors above 4) of every source, either pass order:
               the exact converter, downsample.wgsl's pass as it is (RGBA16F, linear light), then the residual Lanczos on the matrix
               cores reading the f16 texels as they are
               Single-axis plans (only the width or only the height changes) take `wave_rgba` (`wave`) too: one pass on the matrix
               cores whose f32 sums are encoded directly (the 32768 builds), height-only plans on the transposed frame / node.
               Sources with an alpha channel (BGRA / ARGB frames, translucent surfaces: premultiplied RGBA8) take the same routes
               with alpha as a fourth channel (the 65536 builds).
               Windows too wide for a column pair (scales above ~3.2: 9 or 10 k-steps) are worked one 16-column tile per wave
               (axis 4 bands) on the node-texture routes.
* `general`    smr_frame_to_rgba + smr_resample (box pre-reduction and Lanczos pass kernels, f32): nothing of this table any more —
               what still takes it are resample targets smaller than the kernel's minimum source (8 x 2) and degenerate plans.  Nothing falls to the one-launch f32 kernel
               (k_ingest_resample) any more unless SMR_INGEST_VALU_F32 asks for it.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle.oracle import Layout
from tests import convert_model

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


# name -> (source size, tile size): the plan smr_resample_plan_make gives for the full-frame crop
PLANS = {
    "two_pass_h_first": ((640, 360), (426, 240)),
    "two_pass_v_first": ((640, 360), (427, 239)),
    "upscale": ((320, 180), (480, 270)),
    "scale_2": ((640, 360), (320, 180)),
    "single_axis_h": ((640, 360), (426, 360)),
    "single_axis_v": ((640, 360), (640, 240)),
    "box_prereduced": ((1920, 1080), (400, 225)),    # 2x box, residual 2.4 both ways
    "box_prereduced_v_first": ((1280, 720), (250, 140)),   # 2x box, residual 2.57 vertically > 2.56 horizontally
    "box_prereduced_8": ((1280, 720), (160, 90)),      # 2x box, residual 4: windows too wide for a column pair — one tile per unit
    "two_pass_wide": ((1280, 720), (340, 190)),         # no box, scale 3.8: the same for a plain two-pass plan
}
FORMATS = ["yuv420", "yuvj420", "nv12", "yuv422", "yuv444", "uyvy", "yuyv", "bgra", "opaque_surface", "alpha_surface"]
FUSED_YUV = {"yuv420", "yuvj420", "nv12"}
OPAQUE_RGBA_ROUTE = {"yuv422", "yuv444", "uyvy", "yuyv", "opaque_surface"}


def expected_path(fmt, plan, fused_conversion=False):
    if not fused_conversion and fmt in FUSED_YUV:
        fmt = "yuv444"  # (the default: every Y'CbCr frame goes through the exact converter)
    if fmt in ("bgra", "alpha_surface"):  # an alpha channel: the four-channel builds
        return "wave_box" if plan.startswith("box_prereduced") else "wave_rgba"
    if plan in ("single_axis_h", "single_axis_v"):
        return "wave" if fmt in FUSED_YUV else "wave_rgba"
    if plan in ("box_prereduced", "box_prereduced_v_first"):
        return "wave_box"
    if plan == "box_prereduced_8":
        return "wave_box"
    if plan == "two_pass_wide":  # (the fused conversion keeps to column pairs: such a plan takes the node-texture route)
        return "wave_rgba"
    if fmt in FUSED_YUV:
        return "wave"
    return "wave_rgba"


def _smooth(rng, shape, lo=16, hi=235):
    """Camera-like bytes: smooth structure plus a little sensor noise (white noise is the per-stage tests' business)."""
    h, w = shape[:2]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    c = shape[2] if len(shape) == 3 else 1
    out = np.empty((h, w, c), np.float32)
    for k in range(c):
        fx, fy, ph = rng.uniform(0.01, 0.06), rng.uniform(0.01, 0.06), rng.uniform(0, 6.28)
        out[..., k] = (lo + hi) / 2 + (hi - lo) / 2.2 * np.sin(fx * xx + ph) * np.cos(fy * yy - ph) + rng.normal(0, 1.5, (h, w))
    out = np.clip(np.rint(out), lo, hi).astype(np.uint8)
    return out if len(shape) == 3 else out[..., 0]


def _source(ctx, hip, fmt, w, h, rng):
    """-> (device source, the oracle's node texture for it).  _source.kernel_node (the opt-in fused conversion's tests only): the node
    texture as that conversion quantises it — tests/convert_model.py, the kernel's FMA chain in numpy, itself held within one code of the
    oracle's with > 99.98 % of the codes identical."""
    _source.kernel_node = None
    y = _smooth(rng, (h, w))
    if fmt in ("yuv420", "yuvj420", "yuv422", "yuv444"):
        variant = {"yuv420": orc.YUV420, "yuvj420": orc.YUVJ420, "yuv422": orc.YUV422, "yuv444": orc.YUV444}[fmt]
        ch, cw = orc.chroma_shape(w, h, variant)
        u, v = _smooth(rng, (ch, cw), 40, 220), _smooth(rng, (ch, cw), 40, 220)
        f = {"yuv420": hip.FRAME_PLANAR_YUV420, "yuvj420": hip.FRAME_PLANAR_YUVJ420, "yuv422": hip.FRAME_PLANAR_YUV422, "yuv444": hip.FRAME_PLANAR_YUV444}[fmt]
        if fmt in FUSED_YUV:
            _source.kernel_node = convert_model.node_codes(y, u, v, full_range=(fmt == "yuvj420"))
        return ctx.frame(f, w, h, [y, u, v]), orc.planar_yuv_to_rgba(y, u, v, w, h, variant)
    if fmt == "nv12":
        uv = _smooth(rng, (h // 2, w // 2, 2), 40, 220)
        _source.kernel_node = convert_model.node_codes_nv12(y, uv)
        return ctx.frame(hip.FRAME_NV12, w, h, [y, uv]), orc.nv12_to_rgba(y, uv, w, h)
    if fmt in ("uyvy", "yuyv"):
        data = _smooth(rng, (h, w // 2, 4), 30, 225)
        order = 0 if fmt == "uyvy" else 1
        return ctx.frame(hip.FRAME_UYVY422 if order == 0 else hip.FRAME_YUYV422, w, h, [data]), orc.interleaved422_to_rgba(data, w, h, order)
    data = _smooth(rng, (h, w, 4), 0, 255)
    if fmt == "bgra":
        return ctx.frame(hip.FRAME_BGRA, w, h, [data]), orc.swizzle_to_rgba(data, w, h, 0)
    if fmt == "opaque_surface":
        data[..., 3] = 255
        surf = ctx.surface_from(data)
        surf.opaque = True  # SMR_SOURCE_OPAQUE_SURFACE
        return surf, data
    # premultiplied translucent texture
    a = data[..., 3:4].astype(np.uint16)
    data[..., :3] = (data[..., :3].astype(np.uint16) * a // 255).astype(np.uint8)
    return ctx.surface_from(data), data


def _cases():
    out = [pytest.param(f, p, False, id=f"{f}-{p}") for f in FORMATS for p in sorted(PLANS)]
    from smelter_amd import hip as h
    if not h.lab_build():  # the fused conversion (ingest implementation 5) exists in laboratory builds of the library only
        return out
    return out + [pytest.param(f, p, True, id=f"{f}-{p}-fused_conversion") for f in sorted(FUSED_YUV) for p in sorted(PLANS)]


@pytest.mark.parametrize("fmt,plan,fused_conversion", _cases())
def test_path_and_parity(hip, fmt, plan, fused_conversion):
    (sw, sh), (dw, dh) = PLANS[plan]
    ctx = hip.Context(0)
    try:
        if fused_conversion:
            _impl_or_skip(ctx, hip, hip.INGEST_MFMA_F16_FUSED)
        rng = np.random.default_rng(FORMATS.index(fmt) * 100 + sorted(PLANS).index(plan))
        src, node = _source(ctx, hip, fmt, sw, sh, rng)
        out = ctx.surface(dw, dh)
        layouts = [Layout(top=0, left=0, width=dw, height=dh, type=0, source_index=0, crop=(0, 0, sw, sh))]
        before = ctx.kernel_launches()
        ctx.render_layouts(layouts, [src], dw, dh, out_rgba=out)
        got = out.download()
        ran = {k: v - before[k] for k, v in ctx.kernel_launches().items()}
        want_path = expected_path(fmt, plan, fused_conversion)
        fast = {"wave": "ingest_wave", "wave_rgba": "ingest_wave_rgba", "wave_box": "ingest_wave_rgba", "valu": "ingest_valu",
                "general": "resample_general"}[want_path]
        assert ran[fast] == 1, (fmt, plan, want_path, ran)
        for other in ("ingest_wave", "ingest_wave_rgba", "ingest_valu", "resample_general"):
            if other != fast:
                assert ran[other] == 0, (fmt, plan, want_path, ran)
        # the converter runs exactly when the path needs a node texture of a frame
        needs_node = want_path in ("wave_rgba", "wave_box", "general") and fmt not in ("opaque_surface", "alpha_surface")
        assert ran["frame_to_rgba"] == (1 if needs_node else 0), (fmt, plan, ran)
        assert ran["compose_output"] + ran["apply_layouts"] == 1
        # parity: the oracle's resample of the oracle's node texture, composited 1:1 (the source covers the output)
        _, tile = orc.resample(node, (0.0, 0.0, float(sw), float(sh)), dw, dh)
        want = orc.apply_layouts(dw, dh, [Layout(top=0, left=0, width=dw, height=dh, type=0, source_index=0, crop=(0, 0, dw, dh))], [tile])
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        if want_path == "wave":
            # the opt-in fused colour conversion: stage by stage.  Its node texture is within one code of the oracle's (2e-5 .. 3e-4 of the bytes
            # differ); the tile is within 1 LSB — every byte — of the oracle's resample of THAT node texture.  (End to end a flipped bright texel
            # can show as 2..4 codes at a dark output: bounded here and, on white noise, in tests/test_gpu_fused.py.)
            node_k = _source.kernel_node
            dn = np.abs(node_k.astype(np.int16) - node.astype(np.int16))
            assert dn.max() <= 1 and (dn == 0).mean() >= 0.9997, (fmt, plan, int(dn.max()), float((dn == 0).mean()))  # (full range: ~2.5e-4 differ)
            _, tile_k = orc.resample(node_k, (0.0, 0.0, float(sw), float(sh)), dw, dh)
            want_k = orc.apply_layouts(dw, dh, [Layout(top=0, left=0, width=dw, height=dh, type=0, source_index=0, crop=(0, 0, dw, dh))], [tile_k])
            dk = np.abs(got.astype(np.int16) - want_k.astype(np.int16))
            assert dk.max() <= 1, (fmt, plan, int(dk.max()), int((dk > 1).sum()))
            assert d.max() <= 4 and (d > 1).sum() <= 2, (fmt, plan, int(d.max()), int((d > 1).sum()))  # end to end, for the record
        else:
            assert d.max() <= 1, (fmt, plan, int(d.max()))
        assert (d == 0).mean() >= 0.985, (fmt, plan, float((d == 0).mean()))
    finally:
        ctx.close()


@pytest.mark.parametrize("fmt", ["yuv444", "uyvy", "opaque_surface", "alpha_surface", "nv12"])
@pytest.mark.parametrize("seed", range(16))
def test_random_geometries_on_every_route(hip, fmt, seed):
    """Random source sizes, tile sizes and crops (two-pass, single-axis, box-pre-reduced, direct — whatever the planner makes of them)
    through whichever route the source takes: within 1 LSB of the oracle's plan on the oracle's node texture, and the same bytes on a
    second run (uninitialised padding must never reach a sum)."""
    rng = np.random.default_rng(5000 + 97 * FORMATS.index(fmt) + seed)
    sw, sh = int(rng.integers(12, 400)) * 2, int(rng.integers(8, 260)) * 2
    if seed % 4 == 0:
        dw, dh = int(rng.integers(4, 60)), int(rng.integers(4, 60))              # strong shrink: box levels
    elif seed % 4 == 1:
        dw, dh = int(rng.integers(8, 2 * sw)), sh                               # width only
    else:
        dw, dh = int(rng.integers(8, 2 * sw)), int(rng.integers(8, 2 * sh))
    crop = (0.0, 0.0, float(sw), float(sh))
    if seed % 3 == 0:  # a crop with integer and fractional offsets
        cl, ct = float(rng.integers(0, sw // 4)), float(rng.integers(0, sh // 4)) + (0.5 if seed % 2 else 0.0)
        crop = (ct, cl, float(rng.integers(sw // 2, sw - int(cl))), float(rng.integers(sh // 2, sh - int(ct) - 1)))
    ctx = hip.Context(0)
    try:
        src, node = _source(ctx, hip, fmt, sw, sh, rng)
        layouts = [Layout(top=0, left=0, width=dw, height=dh, type=0, source_index=0, crop=crop)]
        outs = []
        for _ in range(2):
            out = ctx.surface(dw, dh)
            ctx.render_layouts(layouts, [src], dw, dh, out_rgba=out)
            outs.append(out.download())
        assert np.array_equal(outs[0], outs[1])
        kind, tile = orc.resample(node, crop, dw, dh)
        srcs = [tile] if kind > 0 else [node]
        lay = [Layout(top=0, left=0, width=dw, height=dh, type=0, source_index=0, crop=(0, 0, dw, dh) if kind > 0 else crop)]
        want = orc.apply_layouts(dw, dh, lay, srcs)
        d = np.abs(outs[0].astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1, (fmt, seed, (sw, sh), (dw, dh), crop, int(d.max()), int((d > 1).sum()))
        assert (d == 0).mean() >= 0.98, (fmt, seed, float((d == 0).mean()))
    finally:
        ctx.close()


@pytest.mark.parametrize("fmt", ["yuv420", "yuvj420", "nv12"])
@pytest.mark.parametrize("plan", ["two_pass_h_first", "two_pass_v_first", "scale_2", "single_axis_h", "box_prereduced"])
def test_default_route_is_within_one_lsb_end_to_end_on_white_noise(hip, fmt, plan):
    """SMR_INGEST_AUTO: the frame goes through the exact converter into its node texture and the matrix-core kernel resamples that.
    White-noise planes (the content on which an inexact conversion's one-code flips would show as 2..4 codes end to end): every byte
    within 1 LSB of the oracle's converter + resampler."""
    (sw, sh), (dw, dh) = PLANS[plan]
    ctx = hip.Context(0)
    try:
        rng = np.random.default_rng(7 + sorted(PLANS).index(plan))
        y = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        if fmt == "nv12":
            uv = rng.integers(0, 256, (sh // 2, sw // 2, 2), dtype=np.uint8)
            src, node = ctx.frame(hip.FRAME_NV12, sw, sh, [y, uv]), orc.nv12_to_rgba(y, uv, sw, sh)
        else:
            u = rng.integers(0, 256, (sh // 2, sw // 2), dtype=np.uint8)
            v = rng.integers(0, 256, (sh // 2, sw // 2), dtype=np.uint8)
            variant = orc.YUV420 if fmt == "yuv420" else orc.YUVJ420
            src = ctx.frame(hip.FRAME_PLANAR_YUV420 if fmt == "yuv420" else hip.FRAME_PLANAR_YUVJ420, sw, sh, [y, u, v])
            node = orc.planar_yuv_to_rgba(y, u, v, sw, sh, variant)
        out = ctx.surface(dw, dh)
        before = ctx.kernel_launches()
        ctx.render_layouts([Layout(top=0, left=0, width=dw, height=dh, type=0, source_index=0, crop=(0, 0, sw, sh))], [src], dw, dh, out_rgba=out)
        got = out.download()
        ran = {k: v - before[k] for k, v in ctx.kernel_launches().items()}
        assert ran["frame_to_rgba"] == 1 and ran["ingest_wave_rgba"] == 1 and ran["ingest_wave"] == 0 and ran["resample_general"] == 0, (fmt, plan, ran)
        _, tile = orc.resample(node, (0.0, 0.0, float(sw), float(sh)), dw, dh)
        want = orc.apply_layouts(dw, dh, [Layout(top=0, left=0, width=dw, height=dh, type=0, source_index=0, crop=(0, 0, dw, dh))], [tile])
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1, (fmt, plan, int(d.max()), int((d > 1).sum()))
        assert (d == 0).mean() >= 0.97, (fmt, plan, float((d == 0).mean()))
    finally:
        ctx.close()


@pytest.mark.parametrize("fmt", ["yuv422", "yuv444", "uyvy"])
def test_shard_entry_point_takes_the_node_texture_route_too(hip, fmt):
    """smr_ingest_resample (the per-shard step of the multi-GPU path) on a frame no fused conversion reads: exact converter + matrix-core
    kernel, not the pass-per-launch resampler."""
    (sw, sh), (dw, dh) = PLANS["two_pass_h_first"]
    ctx = hip.Context(0)
    try:
        rng = np.random.default_rng(77)
        src, node = _source(ctx, hip, fmt, sw, sh, rng)
        tile = ctx.surface(dw, dh)
        before = ctx.kernel_launches()
        ctx.ingest_resample(src, (0.0, 0.0, float(sw), float(sh)), tile)
        ran = {k: v - before[k] for k, v in ctx.kernel_launches().items()}
        assert ran["frame_to_rgba"] == 1 and ran["ingest_wave_rgba"] == 1 and ran["resample_general"] == 0 and ran["ingest_valu"] == 0, (fmt, ran)
        _, want = orc.resample(node, (0.0, 0.0, float(sw), float(sh)), dw, dh)
        d = np.abs(tile.download().astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1, (fmt, int(d.max()))
    finally:
        ctx.close()
