"""k_ingest_wave's index logic on the CPU: the kernel source run by the lane emulator of tests/emu (one host thread per lane, the
gfx950 instructions restated in C++) against the oracle's planar_yuv_to_rgba + resample.  Test infrastructure only — the product
has no CPU path; the -m gpu suite checks the real kernel on the device."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from tests import convert_model

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
ROOT = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
P8 = C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ to build the emulator with")
    from tests import emu_build
    lib = emu_build.build("smr_emu")
    h = C.CDLL(lib)
    h.emu_ingest_wave.argtypes = [P8, P8, P8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, P8, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.POINTER(C.c_int)]
    h.emu_check_encode.restype = C.c_longlong
    orc.build()
    if os.environ.get("SMR_EMU_GUARD"):  # the inner run of test_resampler_never_leaves_its_surfaces
        h.emu_set_guard(int(os.environ["SMR_EMU_GUARD"]), 1)
    return h


def test_one_gather_encode_is_the_step_function_on_every_float(emu):
    """k_ingest_wave's sRGB encode takes one table gather per value (bucket entry = code of the bucket's lowest x | offset of the one threshold inside
    it): equal to u8 = #{i : thr[i] <= x} on all 1.07 billion f32 values of [0, 1.5], on their negatives and at the infinities."""
    assert emu.emu_check_encode() == 0


def _planes(sw, sh, noise, seed):
    rng = np.random.default_rng(seed)
    if noise:
        return (rng.integers(0, 256, (sh, sw), dtype=np.uint8), rng.integers(0, 256, (sh // 2, sw // 2), dtype=np.uint8),
                rng.integers(0, 256, (sh // 2, sw // 2), dtype=np.uint8))
    xx, yy = np.meshgrid(np.arange(sw), np.arange(sh))
    y = (16 + (xx * 3 + yy * 2) % 200 + rng.integers(0, 8, (sh, sw))).astype(np.uint8)
    u = (100 + (np.arange(sw // 2)[None, :] + np.arange(sh // 2)[:, None]) % 60).astype(np.uint8)
    v = (140 - (np.arange(sw // 2)[None, :] * 2 + np.arange(sh // 2)[:, None]) % 70).astype(np.uint8)
    return y, u, v


def _p(a):
    return a.ctypes.data_as(P8)


# (source, tile, crop or None, pieces, specialised build, white noise, NV12, expected (NKS, NKS, KV) or None)
CASES = [
    ((96, 60), (64, 40), None, 1, 1, False, False, (4, 4, 2)),      # the benchmark's class: scale 1.5
    ((96, 60), (64, 40), None, 6, 0, True, False, (4, 4, 2)),       # same through the generic build, more pieces than tile rows
    ((192, 120), (128, 80), None, 4, 1, True, False, (4, 4, 2)),    # several pairs, pieces that cut between tiles
    ((130, 74), (86, 49), None, 2, 0, False, False, None),           # odd tile sizes: a last pair with one tile, partial tiles
    ((64, 36), (96, 54), None, 2, 0, True, False, None),             # upscale (7 taps, several tiles per chunk)
    ((256, 144), (128, 72), None, 2, 0, True, False, None),          # scale 2
    ((256, 144), (128, 72), None, 3, 1, True, False, (5, 5, 2)),     # ... and its class build (windows of 5 .. 8 k-steps, pass-2 windows of 2)
    ((256, 144), (128, 72), None, 2, 1, True, True, (5, 5, 2)),      # ... NV12
    ((384, 216), (128, 72), None, 2, 0, True, False, (7, 7, 3)),     # scale 3: the north-star target's class, generic build
    ((384, 216), (128, 72), None, 3, 1, True, False, (7, 7, 3)),     # ... and its class build (windows of <= 8 k-steps, pass-2 windows of 3)
    ((200, 120), (64, 40), (10.0, 20.0, 96.0, 60.0), 2, 1, True, False, (4, 4, 2)),  # crop: windows inside the frame
    ((96, 60), (64, 40), None, 2, 1, True, True, (4, 4, 2)),        # NV12
    ((16, 8), (12, 6), None, 1, 0, True, False, None),               # smaller than a chunk
]


@pytest.mark.parametrize("src,dst,crop,pieces,spec,noise,nv12,ks", CASES)
def test_emulated_kernel_matches_the_oracle(emu, src, dst, crop, pieces, spec, noise, nv12, ks):
    (sw, sh), (dw, dh) = src, dst
    y, u, v = _planes(sw, sh, noise, seed=sw * 131 + dh)
    crop = crop or (0.0, 0.0, float(sw), float(sh))
    plan = orc.resample_plan(sw, sh, crop, dw, dh)
    assert plan.kind == 2 and plan.levels == (0, 0) and tuple(plan.axis[:2]) == (0, 1), "the case must be a two-pass, horizontal-first plan"
    if nv12:
        node = orc.nv12_to_rgba(y, np.stack([u, v], axis=-1), sw, sh)
        uu = np.ascontiguousarray(np.stack([u, v], axis=-1))
    else:
        node = orc.planar_yuv_to_rgba(y, u, v, sw, sh)
        uu = u
    _, want = orc.resample(node, crop, dw, dh)
    got = np.zeros((dh, dw, 4), np.uint8)
    info = (C.c_int * 4)()
    rc = emu.emu_ingest_wave(_p(y), _p(uu), _p(v), sw, sh, 0, 1 if nv12 else 0, plan.scale[0], plan.offset[0], plan.scale[1], plan.offset[1], _p(got), dw, dh,
                             pieces, spec, info)
    assert rc == 0, (rc, list(info))
    if ks:
        assert tuple(info[:3]) == ks, list(info)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"{(d > 1).sum()} bytes off by more than 1 (max {d.max()})"
    assert (d == 0).mean() >= 0.999, (d == 0).mean()
    assert (got[..., 3] == 255).all()
    # per stage (tests/convert_model.py): against the oracle's resample of the node texture the kernel's own conversion produces
    node_k = convert_model.node_codes_nv12(y, uu) if nv12 else convert_model.node_codes(y, u, v)
    dn = np.abs(node_k.astype(np.int16) - node.astype(np.int16))
    assert dn.max() <= 1 and (dn == 0).mean() >= 0.9998, (dn.max(), (dn == 0).mean())
    _, want_k = orc.resample(node_k, crop, dw, dh)
    dk = np.abs(got.astype(np.int16) - want_k.astype(np.int16))
    assert dk.max() <= 1 and (dk == 0).mean() >= 0.9995, (dk.max(), (dk == 0).mean())


# (source, tile, crop, pieces, specialised build)
RGBA_CASES = [
    ((96, 60), (64, 40), None, 2, 1),          # the benchmark's class, zero fragments skipped
    ((96, 60), (64, 40), None, 3, 0),          # generic build
    ((130, 74), (86, 49), None, 2, 0),         # odd sizes: partial tiles, a width that is not a multiple of 4
    ((64, 36), (96, 54), None, 2, 0),          # upscale
    ((256, 144), (128, 72), None, 2, 0),       # scale 2
    ((256, 144), (128, 72), None, 3, 1),       # ... and its class build (<8, 2>: 5 k-steps, pass-2 windows of 2)
    ((384, 216), (128, 72), None, 3, 1),       # scale 3: the wide class build
    ((200, 120), (64, 40), (10.0, 20.0, 96.0, 60.0), 2, 1),  # crop
]


@pytest.mark.parametrize("src,dst,crop,pieces,spec", RGBA_CASES)
def test_emulated_kernel_on_an_rgba_node_texture(emu, src, dst, crop, pieces, spec):
    """The 8192 builds: the source is the RGBA8 node texture itself (what the exact converter wrote for a 4:2:2 / 4:4:4 / packed / BGRA
    frame), so the only arithmetic between the bytes and the oracle's resample is the matrix-core Lanczos: <= 1 LSB, no conversion term."""
    (sw, sh), (dw, dh) = src, dst
    rng = np.random.default_rng(sw * 7 + dh)
    node = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
    node[..., 3] = 255
    crop = crop or (0.0, 0.0, float(sw), float(sh))
    plan = orc.resample_plan(sw, sh, crop, dw, dh)
    assert plan.kind == 2 and plan.levels == (0, 0) and tuple(plan.axis[:2]) == (0, 1)
    _, want = orc.resample(node, crop, dw, dh)
    got = np.zeros((dh, dw, 4), np.uint8)
    info = (C.c_int * 4)()
    flat = np.ascontiguousarray(node)
    rc = emu.emu_ingest_wave(_p(flat), _p(flat), _p(flat), sw, sh, 0, 2, plan.scale[0], plan.offset[0], plan.scale[1], plan.offset[1], _p(got), dw, dh, pieces,
                             spec, info)
    assert rc == 0, (rc, list(info))
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"{(d > 1).sum()} bytes off by more than 1 (max {d.max()})"
    assert (d == 0).mean() >= 0.9995, (d == 0).mean()
    assert (got[..., 3] == 255).all()


@pytest.mark.parametrize("src,dst,crop,pieces,spec", [c for c in RGBA_CASES if c[0][0] % 4 == 0])
def test_emulated_kernel_on_an_rgb12_node_texture(emu, src, dst, crop, pieces, spec):
    """The 8192 + 131072 builds — the default route of 4:2:0 frames: the node texture as 12-byte groups of four pixels (R x 4, G x 4, B x 4:
    what k_yuv420_to_rgba writes for nodes only this kernel reads).  Same codes as the RGBA8 node, so the tile must be the RGBA8 build's tile
    bit for bit, and within 1 LSB of the oracle's resample."""
    (sw, sh), (dw, dh) = src, dst
    rng = np.random.default_rng(sw * 7 + dh)
    node = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
    node[..., 3] = 255
    crop = crop or (0.0, 0.0, float(sw), float(sh))
    plan = orc.resample_plan(sw, sh, crop, dw, dh)
    _, want = orc.resample(node, crop, dw, dh)
    # rows of groups: [R0 R1 R2 R3 G0 G1 G2 G3 B0 B1 B2 B3] per four pixels
    packed = np.ascontiguousarray(node[..., :3].reshape(sh, sw // 4, 4, 3).transpose(0, 1, 3, 2)).reshape(sh, 3 * sw)
    got, ref = np.zeros((dh, dw, 4), np.uint8), np.zeros((dh, dw, 4), np.uint8)
    info = (C.c_int * 4)()
    tail = (plan.scale[0], plan.offset[0], plan.scale[1], plan.offset[1])
    assert emu.emu_ingest_wave(_p(packed), _p(packed), _p(packed), sw, sh, 0, 6, *tail, _p(got), dw, dh, pieces, spec, info) == 0
    flat = np.ascontiguousarray(node)
    assert emu.emu_ingest_wave(_p(flat), _p(flat), _p(flat), sw, sh, 0, 2, *tail, _p(ref), dw, dh, pieces, spec, info) == 0
    assert np.array_equal(got, ref), int((got != ref).sum())
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1 and (d == 0).mean() >= 0.9995 and (got[..., 3] == 255).all(), (d.max(), (d == 0).mean())


# (source, tile, crop, pieces, white noise, NV12, full range)
PLANE_CASES = [
    ((96, 60), (64, 40), None, 2, True, False, False),        # the benchmark's class (scale 1.5): <4, 2>, zero fragments skipped
    ((192, 120), (128, 80), None, 5, True, False, False),     # several pairs, pieces that cut between tiles
    ((192, 120), (128, 80), None, 3, False, True, False),     # NV12
    ((96, 60), (64, 40), None, 1, True, False, True),         # full range (J420)
    ((200, 120), (64, 40), (10.0, 20.0, 96.0, 60.0), 2, True, False, False),  # crop: windows inside the frame
    ((256, 144), (128, 72), None, 3, True, False, False),     # scale 2: <8, 2>, two blocks per lane
    ((384, 216), (128, 72), None, 3, True, False, False),     # scale 3: <8, 3>, the north-star target's class
    ((384, 216), (128, 72), None, 2, True, True, False),      # ... NV12
    ((100, 62), (66, 41), None, 2, True, False, False),       # a width that is no multiple of 16, a height that is no multiple of 4
]


@pytest.mark.parametrize("src,dst,crop,pieces,noise,nv12,full", PLANE_CASES)
def test_emulated_kernel_on_the_frames_planes_converts_exactly(emu, src, dst, crop, pieces, noise, nv12, full):
    """The 262144 builds — the default route of 4:2:0 frames since round 6: the kernel reads the frame's planes and converts each chunk of its
    window in the wave with the exact converter's block arithmetic (smr_convert_420.h) into LDS.  The virtual node texture is the oracle's
    bit for bit, so — unlike the laboratory builds' one-code-per-stage conversion — there is NO conversion term: the tile is within 1 LSB of the
    oracle's resample of the oracle's node on white noise, and equal to the RGB12-node build's tile except where the two chunk grids (origin 0
    here, - 1 there) sum pass 2 in another order."""
    (sw, sh), (dw, dh) = src, dst
    y, u, v = _planes(sw, sh, noise, seed=sw * 17 + dh)
    crop = crop or (0.0, 0.0, float(sw), float(sh))
    plan = orc.resample_plan(sw, sh, crop, dw, dh)
    assert plan.kind == 2 and plan.levels == (0, 0) and tuple(plan.axis[:2]) == (0, 1)
    if nv12:
        uu = np.ascontiguousarray(np.stack([u, v], axis=-1))
        node = orc.nv12_to_rgba(y, uu, sw, sh)
    else:
        uu = u
        node = orc.planar_yuv_to_rgba(y, u, v, sw, sh, variant=orc.YUVJ420) if full else orc.planar_yuv_to_rgba(y, u, v, sw, sh)
    _, want = orc.resample(node, crop, dw, dh)
    got, ref = np.zeros((dh, dw, 4), np.uint8), np.zeros((dh, dw, 4), np.uint8)
    info = (C.c_int * 4)()
    tail = (plan.scale[0], plan.offset[0], plan.scale[1], plan.offset[1])
    rc = emu.emu_ingest_wave(_p(y), _p(uu), _p(v), sw, sh, 1 if full else 0, 8 if nv12 else 7, *tail, _p(got), dw, dh, pieces, 1, info)
    assert rc == 0, (rc, list(info))
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"{(d > 1).sum()} bytes off by more than 1 (max {d.max()})"
    assert (d == 0).mean() >= 0.9995, (d == 0).mean()
    assert (got[..., 3] == 255).all()
    if sw % 4 == 0:  # against the node route's kernel on the oracle's node as RGB12
        packed = np.ascontiguousarray(node[..., :3].reshape(sh, sw // 4, 4, 3).transpose(0, 1, 3, 2)).reshape(sh, 3 * sw)
        assert emu.emu_ingest_wave(_p(packed), _p(packed), _p(packed), sw, sh, 0, 6, *tail, _p(ref), dw, dh, pieces, 1, info) == 0
        dr = np.abs(got.astype(np.int16) - ref.astype(np.int16))
        assert dr.max() <= 1 and (dr == 0).mean() >= 0.9999, (dr.max(), (dr == 0).mean())


@pytest.mark.parametrize("src,dst,pieces", [((96, 60), (64, 40), 2), ((130, 74), (69, 40), 3), ((100, 56), (64, 36), 1)])
def test_emulated_kernel_on_a_box_reduced_rgba16f_texture(emu, src, dst, pieces):
    """The 8192 + 16384 build: the source is the RGBA16F texture downsample.wgsl leaves (linear light, alpha 1); its f16 texels are
    the matrix cores' operands as they are, so the only difference from the oracle's two passes is the f16-pair weights."""
    (sw, sh), (dw, dh) = src, dst
    rng = np.random.default_rng(sw + 3 * dh)
    lin = rng.random((sh, sw, 4), dtype=np.float32) ** 2.2
    lin[..., 3] = 1.0
    tex = lin.astype(np.float16).view(np.uint16)
    crop = (0.0, 0.0, float(sw), float(sh))
    plan = orc.resample_plan(sw, sh, crop, dw, dh)
    assert plan.kind == 2 and plan.levels == (0, 0) and tuple(plan.axis[:2]) == (0, 1)
    mid = orc.resample_pass(tex, orc.PX_RGBA16F, plan.axis[0], plan.scale[0], plan.offset[0], plan.perp_offset[0], orc.PX_RGBA16F, plan.mid[0], plan.mid[1])
    want = orc.resample_pass(mid, orc.PX_RGBA16F, plan.axis[1], plan.scale[1], plan.offset[1], plan.perp_offset[1], orc.PX_RGBA8_SRGB, dw, dh)
    got = np.zeros((dh, dw, 4), np.uint8)
    info = (C.c_int * 4)()
    flat = np.ascontiguousarray(tex).view(np.uint8)
    rc = emu.emu_ingest_wave(_p(flat), _p(flat), _p(flat), sw, sh, 0, 3, plan.scale[0], plan.offset[0], plan.scale[1], plan.offset[1], _p(got), dw, dh, pieces, 0,
                             info)
    assert rc == 0, (rc, list(info))
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"{(d > 1).sum()} bytes off by more than 1 (max {d.max()})"
    assert (d == 0).mean() >= 0.9995, (d == 0).mean()
    assert (got[..., 3] == 255).all()


# (source, tile width, crop (top, left, width, height) or None, pieces, source kind: 0 planar, 1 NV12, 2 RGBA8 node)
SA_CASES = [
    ((96, 60), 64, None, 2, 0),
    ((96, 60), 64, None, 3, 1),
    ((130, 74), 200, None, 2, 0),                      # upscale of the width only
    ((200, 120), 64, (20.0, 10.0, 96.0, 60.0), 2, 0),  # crop: the rows start at 20 (the perpendicular offset)
    ((96, 60), 64, None, 2, 2),
    ((256, 144), 100, (8.0, 0.0, 256.0, 100.0), 3, 2),
]


@pytest.mark.parametrize("src,dw,crop,pieces,kind", SA_CASES)
def test_emulated_single_axis_plan(emu, src, dw, crop, pieces, kind):
    """The 32768 builds: a plan with one pass (only the width changes).  Pass 1's f32 sums are encoded directly — no f16 rounding, no
    pass 2 — so the tile is the oracle's single resample pass within 1 LSB."""
    sw, sh = src
    crop = crop or (0.0, 0.0, float(sw), float(sh))
    dh = int(crop[3])
    plan = orc.resample_plan(sw, sh, crop, dw, dh)
    assert plan.kind == 1 and plan.axis[0] == 0 and plan.levels == (0, 0), plan
    if kind == 2:
        rng = np.random.default_rng(sw + dw)
        node = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
        node[..., 3] = 255
        a = b = c_ = np.ascontiguousarray(node)
        node_k = node
    else:
        y, u, v = _planes(sw, sh, True, seed=sw * 3 + dw)
        if kind == 1:
            uu = np.ascontiguousarray(np.stack([u, v], axis=-1))
            node, node_k = orc.nv12_to_rgba(y, uu, sw, sh), convert_model.node_codes_nv12(y, uu)
        else:
            uu = u
            node, node_k = orc.planar_yuv_to_rgba(y, u, v, sw, sh), convert_model.node_codes(y, u, v)
        a, b, c_ = y, uu, v
    want = orc.resample_pass(node_k, orc.PX_RGBA8_SRGB, 0, plan.scale[0], plan.offset[0], plan.perp_offset[0], orc.PX_RGBA8_SRGB, dw, dh)
    got = np.zeros((dh, dw, 4), np.uint8)
    info = (C.c_int * 4)()
    rc = emu.emu_ingest_wave(_p(a), _p(b), _p(c_), sw, sh, 0, kind, plan.scale[0], plan.offset[0], 1.0, float(plan.perp_offset[0]), _p(got), dw, dh, pieces, 2, info)
    assert rc == 0, (rc, list(info))
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"{(d > 1).sum()} bytes off by more than 1 (max {d.max()})"
    assert (d == 0).mean() >= 0.9995, (d == 0).mean()
    assert (got[..., 3] == 255).all()


@pytest.mark.parametrize("src,dst,crop,pieces,spec", RGBA_CASES)
def test_emulated_kernel_on_a_node_texture_with_alpha(emu, src, dst, crop, pieces, spec):
    """The 8192 + 65536 builds: a premultiplied RGBA8 node texture with an alpha channel (text, image, nested layout node, BGRA / ARGB
    frame): alpha is a fourth channel through both passes, linear, unorm8 on the way out — resample.wgsl on all four channels."""
    (sw, sh), (dw, dh) = src, dst
    rng = np.random.default_rng(sw * 11 + dh)
    node = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
    a = node[..., 3:4].astype(np.uint16)
    node[..., :3] = (node[..., :3].astype(np.uint16) * a // 255).astype(np.uint8)  # premultiplied
    crop = crop or (0.0, 0.0, float(sw), float(sh))
    plan = orc.resample_plan(sw, sh, crop, dw, dh)
    assert plan.kind == 2 and plan.levels == (0, 0) and tuple(plan.axis[:2]) == (0, 1)
    _, want = orc.resample(node, crop, dw, dh)
    got = np.zeros((dh, dw, 4), np.uint8)
    info = (C.c_int * 4)()
    flat = np.ascontiguousarray(node)
    rc = emu.emu_ingest_wave(_p(flat), _p(flat), _p(flat), sw, sh, 0, 4, plan.scale[0], plan.offset[0], plan.scale[1], plan.offset[1], _p(got), dw, dh, pieces,
                             spec, info)
    assert rc == 0, (rc, list(info))
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"{(d > 1).sum()} bytes off by more than 1 (max {d.max()})"
    assert (d == 0).mean() >= 0.9995, (d == 0).mean()
    assert not (got[..., 3] == 255).all()


def test_emulated_alpha_builds_of_the_single_axis_and_box_routes(emu):
    """+ 65536 on the 32768 (single-axis) and 16384 (RGBA16F) builds: alpha as a fourth channel there too."""
    # single-axis, premultiplied RGBA8 with alpha
    sw, sh, dw = 130, 74, 90
    rng = np.random.default_rng(5)
    node = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
    node[..., :3] = (node[..., :3].astype(np.uint16) * node[..., 3:4].astype(np.uint16) // 255).astype(np.uint8)
    plan = orc.resample_plan(sw, sh, (0.0, 0.0, float(sw), float(sh)), dw, sh)
    assert plan.kind == 1 and plan.axis[0] == 0
    want = orc.resample_pass(node, orc.PX_RGBA8_SRGB, 0, plan.scale[0], plan.offset[0], plan.perp_offset[0], orc.PX_RGBA8_SRGB, dw, sh)
    got = np.zeros((sh, dw, 4), np.uint8)
    info = (C.c_int * 4)()
    flat = np.ascontiguousarray(node)
    assert emu.emu_ingest_wave(_p(flat), _p(flat), _p(flat), sw, sh, 0, 4, plan.scale[0], plan.offset[0], 1.0, 0.0, _p(got), dw, sh, 2, 2, info) == 0
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1 and (d == 0).mean() >= 0.9995 and not (got[..., 3] == 255).all(), (d.max(), (d == 0).mean())
    # RGBA16F (linear light, premultiplied) with alpha, two passes
    sw, sh, dw, dh = 130, 74, 69, 40
    lin = rng.random((sh, sw, 4), dtype=np.float32)
    lin[..., :3] *= lin[..., 3:4]
    tex = lin.astype(np.float16).view(np.uint16)
    plan = orc.resample_plan(sw, sh, (0.0, 0.0, float(sw), float(sh)), dw, dh)
    assert plan.kind == 2 and tuple(plan.axis[:2]) == (0, 1)
    mid = orc.resample_pass(tex, orc.PX_RGBA16F, 0, plan.scale[0], plan.offset[0], 0, orc.PX_RGBA16F, plan.mid[0], plan.mid[1])
    want = orc.resample_pass(mid, orc.PX_RGBA16F, 1, plan.scale[1], plan.offset[1], 0, orc.PX_RGBA8_SRGB, dw, dh)
    got = np.zeros((dh, dw, 4), np.uint8)
    flat = np.ascontiguousarray(tex).view(np.uint8)
    assert emu.emu_ingest_wave(_p(flat), _p(flat), _p(flat), sw, sh, 0, 5, plan.scale[0], plan.offset[0], plan.scale[1], plan.offset[1], _p(got), dw, dh, 2, 0, info) == 0
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1 and (d == 0).mean() >= 0.9995 and not (got[..., 3] == 255).all(), (d.max(), (d == 0).mean())


@pytest.mark.parametrize("kind", [2, 3])
def test_emulated_single_tile_units_for_wide_windows(emu, kind):
    """A residual scale of 3.9 (what a box-pre-reduced plan can leave): a column pair's window would need 10 k-steps, more than the
    kernel holds, so the pass-1 band is built with one tile per unit (axis 4) and a wave works on one 16-column tile."""
    sw, sh, dw, dh = 500, 96, 128, 48
    rng = np.random.default_rng(77 + kind)
    crop = (0.0, 0.0, float(sw), float(sh))
    plan = orc.resample_plan(sw, sh, crop, dw, dh)
    assert plan.kind == 2 and plan.levels == (0, 0) and tuple(plan.axis[:2]) == (0, 1) and plan.scale[0] > 3.8
    if kind == 2:
        node = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
        node[..., 3] = 255
        _, want = orc.resample(node, crop, dw, dh)
        flat = np.ascontiguousarray(node)
    else:
        lin = rng.random((sh, sw, 4), dtype=np.float32) ** 2.2
        lin[..., 3] = 1.0
        tex = lin.astype(np.float16).view(np.uint16)
        mid = orc.resample_pass(tex, orc.PX_RGBA16F, 0, plan.scale[0], plan.offset[0], 0, orc.PX_RGBA16F, plan.mid[0], plan.mid[1])
        want = orc.resample_pass(mid, orc.PX_RGBA16F, 1, plan.scale[1], plan.offset[1], 0, orc.PX_RGBA8_SRGB, dw, dh)
        flat = np.ascontiguousarray(tex).view(np.uint8)
    got = np.zeros((dh, dw, 4), np.uint8)
    info = (C.c_int * 4)()
    rc = emu.emu_ingest_wave(_p(flat), _p(flat), _p(flat), sw, sh, 0, kind, plan.scale[0], plan.offset[0], plan.scale[1], plan.offset[1], _p(got), dw, dh, 2, 3, info)
    assert rc == 0, (rc, list(info))
    assert info[1] <= 8, list(info)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1 and (d == 0).mean() >= 0.9995, (d.max(), (d == 0).mean())


def test_resampler_never_leaves_its_surfaces(emu):
    """The memory contract of include/smr.h (smr_surface_wrap) for k_ingest_wave's node-texture builds (RGBA8 and RGB12 side by side, alpha, RGBA16F, single
    axis, single-tile units): those tests once more in child processes with node textures, tiles and weight bands of exactly pitch * h
    bytes — node and tile on the SMALLEST pitch can_fuse_wave_rgba lets through (16-byte multiples holding the row rounded up to four
    texels) — ending at (mode 1) or starting behind (mode 2) an unmapped page: a 16-byte load of the last texel group that reached past the
    row's pitch, a window clamped a row too late, a store beyond the tile would kill the child."""
    import sys
    if os.environ.get("SMR_EMU_GUARD"):
        pytest.skip("this is the inner run")
    children = {mode: subprocess.Popen([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider", "-k",
                                        "rgb12 or alpha or rgba16f or single_tile or single_axis"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                       env=dict(os.environ, SMR_EMU_GUARD=str(mode)), cwd=ROOT) for mode in (1, 2)}
    for mode, child in children.items():
        out, err = child.communicate(timeout=1500)
        assert child.returncode == 0, f"guard mode {mode}: rc {child.returncode} (-11 = the kernel left its surfaces)\n{out[-3000:]}\n{err[-2000:]}"
        assert " passed" in out
