"""The 4:2:0 input converter (smelter_amd/csrc/smr_convert_420.h, k_yuv420_to_rgba's per-thread 4 x 4 block) compiled for the CPU by
tests/emu/emu_convert.cpp, against the oracle's planar_yuv_to_rgba / nv12_to_rgba: EVERY BYTE EQUAL — the node texture the product
resamples is the reference's node texture.  Test infrastructure only: the product has no CPU path; tests/test_gpu_parity.py holds the
kernel itself to the WGSL-pass kernels and to the oracle on the device."""
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


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ to build the emulator with")
    from tests import emu_build
    lib = emu_build.build("smr_emu_convert", "emu_convert.cpp", ("smr_convert_420.h", "smr_convert_dev.h"))
    h = C.CDLL(lib)
    h.emu_convert_420.argtypes = [P8, P8, P8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P8]
    h.emu_convert_420_run.argtypes = [P8, P8, P8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P8]
    PP8, PI = C.POINTER(P8), C.POINTER(C.c_int)
    h.emu_convert_420_shares.argtypes = [C.c_int, PP8, PP8, PP8, PI, PI, C.c_int, PI, PI, C.c_int, PP8]
    orc.build()
    if os.environ.get("SMR_EMU_GUARD"):  # the inner run of test_converter_never_leaves_its_planes: planes of exactly pitch * h bytes against an unmapped page
        h.emu_set_guard(int(os.environ["SMR_EMU_GUARD"]), 1)
    if os.environ.get("SMR_EMU_TIGHT"):  # ... of test_tight_builds_...: k_yuv420_to_rgba_tight's code, chroma planes whose pitch is the row itself
        h.emu_set_tight(1)
    return h


def _p(a):
    return a.ctypes.data_as(P8)


def _content(kind, w, h, rng):
    cw, ch = w // 2, h // 2
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (ch, cw, 2), dtype=np.uint8)
    if kind == "extremes":  # the range clamps' corners and their neighbours
        return (rng.choice(np.array([0, 1, 15, 16, 17, 234, 235, 236, 254, 255], np.uint8), (h, w)),
                rng.choice(np.array([0, 15, 16, 17, 127, 128, 129, 239, 240, 241, 255], np.uint8), (ch, cw, 2)))
    xx, yy = np.meshgrid(np.arange(w), np.arange(h))  # camera-like: smooth luma, slow chroma
    y = (16 + (xx * 3 + yy * 2) % 200 + rng.integers(0, 8, (h, w))).astype(np.uint8)
    c = np.stack([(100 + (np.arange(cw)[None, :] + np.arange(ch)[:, None]) % 60), (140 - (np.arange(cw)[None, :] * 2 + np.arange(ch)[:, None]) % 70)], -1).astype(np.uint8)
    return y, c


@pytest.mark.parametrize("w,h", [(8, 2), (8, 6), (12, 4), (64, 36), (132, 74), (256, 18), (1920, 16)])
@pytest.mark.parametrize("variant", ["420", "j420", "nv12"])
@pytest.mark.parametrize("kind", ["noise", "extremes", "camera"])
def test_block_converter_is_the_oracle_bit_for_bit(emu, w, h, variant, kind):
    rng = np.random.default_rng(hash((w, h, variant, kind)) % 2**32)
    y, c = _content(kind, w, h, rng)
    got = np.zeros((h, w, 4), np.uint8)
    if variant == "nv12":
        c = np.ascontiguousarray(c)
        assert emu.emu_convert_420(_p(y), _p(c), _p(c), w, h, 1, 0, 0, _p(got)) == 0
        want = orc.nv12_to_rgba(y, c, w, h)
        u = v = c
    else:
        u, v = np.ascontiguousarray(c[..., 0]), np.ascontiguousarray(c[..., 1])
        full = 1 if variant == "j420" else 0
        assert emu.emu_convert_420(_p(y), _p(u), _p(v), w, h, 0, full, 0, _p(got)) == 0
        want = orc.planar_yuv_to_rgba(y, u, v, w, h, orc.YUVJ420 if full else orc.YUV420)
    assert np.array_equal(got, want), (variant, kind, w, h, int((got != want).sum()), np.argwhere(got != want)[:4].tolist())
    # the node as RGB12 (what the matrix-core resampler reads on the default route): the same codes, 12 bytes per four pixels
    packed = np.zeros((h, 3 * w), np.uint8)
    assert emu.emu_convert_420(_p(y), _p(u), _p(v), w, h, 1 if variant == "nv12" else 0, 1 if variant == "j420" else 0, 1, _p(packed)) == 0
    assert np.array_equal(packed.reshape(h, w // 4, 3, 4).transpose(0, 1, 3, 2).reshape(h, w, 3), want[..., :3]) and (want[..., 3] == 255).all()


def test_every_luma_byte_against_every_chroma_pair(emu):
    """All 256 luma bytes x a dense sweep of chroma pairs on flat patches (the interpolated chroma equals the byte: (a/4 + 3a/4 is exact up
    to the lerp's own rounding, which the oracle performs too), through both ranges: every byte equal."""
    w, h = 256, 8
    y = np.tile(np.arange(256, dtype=np.uint8), (h, 1))
    for full in (0, 1):
        for cu in range(0, 256, 5):
            u = np.full((h // 2, w // 2), cu, np.uint8)
            for cv in (0, 16, 17, 77, 128, 129, 201, 240, 255):
                v = np.full((h // 2, w // 2), cv, np.uint8)
                got = np.zeros((h, w, 4), np.uint8)
                assert emu.emu_convert_420(_p(y), _p(u), _p(v), w, h, 0, full, 0, _p(got)) == 0
                want = orc.planar_yuv_to_rgba(y, u, v, w, h, orc.YUVJ420 if full else orc.YUV420)
                assert np.array_equal(got, want), (full, cu, cv)


@pytest.mark.parametrize("w,h", [(8, 2), (8, 6), (12, 4), (64, 36), (132, 74), (256, 130), (1920, 22)])
@pytest.mark.parametrize("variant", ["420", "j420", "nv12"])
@pytest.mark.parametrize("nb", [1, 2, 3, 4, 7])
def test_runs_of_blocks_equal_the_oracle_bit_for_bit(emu, w, h, variant, nb):
    """cv420_run — what a wave of k_yuv420_to_rgba executes: a vertical run of nb blocks that keeps the two chroma window rows neighbouring
    blocks share and requests block P + 1 before it computes block P — writes the oracle's bytes whatever the run length, at every frame
    height (runs that end inside a block, heights of 2 mod 4), RGBA8 and RGB12."""
    rng = np.random.default_rng(hash((w, h, variant, nb)) % 2**32)
    y, c = _content("noise", w, h, rng)
    nv, full = (1 if variant == "nv12" else 0), (1 if variant == "j420" else 0)
    if nv:
        u = v = np.ascontiguousarray(c)
        want = orc.nv12_to_rgba(y, u, w, h)
    else:
        u, v = np.ascontiguousarray(c[..., 0]), np.ascontiguousarray(c[..., 1])
        want = orc.planar_yuv_to_rgba(y, u, v, w, h, orc.YUVJ420 if full else orc.YUV420)
    got = np.zeros((h, w, 4), np.uint8)
    assert emu.emu_convert_420_run(_p(y), _p(u), _p(v), w, h, nv, full, 0, nb, _p(got)) == 0
    assert np.array_equal(got, want), (variant, nb, w, h, int((got != want).sum()), np.argwhere(got != want)[:4].tolist())
    packed = np.zeros((h, 3 * w), np.uint8)
    assert emu.emu_convert_420_run(_p(y), _p(u), _p(v), w, h, nv, full, 1, nb, _p(packed)) == 0
    assert np.array_equal(packed.reshape(h, w // 4, 3, 4).transpose(0, 1, 3, 2).reshape(h, w, 3), want[..., :3])


@pytest.mark.parametrize("nv", [0, 1])
@pytest.mark.parametrize("waves", [1, 3, 8, 9, 13, 64, 1000])
def test_a_launch_cut_into_equal_shares_writes_the_oracles_bytes(emu, nv, waves):
    """k_yuv420_to_rgba's partition (cv420_plan + cv420_share): frames of different sizes, ranges and node formats in one launch of `waves`
    workgroups — bands of block rows dealt to the eight XCDs, every XCD's units cut into equal contiguous shares per wave: shares that start
    and end anywhere inside a column of blocks, straddle column blocks, bands and frames, more waves than units, XCDs with one workgroup and
    with many — every byte of every frame the oracle's (and none written twice differently: a unit missed or doubled would show)."""
    rng = np.random.default_rng(17 * waves + nv)
    sizes = [(264, 38), (8, 2), (516, 26), (64, 130), (20, 6)]
    fulls = [0, 0, 0, 0, 0] if nv else [0, 1, 0, 1, 1]
    rgb12 = [1, 0, 0, 1, 0]
    ys, us, vs, outs, wants = [], [], [], [], []
    for (w, h), full, r12 in zip(sizes, fulls, rgb12):
        y, c = _content("noise", w, h, rng)
        if nv:
            u = v = np.ascontiguousarray(c)
            want = orc.nv12_to_rgba(y, u, w, h)
        else:
            u, v = np.ascontiguousarray(c[..., 0]), np.ascontiguousarray(c[..., 1])
            want = orc.planar_yuv_to_rgba(y, u, v, w, h, orc.YUVJ420 if full else orc.YUV420)
        ys.append(y); us.append(u); vs.append(v); wants.append(want)
        outs.append(np.zeros((h, 3 * w), np.uint8) if r12 else np.zeros((h, w, 4), np.uint8))
    n = len(sizes)
    arr = lambda xs: (P8 * n)(*[_p(x) for x in xs])
    ints = lambda xs: (C.c_int * n)(*xs)
    assert emu.emu_convert_420_shares(n, arr(ys), arr(us), arr(vs), ints([s[0] for s in sizes]), ints([s[1] for s in sizes]), nv, ints(fulls),
                                      ints(rgb12), waves, arr(outs)) == 0
    for (w, h), r12, got, want in zip(sizes, rgb12, outs, wants):
        if r12:
            got = np.concatenate([got.reshape(h, w // 4, 3, 4).transpose(0, 1, 3, 2).reshape(h, w, 3), np.full((h, w, 1), 255, np.uint8)], -1)
        assert np.array_equal(got, want), (w, h, waves, int((got != want).sum()))


@pytest.mark.parametrize("mode", [1, 2])
def test_converter_never_leaves_its_planes(emu, mode):
    """The memory contract of include/smr.h (smr_surface_wrap): an allocation covers pitch * h bytes and no kernel touches a byte outside it.
    The tests above once more in a child process, with every plane and node texture exactly pitch * h bytes — on the SMALLEST pitch the host
    code lets through (conv_420_ok: the reach of the last block's dword loads) — ending at (mode 1) or starting behind (mode 2) an unmapped
    page: a load or store outside the allocation kills the child.  Prefetches included: a run requests nothing behind its last block."""
    if os.environ.get("SMR_EMU_GUARD") or os.environ.get("SMR_EMU_TIGHT"):
        pytest.skip("this is the inner run")
    env = dict(os.environ, SMR_EMU_GUARD=str(mode))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider", "-k",
                        "block_converter or runs_of_blocks or equal_shares"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
    assert r.returncode == 0, f"guard mode {mode}: rc {r.returncode} (-11 = a kernel left its planes)\n{r.stdout[-3000:]}\n{r.stderr[-2000:]}"
    assert " passed" in r.stdout


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_tight_builds_write_the_same_bytes_and_request_nothing_behind_a_row(emu, mode):
    """k_yuv420_to_rgba_tight — the kernel of wrapped frames whose chroma rows fill their pitch (a decoder's tight surfaces): the dwords behind a
    window's last column are not requested.  The tests above once more with the TIGHT builds: every byte the oracle's (mode 0), and with each
    plane an allocation of exactly pitch * h bytes at pitch = the row's bytes rounded up to a dword, ending at / starting behind an unmapped
    page (modes 1, 2) — where the plain kernel's reach would fault."""
    if os.environ.get("SMR_EMU_GUARD") or os.environ.get("SMR_EMU_TIGHT"):
        pytest.skip("this is the inner run")
    env = dict(os.environ, SMR_EMU_TIGHT="1")
    if mode:
        env["SMR_EMU_GUARD"] = str(mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider", "-k",
                        "block_converter or runs_of_blocks or equal_shares"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
    assert r.returncode == 0, f"tight, guard mode {mode}: rc {r.returncode} (-11 = the kernel left its planes)\n{r.stdout[-3000:]}\n{r.stderr[-2000:]}"
    assert " passed" in r.stdout
