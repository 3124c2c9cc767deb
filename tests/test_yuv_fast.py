"""The output conversion's fast path (smelter_amd/csrc/smr_yuv_fast.h: three FMAs per value + a guard flag, the reference sequence behind the flag).

* CPU: tools/check_yuv_fast.cpp — compiled against THE header the kernels include — proves that an unflagged byte is the byte of
  rgba_to_yuv.wgsl:26-54's f32 sequence: luma over all 2^24 (R, G, B); chroma over the byte sums of a 2x2 block times every value the f32 mean can
  take (here every 16th red sum: `quick`; the whole domain, 6.7 G combinations per plane, runs in ~20 s with `tools/check_yuv_fast.cpp` alone).
* GPU: a picture made of the triples that sit closest to a code boundary — the ones the guard flags — through the compositor's copy tiles against
  the oracle's converter, byte for byte."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_path_never_disagrees_outside_its_guard_band(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "check_yuv_fast")
    r = subprocess.run([gxx, "-O2", "-fopenmp", "-ffp-contract=off", "-I", os.path.join(ROOT, "smelter_amd", "csrc"), "-o", exe,
                        os.path.join(ROOT, "tools", "check_yuv_fast.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK: the fast path never disagrees" in r.stdout, r.stdout[-2000:]
    assert "mismatches NOT flagged 0" in r.stdout and "2^24 triples" in r.stdout
    # the guard is narrow: a quarter of a thousandth of the values take the reference sequence
    flagged = float(r.stdout.split("flagged")[1].split("(")[1].split("%")[0])
    assert 0.01 < flagged < 0.05, flagged


def _near_boundary_triples(limit):
    """(R, G, B) whose luma lies closest to a code boundary in real arithmetic: a superset of what the guard flags."""
    ky = np.float64(np.float32(0.85882352941))
    k = [np.float64(np.float32(c)) * ky for c in (0.2126, 0.7152, 0.0722)]
    r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
    x = r * k[0] + g * k[1] + b * k[2] + np.float64(np.float32(16.0 / 255.0)) * 255.0 + 0.5
    d = np.abs(x - np.round(x)).ravel()
    idx = np.argsort(d)[:limit]
    return np.stack([r.ravel()[idx], g.ravel()[idx], b.ravel()[idx]], axis=-1).astype(np.uint8)


@pytest.mark.gpu
def test_pixels_on_code_boundaries_convert_like_the_reference_sequence():
    from oracle import oracle as orc
    from smelter_amd import hip
    from smelter_amd.scene import Layout
    W, H = 512, 256  # 131 072 pixels: the 131 072 triples nearest to a luma code boundary (the guard flags ~ 8 000 of all 2^24)
    px = np.full((H, W, 4), 255, np.uint8)
    px[..., :3] = _near_boundary_triples(W * H).reshape(H, W, 3)
    ctx = hip.Context(0)
    try:
        src = ctx.surface_from(px)
        src.opaque = True
        L = Layout(top=0.0, left=0.0, width=float(W), height=float(H), rotation_degrees=0.0, border_radius=[0.0] * 4, type=0, source_index=0,
                   color=[0.0] * 4, border_color=[0.0] * 4, border_width=0.0, crop=[0.0, 0.0, float(W), float(H)], blur_radius=0.0, masks=[])
        out = ctx.frame(hip.FRAME_PLANAR_YUV420, W, H)
        ctx.render_layouts([L], [src], W, H, out=out)
        got = out.download()
        want = orc.rgba_to_planar_yuv(px, orc.YUV420)
        for g, w_ in zip(got, want):
            assert np.array_equal(g, w_), int((g != w_).sum())
    finally:
        ctx.close()
