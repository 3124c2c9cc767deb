"""tests/convert_model.py (the matrix-core ingest kernels' colour conversion restated in numpy) against the oracle's planar_yuv_to_rgba:
the conversion stage of the fused kernels is within 1 LSB of the reference's pass on every content class, with nearly all codes equal."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import convert_model, scenes


@pytest.mark.parametrize("content", ["camera", "noise", "extremes"])
@pytest.mark.parametrize("full_range", [False, True])
def test_matrix_core_node_texture_matches_the_oracle(content, full_range):
    orc.build()
    w, h = 640, 360
    rng = np.random.default_rng(11)
    if content == "camera":
        y, u, v = scenes.test_input(2, w, h, noise_seed=5)
    elif content == "noise":
        y = rng.integers(0, 256, (h, w), dtype=np.uint8)
        u = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
        v = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
    else:  # out-of-range codes everywhere: the clamps of planar_yuv_to_rgba.wgsl:45-57 act
        y = rng.choice(np.array([0, 1, 15, 16, 17, 234, 235, 236, 254, 255], np.uint8), (h, w))
        u = rng.choice(np.array([0, 15, 16, 128, 240, 241, 255], np.uint8), (h // 2, w // 2))
        v = rng.choice(np.array([0, 15, 16, 128, 240, 241, 255], np.uint8), (h // 2, w // 2))
    want = orc.planar_yuv_to_rgba(y, u, v, w, h, orc.YUVJ420 if full_range else orc.YUV420)
    got = convert_model.node_codes(y, u, v, full_range)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, d.max()
    assert (d == 0).mean() >= 0.9998, (d == 0).mean()
