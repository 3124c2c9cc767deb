"""One frame of the DEFAULT ROUTE on the lane emulator, end to end: k_yuv420_to_rgba (RGB12 node) -> k_ingest_wave (the RGB12 class build)
-> k_classify_tiles + k_compose_output, chained the way smr_render_layouts chains them, against the oracle's whole pass sequence
(tests/refpipe.py: planar_yuv_to_rgba -> resample -> apply_layouts -> rgba_to_yuv): the contract of BASELINE.json — every byte of the output
frame within 1 LSB — on white noise and on camera-like content, for a small configs[2]-shaped scene (a grid of inputs scaled 1.5 : 1 into
tiles, labels over them).  The three stages are held to the oracle one by one in test_emu_convert / test_emu_wave / test_emu_compose; this
is their composition, on the CPU.  Test infrastructure only: the product has no CPU path."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import refpipe
from tests import test_emu_compose as tc
from tests import test_emu_convert as tcv
from tests import test_emu_wave as tw

P8 = C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def emus():
    # (the three emulator libraries, built by their own tests' fixtures)
    return tcv.emu.__wrapped__(), tw.emu.__wrapped__(), tc.emu.__wrapped__()


def _p(a):
    return a.ctypes.data_as(P8)


@pytest.mark.parametrize("content,variant", [("noise", "420"), ("noise", "j420"), ("noise", "nv12"), ("camera", "420")])
def test_a_frame_through_all_three_kernels_is_within_one_lsb_of_the_reference_pass_sequence(emus, content, variant):
    conv, wave, comp = emus
    iw, ih, cols, rows = 192, 108, 2, 2
    W, H = 256, 144
    tw_, th_ = W // cols, H // rows  # 128 x 72: 1.5 : 1, the benchmark's class
    rng = np.random.default_rng(hash((content, variant)) % 2**32)
    planes, nodes, tiles = [], [], []
    for i in range(cols * rows):
        y, c = tcv._content(content, iw, ih, rng)
        if variant == "nv12":
            uv = np.ascontiguousarray(c)
            planes.append((y, uv, uv))
            nodes.append(orc.nv12_to_rgba(y, uv, iw, ih))
        else:
            u, v = np.ascontiguousarray(c[..., 0]), np.ascontiguousarray(c[..., 1])
            planes.append((y, u, v))
            nodes.append(orc.planar_yuv_to_rgba(y, u, v, iw, ih, orc.YUVJ420 if variant == "j420" else orc.YUV420))
        # wave A1: the exact converter into an RGB12 node texture (bit for bit the oracle's node)
        packed = np.zeros((ih, 3 * iw), np.uint8)
        assert conv.emu_convert_420_run(_p(planes[-1][0]), _p(planes[-1][1]), _p(planes[-1][2]), iw, ih, 1 if variant == "nv12" else 0,
                                        1 if variant == "j420" else 0, 1, 3, _p(packed)) == 0
        assert np.array_equal(packed.reshape(ih, iw // 4, 3, 4).transpose(0, 1, 3, 2).reshape(ih, iw, 3), nodes[-1][..., :3])
        # wave A2: the matrix-core resampler on that node (its <4, 2> class build), 1.5 : 1 into the layout's rounded size
        crop = (0.0, 0.0, float(iw), float(ih))
        plan = orc.resample_plan(iw, ih, crop, tw_, th_)
        tile = np.zeros((th_, tw_, 4), np.uint8)
        info = (C.c_int * 4)()
        assert wave.emu_ingest_wave(_p(packed), _p(packed), _p(packed), iw, ih, 0, 6, plan.scale[0], plan.offset[0], plan.scale[1], plan.offset[1],
                                    _p(tile), tw_, th_, 2, 1, info) == 0
        tiles.append(tile)
    # the scene: background, the tiles 1:1 where the layouts put them, a translucent bordered label over each
    layouts_gpu = [orc.Layout(top=0.0, left=0.0, width=float(W), height=float(H), type=1, color=orc.color_to_shader((32, 32, 48, 255), True))]
    layouts_ref = list(layouts_gpu)
    for i in range(cols * rows):
        left, top = float((i % cols) * tw_), float((i // cols) * th_)
        layouts_gpu.append(orc.Layout(top=top, left=left, width=float(tw_), height=float(th_), type=0, source_index=i, crop=(0.0, 0.0, float(tw_), float(th_))))
        layouts_ref.append(orc.Layout(top=top, left=left, width=float(tw_), height=float(th_), type=0, source_index=i, crop=(0.0, 0.0, float(iw), float(ih))))
        label = orc.Layout(top=top + th_ - 22.0, left=left + 6.0, width=60.0, height=16.0, type=1, border_radius=(5.0,) * 4,
                           color=orc.color_to_shader((0, 0, 0, 150), True), border_width=1.0, border_color=orc.color_to_shader((255, 255, 255, 200), True))
        layouts_gpu.append(label)
        layouts_ref.append(label)
    # wave B on the resampled tiles (opaque surfaces)
    got, classes = tc.compose(comp, layouts_gpu, tiles, [2] * len(tiles), W, H, "planar")
    (wy, wu, wv), _rgba = refpipe.render_yuv420(layouts_ref, nodes, W, H)
    want = (wy, wu, wv)
    worst, exact = 0, []
    for g, w in zip(got, want):
        d = np.abs(g.astype(np.int16) - np.asarray(w).reshape(g.shape).astype(np.int16))
        worst = max(worst, int(d.max()))
        exact.append(float((d == 0).mean()))
    assert worst <= 1, (content, variant, worst, classes)
    assert min(exact) > (0.97 if content == "noise" else 0.995), exact   # the resampler's f16-pair arithmetic moves a byte here and there, never two codes
    assert classes["texture"] > 0 and classes["full"] > 0
