"""smr_surface_wrap on the device: frames and node textures over memory the caller owns (here: torch tensors), in place.  The contract of
include/smr.h — an allocation covers exactly pitch x h bytes, no kernel touches a byte outside it — with the pitches a decoder hands out:
tight rows (pitch == bytes per row: k_yuv420_to_rgba_tight, which requests nothing behind a window's last column — the plain kernel's spare
dword would lie past the pitch), rows with four spare bytes and 256-byte pitches (the plain block converter).  Every result equals the oracle's bytes /
the result of the same content in a surface the library allocated; canaries behind every wrapped plane stay intact."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
CANARY = 0xA7


@pytest.fixture(scope="module")
def hip():
    from smelter_amd import hip as h
    return h


@pytest.fixture(scope="module")
def ctx(hip):
    return hip.Context(0)


def _device_plane(torch, arr2d, pitch):
    """rows of `arr2d` (h x row bytes) at `pitch` in a device buffer of EXACTLY pitch * h bytes, followed by a canary block."""
    h, row = arr2d.shape
    buf = torch.full((pitch * h + 256,), CANARY, dtype=torch.uint8, device="cuda")
    view = buf[:pitch * h].view(h, pitch)
    view[:, :row] = torch.from_numpy(np.ascontiguousarray(arr2d)).cuda()
    return buf


def _canary_intact(buf, used):
    return bool((buf[used:] == CANARY).all().item())


@pytest.mark.parametrize("variant", ["420", "j420", "nv12"])
@pytest.mark.parametrize("w,h,slack", [(1920, 1080, 0), (1920, 1080, 4), (1280, 720, 0), (1280, 720, 4), (64, 36, 0), (260, 10, 0), (1922, 1082, 0), (3840, 16, 256)])
def test_wrapped_frames_convert_to_the_oracles_node_texture(ctx, hip, variant, w, h, slack):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(w * 7 + h + slack)
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    c = rng.integers(0, 256, (h // 2, w // 2, 2), dtype=np.uint8)
    pad = lambda row: (row + 3) // 4 * 4 + slack if slack < 256 else (row + 255) // 256 * 256  # (a wrapped pitch is a multiple of 4)
    if variant == "nv12":
        planes = [y, c.reshape(h // 2, w)]
        want = orc.nv12_to_rgba(y, c, w, h)
        fmt = hip.FRAME_NV12
    else:
        planes = [y, np.ascontiguousarray(c[..., 0]), np.ascontiguousarray(c[..., 1])]
        want = orc.planar_yuv_to_rgba(y, planes[1], planes[2], w, h, orc.YUVJ420 if variant == "j420" else orc.YUV420)
        fmt = hip.FRAME_PLANAR_YUVJ420 if variant == "j420" else hip.FRAME_PLANAR_YUV420
    bufs = [_device_plane(torch, p, pad(p.shape[1])) for p in planes]
    frame = ctx.wrapped_frame(fmt, w, h, [(b.data_ptr(), pad(p.shape[1])) for b, p in zip(bufs, planes)])
    before = ctx.kernel_launches()
    got = ctx.frame_to_rgba(frame).download()
    ran = {k: n - before[k] for k, n in ctx.kernel_launches().items()}
    assert np.array_equal(got, want), (variant, w, h, slack, int((got != want).sum()))
    for b, p in zip(bufs, planes):
        assert _canary_intact(b, pad(p.shape[1]) * p.shape[0])
    # every frame the block converter's geometry admits takes it — its plain kernel or, for rows that fill their pitch, its tight one
    assert ran["frame_to_rgba_420"] == (1 if w % 4 == 0 and w >= 8 else 0) and ran["frame_to_rgba"] == 1, ran


def test_a_wrapped_node_texture_is_resampled_and_composited_like_an_owned_one(ctx, hip):
    """A premultiplied RGBA8 texture in caller memory (16-byte aligned rows, the smallest pitch) as a layout's source through
    smr_render_layouts: the same output frame as with a surface of the library holding the same bytes."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(11)
    tw, th, W, H = 332, 186, 640, 360
    tex = rng.integers(0, 256, (th, tw, 4), dtype=np.uint8)
    tex[..., 3] = 255
    pitch = (tw * 4 + 15) // 16 * 16
    buf = _device_plane(torch, tex.reshape(th, tw * 4), pitch)
    wrapped = ctx.wrap(buf.data_ptr(), pitch, tw, th, hip.PX_RGBA8)
    owned = ctx.surface_from(tex)
    layouts = [orc.Layout(top=0.0, left=0.0, width=float(W), height=float(H), type=1, color=orc.color_to_shader((20, 40, 60, 255), True)),
               orc.Layout(top=10.0, left=16.0, width=498.0, height=279.0, type=0, source_index=0, crop=(0.0, 0.0, float(tw), float(th)), border_radius=(12.0,) * 4),
               orc.Layout(top=200.0, left=300.0, width=float(tw), height=float(th), type=0, source_index=0, crop=(0.0, 0.0, float(tw), float(th)))]
    outs = []
    for src in (wrapped, owned):
        out = ctx.frame(hip.FRAME_PLANAR_YUV420, W, H)
        ctx.render_layouts(layouts, [src], W, H, out)
        outs.append(out.download())
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert _canary_intact(buf, pitch * th)
    assert outs[0][0].std() > 10  # a picture, not a constant
