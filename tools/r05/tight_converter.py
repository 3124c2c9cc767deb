"""The converter on wrapped frames with tight rows (k_yuv420_to_rgba_tight) against the same frames in library allocations (k_yuv420_to_rgba) and
against the general kernel they used to fall back to: microseconds per 1080p / 720p NV12 / 4K frame (one frame per launch, device time by events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from smelter_amd import hip

ctx = hip.Context(0)
rng = np.random.default_rng(3)


def planes(fmt, w, h):
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    c = rng.integers(0, 256, (h // 2, w // 2, 2), dtype=np.uint8)
    return [y, c.reshape(h // 2, w)] if fmt == hip.FRAME_NV12 else [y, np.ascontiguousarray(c[..., 0]), np.ascontiguousarray(c[..., 1])]


def wrapped(fmt, w, h, ps):
    bufs = []
    for p in ps:
        t = torch.from_numpy(np.ascontiguousarray(p)).cuda().contiguous()  # tight rows: pitch == bytes per row
        bufs.append(t)
    f = ctx.wrapped_frame(fmt, w, h, [(t.data_ptr(), t.shape[1]) for t in bufs])
    f.keep = bufs
    return f


def time_it(frame, node, n=300):
    for _ in range(20):
        ctx.frame_to_rgba(frame, node)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.frame_to_rgba(frame, node)
    ctx.sync()
    return (time.perf_counter() - t0) / n * 1e6


for name, fmt, w, h in (("1080p planar", hip.FRAME_PLANAR_YUV420, 1920, 1080), ("720p NV12", hip.FRAME_NV12, 1280, 720), ("4K NV12", hip.FRAME_NV12, 3840, 2160),
                        ("4K planar", hip.FRAME_PLANAR_YUV420, 3840, 2160)):
    ps = planes(fmt, w, h)
    owned = ctx.frame(fmt, w, h, ps)
    tight = wrapped(fmt, w, h, ps)
    node = ctx.surface(w, h)
    a = time_it(owned, node)
    before = ctx.kernel_launches()["frame_to_rgba_420"]
    b = time_it(tight, node)
    took_block = ctx.kernel_launches()["frame_to_rgba_420"] > before
    ctx.set_convert_impl(hip.CONVERT_GENERAL)
    g = time_it(tight, node)
    ctx.set_convert_impl(hip.CONVERT_AUTO)
    same = np.array_equal(ctx.frame_to_rgba(owned).download(), ctx.frame_to_rgba(tight).download())
    print(f"{name:14s} library allocation {a:7.1f} us   wrapped, tight rows {b:7.1f} us (block converter: {took_block})   general kernel {g:7.1f} us   same bytes: {same}")
