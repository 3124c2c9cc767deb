"""Where a wave of k_yuv420_to_rgba spends its life: reads the cycle stamps of the timing build (tools/variant.sh timing -DCV_TIMING;
run with SMR_LIB=smelter_amd/variants/libsmr_hip.timing.so) for one launch over 8 x 1920x1080 frames (or --4k) and prints, per phase,
the distribution over waves, when waves start and end relative to the launch's first wave, and how many waves are alive over time.
    SMR_LIB=... [SMR_CONVERT_WG_PER_CU=k] python tools/r05/conv_timing.py [--4k]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smelter_amd import _ffi, hip  # noqa: E402
from tests import scenes  # noqa: E402


def pct(a, qs=(5, 50, 95)):
    return " / ".join(f"{np.percentile(a, q):8.0f}" for q in qs)


def main():
    four_k = "--4k" in sys.argv
    iw, ih = (3840, 2160) if four_k else (1920, 1080)
    n, dw, dh = 8, 1280, 720
    ctx = hip.Context(0)
    planes = [scenes.test_input(i, iw, ih, noise_seed=1234 + i) for i in range(n)]
    frames = [ctx.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(p)) for p in planes]
    crops = [(0.0, 0.0, float(iw), float(ih))] * n
    dsts = [ctx.surface(dw, dh) for _ in range(n)]
    for _ in range(5):
        ctx.ingest_resample_batch(frames, crops, dsts)
    ctx.sync()
    lib = _ffi.load()
    fn = lib.smr_debug_convert_stamps
    fn.argtypes = [C.POINTER(C.c_uint64), C.c_uint]
    per_cu = max(1, min(6, int(os.environ.get("SMR_CONVERT_WG_PER_CU", "4"))))
    tasks = n * ((iw + 255) // 256) * ((ih + 3) // 4)
    nb = tasks / (256 * per_cu * 4)
    blocks = min(256 * per_cu, (tasks + 3) // 4)
    nw = min(blocks * 4, 32768)
    buf = np.zeros((nw, 10), np.uint64)
    assert fn(buf.ctypes.data_as(C.POINTER(C.c_uint64)), nw) == 0
    live = buf[buf[:, 6] != 0].astype(np.int64)
    print(f"{iw}x{ih} x {n}, share {nb:.2f} units per wave, {per_cu} workgroups per CU: {tasks} units, {blocks} workgroups, {nw} waves stamped, {len(live)} with work")
    t0 = live[:, 0]
    names = ["tables + barrier (+ first ticket)", "first task: loads arrive", "   window converted (gathers + lerps)", "   first block's rows", "every other block and task", "stores acknowledged"]
    for i, nm in enumerate(names):
        d = live[:, i + 1] - live[:, i]
        print(f"   {nm:38s} cycles p5 / p50 / p95: {pct(d)}   mean {d.mean():8.0f}")
    life = live[:, 6] - live[:, 0]
    print(f"   {'wave lifetime':38s} cycles p5 / p50 / p95: {pct(life)}   mean {life.mean():8.0f}")
    rt0, rt1 = live[:, 7], live[:, 8]  # 100 MHz, coherent across the chip (the shader clock counters are per XCD)
    base = rt0.min()
    print(f"   launch (realtime counter, us after the first wave's start): last wave start {(rt0.max() - base) / 100.0:.2f}, last wave end {(rt1.max() - base) / 100.0:.2f}")
    dur = np.maximum(rt1 - rt0, 1)
    clk = life / (dur / 100.0) / 1000.0
    print(f"   shader clock over a wave's life (cycles / realtime), GHz p5 / p50 / p95: " + " / ".join(f"{np.percentile(clk, q):.2f}" for q in (5, 50, 95)))
    print(f"   wave lifetime us p5 / p50 / p95 / max: " + " / ".join(f"{v / 100.0:.2f}" for v in (np.percentile(dur, 5), np.percentile(dur, 50), np.percentile(dur, 95), dur.max())))
    edges = np.linspace(base, rt1.max(), 25)
    print("   waves alive at 24 equally spaced instants:", [int(((rt0 <= e) & (rt1 > e)).sum()) for e in edges])
    # placement: waves per (XCC, SE, CU, SIMD) from HW_ID (gfx9: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 [+ higher bits])
    hw = live[:, 9]
    xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
    cu_key = (xcc << 16) | (((hwid >> 13) & 0x7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xf)
    _, per_cu = np.unique(cu_key, return_counts=True)
    print(f"   distinct CUs seen {len(per_cu)}; working waves per CU min / median / max: {per_cu.min()} / {int(np.median(per_cu))} / {per_cu.max()}")
    ends_by_cu = {}
    for k, e in zip(cu_key, rt1):
        ends_by_cu[k] = max(ends_by_cu.get(k, 0), e)
    ce = (np.array(list(ends_by_cu.values())) - base) / 100.0
    print(f"   per-CU finish time us p5 / p50 / p95 / max: {np.percentile(ce, 5):.2f} / {np.percentile(ce, 50):.2f} / {np.percentile(ce, 95):.2f} / {ce.max():.2f}")
    ctx.close()


if __name__ == "__main__":
    main()
