"""CPU model of the matrix-core resamplers' arithmetic (round 2's kernel, retired since: T pair, Wh pair, Wv feedback — the default below;
k_ingest_wave: pairs for all three) against the oracle: where the matrix-core formulation
of the two Lanczos passes spends the resampler's 1-LSB budget, variant by variant.  numpy only — the f32 accumulation order of the
matrix cores is not modelled (np.matmul's is used), so the counts are close to, not equal to, what tools/ingest_ab.py measures on
the device.

    T      sRGB-decoded texel of the u8 node texture:  "pair" = f16 hi + f16 lo (the kernel), "single" = one f16
    Wh     pass-1 weights (normalised, clamp-to-edge folded into the band):  "pair" (the kernel) or "single"
    Wv     pass-2 weights:  "feedback" = one f16 with the row's rounding residual put on the tap that absorbs it best (the kernel),
           "single" = plain f16, "pair"
    between the passes: f16, round to nearest even (the reference's Rgba16Float intermediate); then the sRGB threshold encode.

python tools/mfma_precision_sim.py [iw ih dw dh]        prints max LSB / % identical bytes per variant and content class"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402  (checker only)

F = np.float32
PI = F(3.14159265359)


def band(scale, offset, n_dst, n_src):
    """Dense [n_dst, n_src] f32 matrix of the normalised Lanczos3 weights of resample.wgsl:31-87 (the f32 sequence of
    lanczos_weights in smr_resample_dev.h), taps that clamp onto one texel summed."""
    scale, offset = F(scale), F(offset)
    k = max(scale, F(1.0))
    inv_k = F(1.0) / k
    support = F(3.0) * k
    taps = min(int(np.ceil(F(2.0) * support)) + 1, 64)
    W = np.zeros((n_dst, n_src), F)
    for o in range(n_dst):
        center = offset + (F(o) + F(0.5)) * scale - F(0.5)
        first = np.ceil(center - support).astype(F)
        x0 = (first - center) * inv_k
        s1, c1 = np.sin(PI * x0, dtype=F), np.cos(PI * x0, dtype=F)
        s3, c3 = np.sin(PI * x0 / F(3.0), dtype=F), np.cos(PI * x0 / F(3.0), dtype=F)
        sd1, cd1 = np.sin(PI * inv_k, dtype=F), np.cos(PI * inv_k, dtype=F)
        sd3, cd3 = np.sin(PI * inv_k / F(3.0), dtype=F), np.cos(PI * inv_k / F(3.0), dtype=F)
        w = np.zeros(taps, F)
        for t in range(taps):
            xx = x0 + F(t) * inv_k
            if abs(xx) < F(1e-5):
                w[t] = F(1.0)
            elif abs(xx) < F(3.0):
                w[t] = F(3.0) * s1 * s3 / (PI * PI * xx * xx)
            s1, c1 = s1 * cd1 + c1 * sd1, c1 * cd1 - s1 * sd1
            s3, c3 = s3 * cd3 + c3 * sd3, c3 * cd3 - s3 * sd3
        ws = F(0.0)
        for t in range(taps):
            ws = ws + w[t]
        for t in range(taps):
            idx = min(max(int(first) + t, 0), n_src - 1)
            W[o, idx] += w[t] / ws
    return W


def f16(x):
    return x.astype(np.float16).astype(F)


def weights(W, mode):
    """-> list of f32 matrices whose sum is applied (each entry an f16 value)."""
    hi = f16(W)
    if mode == "single":
        return [hi]
    if mode == "pair":
        return [hi, f16(W - hi)]
    q = hi.copy()  # "feedback": the row's residual goes to the tap whose f16 rounding absorbs it best
    for o in range(W.shape[0]):
        r = F(1.0) - q[o].sum(dtype=F)
        nz = np.nonzero(W[o])[0]
        cand = q[o, nz] + r
        best = nz[np.argmin(np.abs(f16(cand) - cand))]
        q[o, best] = f16(np.array([q[o, best] + r], F))[0]
    return [q]


def simulate(node_rgba, crop, dw, dh, t_mode="pair", wh_mode="pair", wv_mode="feedback"):
    """The kernel's arithmetic on an RGBA8 node texture (opaque): -> RGBA8 tile."""
    ih, iw = node_rgba.shape[:2]
    plan = orc.resample_plan(iw, ih, crop, dw, dh)
    assert plan.kind == 2 and tuple(plan.levels[:2]) == (0, 0), "two filtered axes, no box pre-reduction"
    slot = {plan.axis[0]: 0, plan.axis[1]: 1}
    Wh = band(plan.scale[slot[0]], plan.offset[slot[0]], dw, iw)
    Wv = band(plan.scale[slot[1]], plan.offset[slot[1]], dh, ih)
    dec = orc.srgb_decode_table().astype(F)
    thr = orc.srgb_threshold_table().astype(F)
    out = np.empty((dh, dw, 4), np.uint8)
    out[..., 3] = 255
    h_first = plan.axis[0] == 0
    for c in range(3):
        T = dec[node_rgba[..., c]]
        Tparts = [f16(T)] if t_mode == "single" else [f16(T), f16(T - f16(T))]
        if not h_first:  # the kernel runs a vertical-first plan on the transposed frame: same arithmetic, axes swapped
            Tparts = [p.T for p in Tparts]
        W1, W2 = (Wh, Wv) if h_first else (Wv, Wh)
        w1 = weights(W1, wh_mode)
        H = np.zeros((Tparts[0].shape[0], W1.shape[0]), F)
        for i, wp in enumerate(w1):
            for j, tp in enumerate(Tparts):
                if i == 1 and j == 1:
                    continue  # t_lo * w_lo is below the f32 accumulator's resolution: the kernel skips it too
                H += tp @ wp.T
        H = f16(H)
        O = np.zeros((W2.shape[0], H.shape[1]), F)
        for wp in weights(W2, wv_mode):
            O += wp @ H
        if not h_first:
            O = O.T
        out[..., c] = np.clip(np.searchsorted(thr[1:256], O, side="right"), 0, 255).astype(np.uint8)
    return out


def compare(a, b):
    d = np.abs(a[..., :3].astype(np.int16) - b[..., :3].astype(np.int16))
    return int(d.max()), 100.0 * float((d == 0).mean()), int((d > 1).sum())


def contents(iw, ih):
    from tests import scenes
    y, u, v = scenes.test_input(0, iw, ih, noise_seed=1234)
    yield "camera-like", orc.planar_yuv_to_rgba(y, u, v, iw, ih)
    rng = np.random.default_rng(50)
    yield "white noise", orc.planar_yuv_to_rgba(rng.integers(0, 256, (ih, iw), dtype=np.uint8), rng.integers(0, 256, (ih // 2, iw // 2), dtype=np.uint8),
                                                rng.integers(0, 256, (ih // 2, iw // 2), dtype=np.uint8), iw, ih)


def main():
    iw, ih, dw, dh = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (640, 360, 427, 240)
    crop = (0.0, 0.0, float(iw), float(ih))
    variants = [("kernel: T pair, Wh pair, Wv feedback", "pair", "pair", "feedback"), ("T single", "single", "pair", "feedback"),
                ("Wh single", "pair", "single", "feedback"), ("Wv plain f16", "pair", "pair", "single"), ("Wv pair", "pair", "pair", "pair"),
                ("everything single", "single", "single", "single")]
    for name, node in contents(iw, ih):
        _, want = orc.resample(node, crop, dw, dh, omp=True)
        print(f"{name}: {iw}x{ih} -> {dw}x{dh}")
        for label, t, wh, wv in variants:
            mx, ident, over = compare(simulate(node, crop, dw, dh, t, wh, wv), want)
            print(f"   {label:<40} max {mx}  identical {ident:8.4f} %  > 1 LSB: {over}")


if __name__ == "__main__":
    main()
