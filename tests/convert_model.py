"""The node texture the matrix-core ingest kernels see (test infrastructure).

k_ingest_wave's fused-conversion builds (SMR_INGEST_MFMA_F16_FUSED, opt-in) never materialise the reference's RGBA8 node texture: m_convert_px (smelter_amd/csrc/smr_ingest_common.h)
turns Y'CbCr into the u8 code of every channel with the range expansion, the chroma up-sampling (exact, in 1/16 units) and the BT.709
matrix folded into one FMA chain per channel — the same values as planar_yuv_to_rgba.wgsl:35-58 up to f32 rounding, i.e. the same code
except where 255 R' + 0.5 lands within ~1e-4 of an integer (a few texels per hundred thousand, then one code off).  This module restates
that arithmetic in numpy, FMA for FMA, so that the tests can state the contract per stage:

    colour conversion   node codes within 1 LSB of the oracle's planar_yuv_to_rgba, > 99.98 % identical       (model vs oracle)
    resampling          the kernel's tile within 1 LSB of the oracle's resample of the node texture it saw     (kernel vs oracle(model))

The distinction only matters on adversarial content: a dark output pixel that is a cancelling sum of bright white-noise texels moves by
several codes when ONE of those texels flips by one code (linear-light resampling of 8-bit data is that sensitive, in the reference
too); on camera-like content the end-to-end comparison with the oracle is within 1 LSB as well and the tests check that directly."""
import numpy as np

f32 = np.float32


def _fma(a, b, c):
    # f32 fused multiply-add: the product of two f32 is exact in f64; one rounding to f64 (of the sum) before the one to f32 can only
    # differ from a true FMA when the f64 sum is a tie of the f32 grid to 29 bits — never observed on these value ranges
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(f32)


def conv_constants(full: bool):
    ys, y0 = (1.0, 0.0) if full else (255.0 / 219.0, 16.0)
    cs = 1.0 / 16.0 if full else 255.0 / (16.0 * 224.0)
    c0, half = (0.0, 16.0 * 127.5) if full else (16.0 * 16.0, 16.0 * 112.0)
    bias = 1024.0 + 256.0 + 0.5
    K = dict(ky=f32(ys), krv=f32(1.5748 * cs), kgu=f32(-0.1873 * cs), kgv=f32(-0.4681 * cs), kbu=f32(1.8556 * cs),
             cr=f32(bias - ys * y0 - 1.5748 * cs * (c0 + half)), cg=f32(bias - ys * y0 + (0.1873 + 0.4681) * cs * (c0 + half)),
             cb=f32(bias - ys * y0 - 1.8556 * cs * (c0 + half)),
             ylo=f32(0.0 if full else 16.0), yhi=f32(255.0 if full else 235.0), clo=f32(0.0 if full else 256.0), chi=f32(4080.0 if full else 3840.0))
    return K


def _chroma16(c):
    """Bilinear 4:2:0 -> luma grid in 1/16 units (weights 9, 3, 3, 1; clamp to edge), exact integers."""
    ch, cw = c.shape
    c = c.astype(np.int32)
    h, w = 2 * ch, 2 * cw
    yy, xx = np.arange(h), np.arange(w)
    # luma row r: even r = 2 p -> rows p - 1 (1), p (3); odd r = 2 p + 1 -> rows p (3), p + 1 (1); same along x
    r_lo = np.where(yy % 2 == 0, yy // 2 - 1, yy // 2)
    r_hi = r_lo + 1
    wr_lo = np.where(yy % 2 == 0, 1, 3)
    c_lo = np.where(xx % 2 == 0, xx // 2 - 1, xx // 2)
    c_hi = c_lo + 1
    wc_lo = np.where(xx % 2 == 0, 1, 3)
    r_lo, r_hi = np.clip(r_lo, 0, ch - 1), np.clip(r_hi, 0, ch - 1)
    c_lo, c_hi = np.clip(c_lo, 0, cw - 1), np.clip(c_hi, 0, cw - 1)
    rows = wr_lo[:, None] * c[r_lo] + (4 - wr_lo)[:, None] * c[r_hi]          # (h, cw)
    return wc_lo[None, :] * rows[:, c_lo] + (4 - wc_lo)[None, :] * rows[:, c_hi]  # (h, w)


def node_codes(y, u, v, full_range: bool = False):
    """(h, w, 4) u8: the RGBA8 node texture as the matrix-core kernels quantise it (alpha 255).  y: (h, w); u, v: (h/2, w/2)."""
    K = conv_constants(full_range)
    h, w = y.shape
    u16 = _chroma16(u)[:h, :w].astype(f32)
    v16 = _chroma16(v)[:h, :w].astype(f32)
    uf = np.clip(u16, K["clo"], K["chi"])
    vf = np.clip(v16, K["clo"], K["chi"])
    yf = np.clip(y.astype(f32), K["ylo"], K["yhi"])
    r = _fma(yf, K["ky"], _fma(vf, K["krv"], np.full_like(yf, K["cr"])))
    g = _fma(yf, K["ky"], _fma(uf, K["kgu"], _fma(vf, K["kgv"], np.full_like(yf, K["cg"]))))
    b = _fma(yf, K["ky"], _fma(uf, K["kbu"], np.full_like(yf, K["cb"])))
    out = np.empty((h, w, 4), np.uint8)
    for i, ch in enumerate((r, g, b)):
        idx = (ch.view(np.uint32) >> 13) & 0x3FF  # integer part of ch - 1024 = code + 256
        out[..., i] = np.clip(idx.astype(np.int32) - 256, 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out


def node_codes_nv12(y, uv, full_range: bool = False):
    return node_codes(y, np.ascontiguousarray(uv[..., 0]), np.ascontiguousarray(uv[..., 1]), full_range)
