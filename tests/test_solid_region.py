"""The compositor kernels skip the fragment stage inside a layout's "solid region" (smr_layout_dev.h rect_solid_box,
host side smr_layout.hip: inset / corner / slack).  This checks the claim itself on the CPU: wherever the predicate
holds, the oracle's full fragment stage (rounded-rect SDF, masks, border, shadow smoothsteps — apply_layouts.wgsl:246-377)
returns the base colour bit for bit.  The predicate below restates the host logic; the GPU parity tests cover the kernels.
"""
import numpy as np
import pytest

from oracle import oracle as orc

F = np.float32


def half_int(v):
    v = F(v)
    return bool(v * F(2) == np.floor(v * F(2)) and abs(v) < F(32768))


def solid_params(left, top, width, height, radii, need, extra_exact=()):
    """(inset, corner) as smr_pack_layouts computes them."""
    exact = all(half_int(x) for x in (left, top, width, height, need, *radii, *extra_exact))
    slack = F(0) if exact else F(0.015625)
    inset = F(need) + slack
    rmax = max(F(r) for r in radii)
    corner = rmax + slack if rmax + slack > inset else F(0)
    return inset, corner


def solid_mask(W, H, left, top, width, height, inset, corner):
    left, top, width, height = F(left), F(top), F(width), F(height)
    fx = (np.arange(W, dtype=F) + F(0.5))[None, :]
    fy = (np.arange(H, dtype=F) + F(0.5))[:, None]
    inside = (left + inset <= fx) & (fx <= left + width - inset) & (top + inset <= fy) & (fy <= top + height - inset)
    if corner > 0:
        near_x = (fx < left + corner) | (fx > left + width - corner)
        near_y = (fy < top + corner) | (fy > top + height - corner)
        inside = inside & ~(near_x & near_y)
    return inside


def random_rect(rng, W, H, aligned):
    if aligned:
        q = lambda lo, hi: float(rng.integers(int(lo * 2), int(hi * 2) + 1)) / 2.0  # noqa: E731
    else:
        q = lambda lo, hi: float(F(rng.uniform(lo, hi)))  # noqa: E731
    w, h = q(8, W), q(8, H)
    return q(-8, W - w + 8), q(-8, H - h + 8), w, h, q


@pytest.mark.parametrize("seed", range(40))
def test_solid_region_is_exactly_the_base_colour(seed):
    rng = np.random.default_rng(seed)
    W, H = 96, 64
    aligned = seed % 2 == 0
    left, top, w, h, q = random_rect(rng, W, H, aligned)
    kind = int(rng.integers(0, 3))  # 0 plain colour, 1 colour with border, 2 shadow
    rmaxv = min(w, h) / 2
    radii = [q(0, rmaxv)] * 4 if rng.random() < 0.6 else [q(0, rmaxv) for _ in range(4)]
    color = [float(F(x)) for x in (0.2, 0.4, 0.6, 0.8)]
    masks = []
    for _ in range(int(rng.integers(0, 3))):
        ml, mt, mw, mh, _ = random_rect(rng, W, H, aligned)
        mr = [q(0, min(mw, mh) / 2)] * 4
        masks.append(orc.Mask(mr, mt, ml, mw, mh))
    L = orc.Layout(top=top, left=left, width=w, height=h, type=1, border_radius=radii, color=color, masks=masks)
    need = 0.5
    extra = ()
    if kind == 1:
        L.border_width = q(1, 6)
        L.border_color = [0.1, 0.1, 0.1, 1.0]
        need = float(F(L.border_width) + F(1.0))
    elif kind == 2:
        L.type = 2
        L.blur_radius = q(0, 12)
        need = max(float(F(L.blur_radius) / F(2)), 0.5)
        b = F(L.blur_radius)
        extra = (F(left) - b, F(top) - b, F(w) + F(2) * b, F(h) + F(2) * b)
    frag = orc.layout_fragments(W, H, L)
    inset, corner = solid_params(left, top, w, h, radii, need, extra)
    solid = solid_mask(W, H, left, top, w, h, inset, corner)
    for m in masks:
        mi, mc = solid_params(m.left, m.top, m.width, m.height, m.radius, 0.5)
        solid &= solid_mask(W, H, m.left, m.top, m.width, m.height, mi, mc)
    covered = ~np.isnan(frag[..., 0])
    assert not (solid & ~covered).any(), "a solid pixel must be covered by the quad"
    want = np.array(color, F)
    bad = solid & (frag != want[None, None, :]).any(axis=-1)
    assert not bad.any(), f"{int(bad.sum())} solid pixels differ from the base colour, first at {np.argwhere(bad)[0]}"
    # the region is not vacuous for rects of reasonable size
    if w > 2 * max(radii) + 4 * need + 4 and h > 4 * need + 4 and not masks:
        assert solid.any()


def test_rounded_rect_solid_region_is_large():
    """1280x720 tile with radius 24 at an integer position: everything but the four 24x24 corner squares is solid."""
    inset, corner = solid_params(0.0, 0.0, 1280.0, 720.0, [24.0] * 4, 0.5)
    assert (inset, corner) == (0.5, 24.0)
    solid = solid_mask(1280, 720, 0.0, 0.0, 1280.0, 720.0, inset, corner)
    assert int((~solid).sum()) == 4 * 24 * 24
