#!/usr/bin/env python3
"""Host-side emulation of k_compose_output's tile classification for the bench scene (how many 128x16 tiles take the
general path, and with which layer sets).  Mirrors smr_layout.hip (bbox / inset / corner) and smr_fused_compose.h."""
import collections, math, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import numpy as np

F = np.float32


def half_int(v):
    v = F(v)
    return bool(v * F(2) == np.floor(v * F(2)) and abs(v) < F(32768))


def solid_params(left, top, width, height, radii, need):
    """(inset, corner) as smr_pack_layouts computes them (smr_layout.hip)."""
    exact = all(half_int(x) for x in (left, top, width, height, need, *radii))
    slack = F(0) if exact else F(0.015625)
    inset = F(need) + slack
    rmax = max(F(r) for r in radii)
    return inset, (rmax + slack if rmax + slack > inset else F(0))

L, res = bench.build_scene()
W, H, TW, TH = 3840, 2160, 128, 16

def box_solid(left, top, w, h, inset, corner, cx0, cy0, cx1, cy1):
    if not (left + inset <= cx0 and cx1 <= left + w - inset and top + inset <= cy0 and cy1 <= top + h - inset): return False
    if corner > 0 and (cx0 < left + corner or cx1 > left + w - corner) and (cy0 < top + corner or cy1 > top + h - corner): return False
    return True

info = []
for l in L:
    need = 0.5
    if l.type == 2: need = max(l.blur_radius / 2, 0.5)
    elif l.border_width >= 1: need = l.border_width + (0.5 if l.type == 0 else 1.0)
    ins, cor = solid_params(l.left, l.top, l.width, l.height, l.border_radius, need)
    ms = [(m.left, m.top, m.width, m.height) + solid_params(m.left, m.top, m.width, m.height, m.radius, 0.5) for m in l.masks]
    opaque = (l.type == 0 and res[l.source_index] == (1920, 1080)) or (l.type != 0 and l.color[3] == 1.0)
    s = 0.0 if all(half_int(v) for v in (l.left, l.top, l.width, l.height)) else 1 / 64
    bb = (math.ceil(l.left - .5 - s), math.ceil(l.left + l.width - .5 + s), math.ceil(l.top - .5 - s), math.ceil(l.top + l.height - .5 + s))
    info.append((l, ins, cor, ms, opaque, bb))
gen = 0
cnt = collections.Counter()
for ty in range(0, H, TH):
    for tx in range(0, W, TW):
        cx0, cy0, cx1, cy1 = tx + .5, ty + .5, tx + TW - .5, ty + TH - .5
        touch, start = [], -1
        for i, (l, ins, cor, ms, opq, bb) in enumerate(info):
            if not (bb[0] < tx + TW and bb[1] > tx and bb[2] < ty + TH and bb[3] > ty): continue
            touch.append(i)
            if box_solid(l.left, l.top, l.width, l.height, ins, cor, cx0, cy0, cx1, cy1) and all(box_solid(*m, cx0, cy0, cx1, cy1) for m in ms) and opq:
                start = i
        g = any(not (i == start) for i in touch if i >= start)
        if g:
            gen += 1
            cnt[tuple(t for t in touch if t >= start)] += 1
print("general tiles", gen, "of", (W // TW) * (H // TH))
for k, v in cnt.most_common(16): print(v, k)
