"""The reference's per-frame pass sequence, restated with the CPU oracle (test infrastructure).

populate_inputs -> resample_scaled_children -> LayoutShader::render -> read_outputs
(smelter-render/src/state/render_loop.rs:19-230, transformations/layout.rs:169-278)."""
from __future__ import annotations

from dataclasses import replace

import numpy as np

from oracle import oracle as orc


def nodes_from_yuv420(planes_list, omp=False):
    return [orc.planar_yuv_to_rgba(y, u, v, y.shape[1], y.shape[0], orc.YUV420, omp=omp) for (y, u, v) in planes_list]


def layout_node_render(layouts, nodes, W, H, srgb=True, omp=False):
    """nodes: list of RGBA8 node textures (or None). Returns the RGBA8 output node texture."""
    eff, srcs = [], list(nodes)
    for L in layouts:
        if L.type == 0 and srgb and L.source_index < len(nodes) and nodes[L.source_index] is not None:
            dw, dh = max(_rust_round(L.width), 1), max(_rust_round(L.height), 1)  # layout.rs:258-261
            kind, tile = orc.resample(nodes[L.source_index], L.crop, dw, dh, omp=omp)
            if kind > 0:
                srcs.append(tile)
                L = replace(L, crop=(0.0, 0.0, float(dw), float(dh)), source_index=len(srcs) - 1)
        eff.append(L)
    return orc.apply_layouts(W, H, eff, srcs, srgb=srgb, omp=omp)


def _rust_round(x) -> int:
    # f32::round: half away from zero
    x = float(np.float32(x))
    return int(np.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))


def render_yuv420(layouts, nodes, W, H, srgb=True, omp=False):
    rgba = layout_node_render(layouts, nodes, W, H, srgb, omp)
    return orc.rgba_to_planar_yuv(rgba, orc.YUV420, omp=omp), rgba


def max_diff(a, b) -> int:
    return int(np.abs(np.asarray(a, np.int32) - np.asarray(b, np.int32)).max()) if np.asarray(a).size else 0


def exact_fraction(a, b) -> float:
    a, b = np.asarray(a), np.asarray(b)
    return float((a == b).mean())
