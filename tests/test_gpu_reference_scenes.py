"""The reference's own render-test scenes (integration-tests/src/render_tests/{view,rescaler,tiles,tiles_transitions,
transition}.rs — 110 tests, 198 snapshots, extracted by tests/golden/gen_render_test_scenes.py) through the product path:
scene JSON -> smr_renderer_update_scene -> smr_renderer_render (C++ scene engine + HIP kernels), each snapshot compared with
the oracle's restatement of the reference's pass sequence (<= 1 LSB, >= 99 % of the bytes identical) at the test's own
resolution with the harness's TestInput frames.  The snapshots' PNGs are not in the tree (un-vendored submodule): what is
pinned here is that the GPU path and the oracle agree on every scene geometry the reference tests — overflow modes, padding,
absolute positioning, border / radius / box-shadow combinations, fit / fill rescalers and their alignments, tile grids of
1..15 inputs, and the mid-transition states.  The oracle picture of EVERY snapshot — scenes in transition included — is rendered from
layouts that never saw the product's scene engine: oracle/transition.py (the transition state machine, interpolation by component and
tile id) driving oracle/scene.py, over tests/scene_json.py's reading of the JSON; tests/test_oracle_transition.py pins that chain."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import transition as T
from tests import refpipe, scene_json, scenes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORPUS = json.load(open(os.path.join(ROOT, "tests", "golden", "render_test_scenes.json")))["tests"]
OUTPUT_ID = "output_1"  # harness/mod.rs:11


@pytest.fixture(scope="module")
def hip():
    from smelter_amd import hip as h
    return h


@pytest.fixture(scope="module")
def ctxs(hip):
    c = {"gpu_optimized": hip.Context(0, mode=hip.MODE_GPU_OPTIMIZED), "cpu_optimized": hip.Context(0, mode=hip.MODE_CPU_OPTIMIZED)}
    yield c
    for x in c.values():
        x.close()


independent = [0]  # snapshots whose oracle picture came from the oracle's own layouts


def _input_planes(inp):
    if inp["kind"] == "multiscale_grid":
        from smelter_amd import synth
        return synth.multiscale_grid(inp["width"], inp["height"])
    return scenes.test_input(inp["index"], inp["width"], inp["height"])


def test_corpus_is_the_reference_corpus():
    assert len(CORPUS) >= 100
    assert sum(1 for t in CORPUS for s in t["steps"] if "snapshot_ms" in s) >= 150
    assert {t["module"] for t in CORPUS} == {"view", "rescaler", "tiles", "tiles_transitions", "transition"}


@pytest.mark.parametrize("case", CORPUS, ids=[f'{t["module"]}.{t["name"]}' for t in CORPUS])
def test_reference_scene(ctxs, hip, case):
    from smelter_amd import _ffi
    from smelter_amd.renderer import Renderer
    from smelter_amd.scene import Scene
    ctx = ctxs[case["mode"]]
    srgb = case["mode"] == "gpu_optimized"
    W, H = case["resolution"]
    if W % 2 or H % 2:
        pytest.skip("odd output size: 4:2:0 output planes need even dimensions")
    renderer = Renderer(ctx)
    engine = Scene()  # the same scene state beside the renderer's: layouts for the oracle at the same pts sequence
    planes, frames, nodes_o = {}, {}, {}
    for inp in case["inputs"]:
        renderer.register_input(inp["id"])
        planes[inp["id"]] = _input_planes(inp)
        frames[inp["id"]] = ctx.frame(hip.FRAME_PLANAR_YUV420, inp["width"], inp["height"], list(planes[inp["id"]]))
        nodes_o[inp["id"]] = orc.planar_yuv_to_rgba(*planes[inp["id"]], inp["width"], inp["height"], omp=True)
    graph = None
    snaps = 0
    oracle_scene, oracle_inputs = T.SceneState(), None   # the oracle's own scene state: same updates, same pts sequence
    try:
        for step in case["steps"]:
            if "update" in step:
                renderer.update_scene(OUTPUT_ID, W, H, step["update"])
                graph = engine.update(step["update"], W, H)
                oracle_root, oracle_inputs = scene_json.to_oracle(step["update"])
                oracle_scene.update(oracle_root, W, H)
                continue
            pts_ms = step.get("snapshot_ms", step.get("render_ms"))
            got = renderer.render(pts_ms / 1e3, frames)[OUTPUT_ID].download()
            # oracle: the root layout node's flattened list at this pts, inputs in child order
            assert graph[0].kind == _ffi.NODE_LAYOUT, "corpus scenes are layout trees"
            kids = [graph[k] for k in graph[0].children]
            assert all(k.kind == _ffi.NODE_INPUT_STREAM for k in kids)
            res = [(nodes_o[k.ref_id].shape[1], nodes_o[k.ref_id].shape[0]) if k.ref_id in nodes_o else None for k in kids]
            engine.layouts(0, int(pts_ms * 1e6), res, hip.MODE_GPU_OPTIMIZED if srgb else hip.MODE_CPU_OPTIMIZED)  # (keeps the twin engine's pts history in step)
            # the picture the product is compared with is rendered from layouts that never saw the product's scene engine: the oracle's own
            # scene state at this pts (every rendered pts advances it, as register_render_event does)
            assert [k.ref_id for k in kids] == oracle_inputs
            layouts = oracle_scene.layouts(int(round(pts_ms * 1e6)), res, srgb=srgb)
            if "snapshot_ms" not in step:
                continue
            independent[0] += 1
            want, _ = refpipe.render_yuv420(layouts, [nodes_o.get(k.ref_id) for k in kids], W, H, srgb=srgb, omp=True)
            for g, w_, pl in zip(got, want, "YUV"):
                d, ex = refpipe.max_diff(g, w_), refpipe.exact_fraction(g, w_)
                assert d <= 1, f'{case["name"]} @ {pts_ms} ms plane {pl}: {d} LSB off the oracle'
                assert ex >= 0.99, f'{case["name"]} @ {pts_ms} ms plane {pl}: only {ex:.4f} identical'
            snaps += 1
    finally:
        renderer.close()
        for f in frames.values():
            f.destroy()
    assert snaps >= 1


def test_most_snapshots_were_checked_against_independent_layouts():
    """(runs after the scenes: pytest keeps file order)"""
    assert independent[0] >= sum(1 for t in CORPUS for s in t["steps"] if "snapshot_ms" in s and not (t["resolution"][0] % 2 or t["resolution"][1] % 2))
