"""The reference's own render-test scenes — all 143 of integration-tests/src/render_tests/{view,rescaler,tiles,tiles_transitions,transition,
text,image,shader,simple}.rs, extracted by tests/golden/gen_render_test_scenes.py; first the 110 layout scenes (198 snapshots), at the end of
the file the 33 with Text / Image / Shader nodes — through the product path:
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
ALL = json.load(open(os.path.join(ROOT, "tests", "golden", "render_test_scenes.json")))["tests"]
LAYOUT_MODULES = {"view", "rescaler", "tiles", "tiles_transitions", "transition"}
def _only_layout_components(test):
    """True for the corpus scenes that are trees of View / Rescaler / Tiles over input streams (110 of the reference's 143 render tests)."""
    def ok(c):
        if c["type"] == "input_stream":
            return True
        if c["type"] not in ("view", "rescaler", "tiles"):
            return False
        return all(ok(k) for k in c.get("children", [])) and ("child" not in c or ok(c["child"]))
    return all(ok(s["update"]) for s in test["steps"] if "update" in s)


CORPUS = [t for t in ALL if t["module"] in LAYOUT_MODULES and _only_layout_components(t)]  # layout trees over input streams: checked against the oracle's OWN scene state
NODE_CORPUS = [t for t in ALL if t not in CORPUS]  # text.rs, image.rs, shader.rs, simple.rs + tiles.rs' labelled video call: every node kind (below)
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
    assert {t["module"] for t in CORPUS} == LAYOUT_MODULES


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


# ------------------------------------------------------------------------------------------------------------------------------------------
# text.rs (15), image.rs (8), shader.rs (8), simple.rs (1): the rest of the reference's 143 render tests — scenes with Text, Image and Shader
# nodes, also as the ROOT of an output.  The product renders them through smr_renderer_* (the renderer draws its Text nodes with its font book,
# scales images, runs the built-in ports of the test shaders); the oracle side walks the same node graph depth first with the oracle's passes:
#   InputStream  orc.planar_yuv_to_rgba                     Text    the Python text twin's glyph run + orc.blit_glyphs
#   Image        orc.add_premultiplied_alpha (+ bilinear)    Shader  orc.builtin_shader on the children's textures, ShaderParam::to_bytes
#   layout node  the engine's layouts at this pts + refpipe.layout_node_render            root  (bilinear to the output size +) orc.rgba_to_planar_yuv
# What these scenes cannot pin (stated in include/smr.h and DESIGN.md): glyph SHAPES against glyphon (fonts of this machine, own rasteriser), and
# image DECODING — out of scope: the assets the tests load (a JPEG from a URL, an SVG, two GIFs) are replaced by seeded bitmaps
# of fixed sizes registered under the same ids; an animated GIF is its first frame.
SHADER_OF = {"layout_planes.wgsl": "SHADER_LAYOUT_PLANES", "fade_to_ball.wgsl": "SHADER_FADE_TO_BALL", "color_output_with_texture_count.wgsl": "SHADER_COLOR_BY_TEXTURE_COUNT",
             "red_border.wgsl": "SHADER_RED_BORDER", "circle_layout.wgsl": "SHADER_CIRCLE_LAYOUT"}
IMAGE_SIZES = {"jpeg": (1200, 630), "svg": (512, 512), "gif": (300, 200)}


def _bitmap(renderer_spec):
    import zlib
    w, h = IMAGE_SIZES[renderer_spec["image_type"]]
    rng = np.random.default_rng(zlib.crc32(renderer_spec["id"].encode()))
    xx, yy = np.meshgrid(np.arange(w), np.arange(h))
    img = np.stack([(xx * 255 // w), (yy * 255 // h), ((xx + yy) * 3 % 256), np.full((h, w), 255)], axis=-1).astype(np.uint8)
    img[h // 4:h // 2, w // 4:w // 2, 3] = 128                       # a translucent patch (straight alpha: premultiplied on registration)
    img[..., :3] ^= rng.integers(0, 16, (h, w, 3), dtype=np.uint8)
    return np.ascontiguousarray(img)


def _leaves(component):
    """The non-layout components of a scene in depth-first order — the order of the render graph's non-layout nodes."""
    out = []
    t = component["type"]
    if t in ("text", "image", "input_stream"):
        out.append(component)
    elif t == "shader":
        out.append(component)
        for k in component.get("children", []):
            out += _leaves(k)
    else:
        kids = component.get("children", [])
        if "child" in component:
            kids = [component["child"]]
        for k in kids:
            out += _leaves(k)
    return out


def _param_bytes(p):
    import struct
    if p is None:
        return b""
    if p["type"] == "f32":
        return struct.pack("<f", p["value"])
    if p["type"] == "u32":
        return struct.pack("<I", p["value"])
    if p["type"] == "i32":
        return struct.pack("<i", p["value"])
    return b"".join(_param_bytes(x) for x in p["value"])


def test_node_corpus_is_the_rest_of_the_reference_corpus():
    assert len(ALL) == 143 and len(CORPUS) == 110
    assert sorted((m, sum(1 for t in NODE_CORPUS if t["module"] == m)) for m in {t["module"] for t in NODE_CORPUS}) == [("image", 8), ("shader", 8), ("simple", 1), ("text", 15), ("tiles", 1)]


@pytest.mark.parametrize("case", NODE_CORPUS, ids=[f'{t["module"]}.{t["name"]}' for t in NODE_CORPUS])
def test_reference_scene_with_text_image_and_shader_nodes(ctxs, hip, case):
    from smelter_amd import _ffi
    from smelter_amd.renderer import Renderer
    from smelter_amd.scene import Scene
    from tests import text_twin as TT
    ctx = ctxs[case["mode"]]
    srgb = case["mode"] == "gpu_optimized"
    mode = hip.MODE_GPU_OPTIMIZED if srgb else hip.MODE_CPU_OPTIMIZED
    W, H = case["resolution"]
    renderer, engine = Renderer(ctx), Scene()
    has_text = any(c["type"] == "text" for st in case["steps"] if "update" in st for c in _leaves(st["update"]))
    native = twin = None
    if has_text:
        try:
            native, twin = TT.NativeFontBook.system(), TT.Shaper(TT.FontBook.system())
        except FileNotFoundError:
            pytest.skip("no TrueType fonts on this machine")
        renderer.set_fontbook(native)
        engine.set_text_measurer(twin.measurer)
    planes, frames, nodes_o, images, shader_ids = {}, {}, {}, {}, {}
    for inp in case["inputs"]:
        renderer.register_input(inp["id"])
        planes[inp["id"]] = _input_planes(inp)
        frames[inp["id"]] = ctx.frame(hip.FRAME_PLANAR_YUV420, inp["width"], inp["height"], list(planes[inp["id"]]))
        nodes_o[inp["id"]] = orc.planar_yuv_to_rgba(*planes[inp["id"]], inp["width"], inp["height"], omp=True)
    for spec in case.get("renderers", []):
        if spec["kind"] == "image":
            bmp = _bitmap(spec)
            renderer.register_image(spec["id"], bmp)
            engine.register_image(spec["id"], bmp.shape[1], bmp.shape[0])
            images[spec["id"]] = orc.add_premultiplied_alpha(bmp, srgb)
        else:
            sid = getattr(_ffi, SHADER_OF[spec["wgsl"]])
            renderer.register_shader(spec["id"], sid)
            shader_ids[spec["id"]] = getattr(orc, SHADER_OF[spec["wgsl"]])
    snaps = 0
    try:
        graph = leaves = None
        for step in case["steps"]:
            if "update" in step:
                renderer.update_scene(OUTPUT_ID, W, H, step["update"])
                graph = engine.update(step["update"], W, H)
                leaves = _leaves(step["update"])
                continue
            pts_ms = step.get("snapshot_ms", step.get("render_ms"))
            got = renderer.render(pts_ms / 1e3, frames)[OUTPUT_ID].download()
            comp_of = dict(zip([n.index for n in graph if n.kind != _ffi.NODE_LAYOUT], leaves))
            assert len(comp_of) == len(leaves)

            def surface(i):  # -> HxWx4 RGBA8 node texture of graph node i, or None
                n = graph[i]
                if n.kind == _ffi.NODE_INPUT_STREAM:
                    return nodes_o.get(n.ref_id)
                if n.kind == _ffi.NODE_IMAGE:
                    img = images[n.ref_id]
                    return img if (img.shape[1], img.shape[0]) == (n.width, n.height) else orc.rescale_bilinear(img, n.width, n.height)
                if n.kind == _ffi.NODE_TEXT:
                    c = comp_of[i]
                    col = tuple(int(c.get("color", "#FFFFFFFF")[k:k + 2], 16) for k in (1, 3, 5, 7))
                    bg = tuple(int(c.get("background_color", "#00000000")[k:k + 2], 16) for k in (1, 3, 5, 7))
                    cap = lambda v: "".join(w.capitalize() for w in v.split("_"))
                    glyphs, atlas = twin.rasterise(c["text"], n.width, n.height, c["font_size"], c.get("line_height"), family=c.get("font_family", ""),
                                                   weight=cap(c.get("weight", "normal")), style=cap(c.get("style", "normal")), wrap=cap(c.get("wrap", "none")),
                                                   align=cap(c.get("align", "left")), color=tuple(v / 255.0 for v in col))
                    return orc.blit_glyphs(n.width, n.height, orc.color_to_shader(bg, srgb), glyphs, atlas, srgb)
                kids = [surface(k) for k in n.children]
                if n.kind == _ffi.NODE_SHADER:
                    c = comp_of[i]
                    return orc.builtin_shader(shader_ids[n.ref_id], [k for k in kids if k is not None], n.width, n.height, params=_param_bytes(c.get("shader_param")),
                                              time=pts_ms / 1e3, srgb=srgb)
                res = [(k.shape[1], k.shape[0]) if k is not None else None for k in kids]
                from smelter_amd.scene import Layout
                arr, cnt, w, h = engine.node_layouts(i, int(round(pts_ms * 1e6)), res, mode)
                layouts = [Layout.from_c(arr[k]) for k in range(cnt)]
                return refpipe.layout_node_render(layouts, kids, w, h, srgb=srgb, omp=True) if w and h else None
            root = surface(0)
            if "snapshot_ms" not in step:
                continue
            if root is None:
                want = [np.full((H, W), 16, np.uint8), np.full((H // 2, W // 2), 128, np.uint8), np.full((H // 2, W // 2), 128, np.uint8)]
            else:
                if (root.shape[1], root.shape[0]) != (W, H):
                    root = orc.rescale_bilinear(root, W, H)
                want = orc.rgba_to_planar_yuv(root, orc.YUV420, omp=True)
            for g, w_, pl in zip(got, want, "YUV"):
                d, ex = refpipe.max_diff(g, w_), refpipe.exact_fraction(g, w_)
                assert d <= 1, f'{case["name"]} @ {pts_ms} ms plane {pl}: {d} LSB off the oracle'
                assert ex >= 0.99, f'{case["name"]} @ {pts_ms} ms plane {pl}: only {ex:.4f} identical'
            snaps += 1
    finally:
        renderer.close()
        for f in frames.values():
            f.destroy()
        if native is not None:
            native.close()
    assert snaps >= 1
