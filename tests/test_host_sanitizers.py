"""The host side of the library (scene engine, text pipeline, renderer: smelter_amd/csrc/host) under AddressSanitizer + UBSan.

tests/san/host_fuzz.cpp is compiled by g++ together with the host sources (no HIP involved: they are plain C++17) and drives the scene
engine and the text pipeline through the C ABI with the reference's own scenes, mutated scenes and corrupted fonts.
tests/san/renderer_fuzz.cpp does the same to the renderer (registry, update_scene, render-graph walk, lanes, text nodes), linked against
tests/san/null_device.cpp — a stand-in for the GPU half of the ABI that checks and reads every argument, counts the surfaces it hands
out and fails allocations on request.  A rejected input is an answer; a memory error, undefined behaviour, a leaked device surface or a
hang fails the test.  (What they found when they were written: an empty-contour glyph whose extents stayed at infinity, pen positions
of absurd font sizes cast to int, a cmap-12 group count above 2^31 that sent the binary search round in circles, an output frame
leaked when the second of a lane's two frames could not be allocated, float -> integer casts of hostile node sizes.)"""
import json
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HOST = os.path.join(ROOT, "smelter_amd", "csrc", "host")
BUILD = os.path.join(HERE, "san", "_build")
HOST_SOURCES = [os.path.join(HOST, f) for f in ("scene.cpp", "scene_build.cpp", "scene_capi.cpp", "text.cpp", "text_capi.cpp")]
PROGRAMS = {  # executable -> its sources beyond HOST_SOURCES
    "host_fuzz": [os.path.join(HERE, "san", "host_fuzz.cpp")],
    "renderer_fuzz": [os.path.join(HERE, "san", "renderer_fuzz.cpp"), os.path.join(HERE, "san", "null_device.cpp"), os.path.join(HOST, "renderer.cpp")],
}
SOURCES = HOST_SOURCES + [p for v in PROGRAMS.values() for p in v]
SANITIZE = "-fsanitize=address,undefined,float-cast-overflow"
FLAGS = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", SANITIZE, "-fno-sanitize-recover=undefined,float-cast-overflow",
         "-I", os.path.join(ROOT, "include"), "-I", HOST]
FONT_DIRS = ["/usr/share/fonts/truetype/dejavu", "/root/reference/smelter-render/fonts"]


def _newest_dependency():
    deps = SOURCES + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")] + [os.path.join(ROOT, "include", "smr.h")]
    return max(os.path.getmtime(p) for p in deps)


@pytest.fixture(scope="module")
def harness():
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    os.makedirs(BUILD, exist_ok=True)
    probe = os.path.join(BUILD, "probe.cpp")
    with open(probe, "w") as f:
        f.write("int main() { return 0; }\n")
    if subprocess.run([gxx, SANITIZE, probe, "-o", os.path.join(BUILD, "probe")], capture_output=True).returncode != 0:
        pytest.skip("this g++ has no sanitizer runtimes")
    exes = {name: os.path.join(BUILD, name) for name in PROGRAMS}
    newest = _newest_dependency()
    if all(os.path.exists(e) and os.path.getmtime(e) >= newest for e in exes.values()):
        return exes

    def compile_one(src):
        obj = os.path.join(BUILD, os.path.basename(src) + ".o")
        r = subprocess.run([gxx] + FLAGS + ["-c", src, "-o", obj], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return obj
    with ThreadPoolExecutor(max_workers=9) as ex:
        objs = dict(zip(SOURCES, ex.map(compile_one, SOURCES)))
    for name, own in PROGRAMS.items():
        r = subprocess.run([gxx, SANITIZE] + [objs[p] for p in HOST_SOURCES + own] + ["-o", exes[name]], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    return exes


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    """One file per scene: the reference's API vectors (smelter-api/tests/scene_deserialization.rs, accepted and rejected ones alike) and
    every update of its render tests (tests/golden/*.json, extracted by the scripts beside them); the fonts this machine has."""
    out = tmp_path_factory.mktemp("host_fuzz_corpus")
    n = 0
    with open(os.path.join(HERE, "golden", "scene_api_vectors.json")) as f:
        for v in json.load(f)["vectors"]:
            (out / f"api_{n:04d}.json").write_text(json.dumps(v["scene"]))
            n += 1
    with open(os.path.join(HERE, "golden", "render_test_scenes.json")) as f:
        for t in json.load(f)["tests"]:
            for step in t["steps"]:
                if "update" in step:
                    (out / f"render_{n:04d}.json").write_text(json.dumps(step["update"]))
                    n += 1
    fonts = 0
    for d in FONT_DIRS:
        if os.path.isdir(d):
            for name in sorted(os.listdir(d)):
                if name.endswith(".ttf") and fonts < 6:
                    shutil.copy(os.path.join(d, name), out / f"{fonts}_{name}")
                    fonts += 1
    return str(out), n, fonts


def _run(exe, corpus_dir, iterations, seed, timeout):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, corpus_dir, str(iterations), str(seed)], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, f"seed {seed}: rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
    assert "runtime error" not in r.stderr, r.stderr[-6000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_scene_engine_and_text_pipeline_survive_hostile_input(harness, corpus, seed):
    corpus_dir, n_scenes, n_fonts = corpus
    got = _run(harness["host_fuzz"], corpus_dir, 4000, seed, timeout=600)
    assert got["corpus_scenes"] == n_scenes
    # every scene of the corpus runs twice (alone; as an update of a scene that lives on): what the reference accepts must be accepted
    assert got["corpus_ok"] >= n_scenes, got
    assert got["scenes_ok"] > got["corpus_ok"] + 500 and got["scenes_rejected"] > 500, got   # mutations reach both sides of the validator
    assert got["layouts"] > 10000, got
    if n_fonts:
        assert got["runs"] > 500 and got["glyphs"] > 5000, got
        assert got["fonts_ok"] > n_fonts and got["fonts_rejected"] > 50, got                  # corrupted fonts: some load, some are refused


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_renderer_survives_hostile_hosts_and_a_failing_device(harness, corpus, seed):
    """smr_renderer_* on the null device: scenes (also mutated) on three outputs, frames of every format — fresh, stale, missing, for
    unknown inputs — on one to three lanes, registrations coming and going, glyph runs, a font book, device allocations that fail.  The
    null device refuses what the library would refuse and aborts on what would corrupt memory; every surface it handed out must be back
    when a renderer is destroyed."""
    corpus_dir, n_scenes, n_fonts = corpus
    got = _run(harness["renderer_fuzz"], corpus_dir, 12000, seed, timeout=600)
    assert got["updates_ok"] > 1000 and got["updates_refused"] > 500, got
    assert got["renders_ok"] > 4000 and got["frames_out"] > got["renders_ok"], got     # several outputs per render
    assert got["renders_refused"] > 50 and got["injected_failures"] > 100, got          # the error paths ran
    assert got["text_runs"] > 10, got


def test_replay_modes(harness, corpus, tmp_path):
    """--scene / --font replay one input (what HOST_FUZZ_TRACE leaves behind after an abort)."""
    harness = harness["host_fuzz"]
    corpus_dir = corpus[0]
    scene = sorted(f for f in os.listdir(corpus_dir) if f.endswith(".json"))[0]
    r = subprocess.run([harness, "--scene", os.path.join(corpus_dir, scene)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    fonts = sorted(f for f in os.listdir(corpus_dir) if f.endswith(".ttf"))
    if fonts:
        r = subprocess.run([harness, "--font", os.path.join(corpus_dir, fonts[0])], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]


def test_edges_that_end_on_the_bitmap_border_stay_inside_the_accumulator(harness):
    """x advanced row by row could arrive at -1e-17 for an edge that ends exactly on column 0 (floor -> column -1, eight bytes in
    front of the accumulator).  A million random edges ending on the first / last column, the advisor's own case first, into accumulators of
    exactly w * h + 1 doubles under AddressSanitizer."""
    r = subprocess.run([harness["host_fuzz"], "--edges", "1000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "runtime error" not in r.stderr, r.stderr[-4000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["edges"] == 1000000
