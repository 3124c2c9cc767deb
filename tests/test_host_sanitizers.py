"""The host side of the library (scene engine + text pipeline, smelter_amd/csrc/host) under AddressSanitizer + UBSan.

tests/san/host_fuzz.cpp is compiled by g++ together with the host sources (no HIP involved: they are plain C++17) and drives them through
the C ABI with the reference's own scenes, mutated scenes and corrupted fonts.  A rejected input is an answer; a memory error, undefined
behaviour or a hang fails the test.  (What it found when it was written: an empty-contour glyph whose extents stayed at infinity, pen
positions of absurd font sizes cast to int, a cmap-12 group count above 2^31 that sent the binary search round in circles.)"""
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
SOURCES = [os.path.join(HERE, "san", "host_fuzz.cpp")] + [os.path.join(HOST, f) for f in
                                                         ("scene.cpp", "scene_build.cpp", "scene_capi.cpp", "text.cpp", "text_capi.cpp")]
FLAGS = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
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
    if subprocess.run([gxx, "-fsanitize=address,undefined", probe, "-o", os.path.join(BUILD, "probe")], capture_output=True).returncode != 0:
        pytest.skip("this g++ has no sanitizer runtimes")
    exe = os.path.join(BUILD, "host_fuzz")
    if os.path.exists(exe) and os.path.getmtime(exe) >= _newest_dependency():
        return exe

    def compile_one(src):
        obj = os.path.join(BUILD, os.path.basename(src) + ".o")
        r = subprocess.run([gxx] + FLAGS + ["-c", src, "-o", obj], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return obj
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([gxx, "-fsanitize=address,undefined"] + objs + ["-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


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
    got = _run(harness, corpus_dir, 4000, seed, timeout=600)
    assert got["corpus_scenes"] == n_scenes
    # every scene of the corpus runs twice (alone; as an update of a scene that lives on): what the reference accepts must be accepted
    assert got["corpus_ok"] >= n_scenes, got
    assert got["scenes_ok"] > got["corpus_ok"] + 500 and got["scenes_rejected"] > 500, got   # mutations reach both sides of the validator
    assert got["layouts"] > 10000, got
    if n_fonts:
        assert got["runs"] > 500 and got["glyphs"] > 5000, got
        assert got["fonts_ok"] > n_fonts and got["fonts_rejected"] > 50, got                  # corrupted fonts: some load, some are refused


def test_replay_modes(harness, corpus, tmp_path):
    """--scene / --font replay one input (what HOST_FUZZ_TRACE leaves behind after an abort)."""
    corpus_dir = corpus[0]
    scene = sorted(f for f in os.listdir(corpus_dir) if f.endswith(".json"))[0]
    r = subprocess.run([harness, "--scene", os.path.join(corpus_dir, scene)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    fonts = sorted(f for f in os.listdir(corpus_dir) if f.endswith(".ttf"))
    if fonts:
        r = subprocess.run([harness, "--font", os.path.join(corpus_dir, fonts[0])], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
