"""bench.py's self-evidence (no GPU): counter profiles are quoted only for the library they were collected on, the sharded path's watchdog
turns a stuck step into one JSON error line and exit code 3, the roofline block's headline cannot be moved by splitting kernels."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from smelter_amd import build as B  # noqa: E402


def test_device_code_hash_is_the_fatbin_sections_and_ignores_host_code(tmp_path):
    h = B.kernels_sha256()
    assert len(h) == 64 and h == B.kernels_sha256()
    # appending bytes to the file (host side, after the sections) does not change the identity; another fatbin would
    lib2 = tmp_path / "lib.so"
    data = open(B.LIB, "rb").read()
    lib2.write_bytes(data + b"trailing host bytes")
    assert B.kernels_sha256(str(lib2)) == h
    with pytest.raises(ValueError):
        B.kernels_sha256(__file__)


def test_profiles_are_quoted_only_for_the_library_they_were_collected_on(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    data, ident = bench.quoted_profile("r06_traffic.json")
    assert data is None and ident["stale"] and "does not exist" in ident["why"]
    json.dump({"k": {"hbm_bytes_per_launch": 1}, "_identity": {"lib_kernels_sha256": "0" * 64}}, open(tmp_path / "profiles" / "r06_traffic.json", "w"))
    data, ident = bench.quoted_profile("r06_traffic.json")
    assert data is None and ident["stale"] and "was collected on device code 000000000000" in ident["why"]
    json.dump({"k": {"hbm_bytes_per_launch": 1}, "_identity": {"lib_kernels_sha256": B.kernels_sha256()}}, open(tmp_path / "profiles" / "r06_traffic.json", "w"))
    data, ident = bench.quoted_profile("r06_traffic.json")
    assert data is not None and not ident["stale"]


def test_committed_profiles_belong_to_the_library_that_is_built():
    """The counter files bench.py quotes in the judged line were collected on exactly this device code (the build is deterministic)."""
    have = B.kernels_sha256()
    for name in ("r06_traffic.json", "r06_issue.json", "r06_traffic_configs3.json"):
        ident = json.load(open(os.path.join(ROOT, "profiles", name)))["_identity"]
        assert ident["lib_kernels_sha256"] == have, f"profiles/{name} is stale: re-run tools/round_end_r06.sh on the final build"


def test_whole_frame_roofline_does_not_move_when_a_kernel_is_split(monkeypatch):
    args = types.SimpleNamespace(config=9, ingest="auto")  # (no committed counters for this config: traffic None)
    monkeypatch.setattr(bench, "ALGO_BYTES_PER_FRAME", 37324800)
    kb = {"a": 1, "b": 1, "c": 1}
    one = bench.roofline_block(args, {"a": {"avg_us": 55.0}, "c": {"avg_us": 16.0}}, kb, {}, 0.06)
    split = bench.roofline_block(args, {"a": {"avg_us": 22.0}, "b": {"avg_us": 33.0}, "c": {"avg_us": 16.0}}, kb, {}, 0.06)
    assert one["frac"] == split["frac"] and one["avg_launch_us"] == split["avg_launch_us"] == 71.0
    assert split["dominant_kernel"]["frac"] > one["dominant_kernel"]["frac"]  # (the single-launch reading does move: reported, not the headline)
    assert one["traffic"] is None and one["bound"] == "hbm" and abs(one["frac"] - 37324800 / 71e-6 / 8e12) < 1e-4


def test_watchdog_ends_a_stuck_rank_with_an_error_line_and_rc_3():
    code = ("import sys, time; sys.path.insert(0, %r); import bench; w = bench.Watchdog(0.6, 0, 8, 3); w.beat('step 7'); time.sleep(30)" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert p.returncode == 3
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 8 and "no progress" in line["error"] and "step 7" in line["error"]
    # ... and stays silent while the steps keep coming
    code = ("import sys, time; sys.path.insert(0, %r); import bench; w = bench.Watchdog(0.6, 1, 8, 3)\n"
            "for i in range(12):\n    w.beat('step'); time.sleep(0.2)\nw.stop(); time.sleep(1.0); print('done')" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert p.returncode == 0 and p.stdout.strip() == "done"


def test_device_code_identity_does_not_depend_on_the_build_directory(tmp_path):
    """kernels_sha256 — what ties committed counter profiles to a library — hashes the code objects' machine code and descriptors, not the
    fatbin as a whole: the offload bundles carry a compilation-unit id derived from the source's absolute path, so the same sources built in
    two directories differ in fatbin_sha256 and must not differ in kernels_sha256 (a checkout elsewhere still matches profiles/)."""
    import shutil
    hipcc = B.HIPCC
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    got = []
    for name in ("here", "somewhere/else/entirely"):
        d = tmp_path / name
        shutil.copytree(os.path.join(ROOT, "smelter_amd", "csrc"), d / "csrc", ignore=shutil.ignore_patterns("host"))
        shutil.copytree(os.path.join(ROOT, "include"), d / "include")
        obj, lib = str(d / "m.o"), str(d / "libm.so")
        flags = [f for f in B.FLAGS if not f.startswith(ROOT) and f != "-I"] + ["-I", str(d / "include"), "-I", str(d / "csrc")]
        subprocess.run([hipcc] + flags + ["-x", "hip", "-c", str(d / "csrc" / "smr_misc.hip"), "-o", obj], check=True, capture_output=True)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj], check=True, capture_output=True)
        got.append((B.kernels_sha256(lib), B.fatbin_sha256(lib)))
    assert got[0][0] == got[1][0], got
    assert got[0][1] != got[1][1], "the fatbin no longer depends on the path: kernels_sha256 could be simplified again"
