"""CPU-side checks of the drop-in boundary: libsmr_hip.so loads and exports every symbol that
include/smr.h declares, struct layouts agree across C / ctypes / oracle, and — with no GPU in the
container — context creation fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from smelter_amd import _ffi
    return _ffi.load()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "smr.h")).read()
    declared = set(re.findall(r"SMR_API\s+[\w\s\*]+?\b(smr_\w+)\s*\(", hdr))
    assert len(declared) >= 30
    from smelter_amd import _ffi
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), f"libsmr_hip.so does not export {name}"


def test_struct_layouts_agree(lib):
    from smelter_amd import _ffi
    from oracle import oracle
    assert lib.smr_sizeof_layout() == C.sizeof(_ffi.Layout) == C.sizeof(oracle._Layout) == 744
    assert C.sizeof(_ffi.Mask) == 32
    assert lib.smr_abi_version() == 2  # include/smr.h SMR_ABI_VERSION (history there)


def test_product_build_reads_nothing_from_the_environment(lib):
    """A product build (what build() makes, what everything is measured on) holds no A/B knob: smr_build_flags() is 0 and the library does
    not even import getenv — the knobs of laboratory builds (tools/variant.sh, -DSMR_LAB) are compiled out, not switched off."""
    import subprocess
    from smelter_amd import build
    if os.environ.get("SMR_LIB") or os.environ.get("SMR_LAB"):
        pytest.skip("a laboratory build was asked for")
    lib.smr_build_flags.restype = C.c_uint32
    assert lib.smr_build_flags() == 0
    undefined = subprocess.run(["nm", "-D", "--undefined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    imported = {line.split()[-1].split("@")[0] for line in undefined.splitlines() if line.strip()}
    assert not ({"getenv", "secure_getenv"} & imported), imported & {"getenv", "secure_getenv"}
    assert "hipLaunchKernel" in imported or any(n.startswith("hip") for n in imported), "nm listed no HIP imports: wrong file?"


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from smelter_amd import hip
    with pytest.raises(hip.SmrError):
        hip.Context(0)


def test_plan_make_is_host_only_and_matches_oracle(lib):
    # smr_resample_plan_make is a pure host function: usable without a device
    from smelter_amd import _ffi
    from oracle import oracle
    cases = [((0, 0, 640, 360), (640, 360), (640, 360)), ((40, 100, 640, 360), (640, 360), (1920, 1080)),
             ((0, 100, 640, 360), (640, 300), (1920, 1080)), ((0, 100.5, 640, 360), (640, 300), (1920, 1080)),
             ((0, 0, 1920, 1080), (960, 270), (1920, 1080)), ((0, 0, 5760, 3240), (640, 360), (5760, 3240)),
             ((0, 0, 3840, 2160), (960, 540), (3840, 2160)), ((0, 0, 1920, 1080), (1266, 712), (1920, 1080)),
             ((3.5, 7.25, 301.5, 200.0), (97, 411), (640, 360))]
    for crop, dst, src in cases:
        p = _ffi.ResamplePlan()
        c = (C.c_float * 4)(*crop)
        kind = lib.smr_resample_plan_make(src[0], src[1], c, dst[0], dst[1], C.byref(p))
        o = oracle.resample_plan(src[0], src[1], crop, dst[0], dst[1])
        assert kind == o.kind == p.kind
        if kind:
            assert tuple(p.levels) == o.levels and (p.reduced_w, p.reduced_h) == o.reduced
            n = 2 if kind == 2 else 1
            assert tuple(p.axis)[:n] == o.axis[:n] and tuple(p.perp_offset)[:n] == o.perp_offset[:n]
            assert tuple(p.scale)[:n] == o.scale[:n] and tuple(p.offset)[:n] == o.offset[:n]
            if kind == 2:
                assert (p.mid_w, p.mid_h) == o.mid


def _build_c_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "render_scene")
    libdir = os.path.join(ROOT, "smelter_amd")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "render_scene.c"),
           "-o", exe, "-L", libdir, "-l:libsmr_hip.so", f"-Wl,-rpath,{libdir}", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_header_is_plain_c_and_the_example_links(lib, tmp_path):
    """include/smr.h compiles as C11 (-Wall -Wextra -Werror) and examples/render_scene.c links against the library; with no
    GPU in the container the program stops at smr_ctx_create — loudly, exit code 2."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "no HIP device" in p.stderr
