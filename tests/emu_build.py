"""Builds the lane emulator's libraries (tests/emu/*.cpp: the kernel sources of smelter_amd/csrc compiled for the CPU with SMR_EMU) — test
infrastructure.  SMR_EMU_ASAN=1 (the inner runs of tests/test_emu_asan.py, under LD_PRELOAD of the compiler's AddressSanitizer runtime)
builds them instrumented: LDS arrays (function-local statics here), the dynamic LDS block, register arrays and every host-side buffer get
red zones, and undefined behaviour in the kernels' index arithmetic (misaligned vector loads, shifts, signed overflow) aborts."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU = os.path.join(HERE, "emu")
CSRC = os.path.join(ROOT, "smelter_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def asan_runtime():
    """Path of the shared AddressSanitizer runtime of the clang that builds the emulator, or None."""
    if not os.path.exists(CLANG):
        return None
    r = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = r.stdout.strip()
    return path if r.returncode == 0 and os.path.isabs(path) and os.path.exists(path) else None


LIBS = {  # name -> (source, kernel headers it compiles)
    "smr_emu_convert": ("emu_convert.cpp", ("smr_convert_420.h", "smr_convert_dev.h", "smr_yuv_fast.h")),
    "smr_emu": ("emu_wave.cpp", ("smr_ingest_wave.h", "smr_ingest_common.h", "smr_tables.h", "smr_convert_dev.h", "smr_yuv_fast.h", "smr_convert_420.h")),
    "smr_emu_compose": ("emu_compose.cpp", ("smr_fused_compose.h", "smr_layout_dev.h", "smr_convert_dev.h", "smr_yuv_fast.h", "smr_tables.h")),
}


def build(name, source=None, deps=None):
    """-> path of tests/emu/_build/lib<name>[_asan].so, rebuilt when its source, a kernel header or an emulator header is newer.  The first
    call of a plain (not instrumented) run builds every library that is out of date, side by side: the resampler alone takes half a minute."""
    if source is None:
        source, deps = LIBS[name]
    if not os.environ.get("SMR_EMU_ASAN"):
        from concurrent.futures import ThreadPoolExecutor
        others = [n for n in LIBS if n != name]
        with ThreadPoolExecutor(max_workers=3) as ex:
            futures = [ex.submit(_build_one, n, *LIBS[n]) for n in others]
            lib = _build_one(name, source, deps)
            for f in futures:
                f.result()
        return lib
    return _build_one(name, source, deps)


def _build_one(name, source, deps=()):
    asan = bool(os.environ.get("SMR_EMU_ASAN"))
    out_dir = os.path.join(EMU, "_build")
    os.makedirs(out_dir, exist_ok=True)
    extra = os.environ.get("SMR_EMU_DEFINES", "").split()  # e.g. -DSMR_COMPOSE_BAND_ROWS=8: the same tests against another build of the kernels
    tag = "".join(c if c.isalnum() else "_" for c in "".join(extra))
    lib = os.path.join(out_dir, f"lib{name}{'_asan' if asan else ''}{tag}.so")
    srcs = [os.path.join(EMU, source), os.path.join(EMU, "emu_device.h"), os.path.join(EMU, "emu_guard.h"), os.path.join(EMU, "shim/hip/hip_runtime.h"),
            os.path.join(CSRC, "smr_internal.h")] + [os.path.join(CSRC, d) for d in deps]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in srcs):
        flags = ["-std=c++17", "-fPIC", "-shared", "-DSMR_EMU=1", "-ffp-contract=off", "-Wno-unused-function"]
        flags += ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared-libasan", "-DSMR_EMU_ASAN=1"] if asan else ["-O2"]
        cmd = [CLANG] + flags + extra + ["-I", os.path.join(EMU, "shim"), "-I", EMU, "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", lib, srcs[0], "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    return lib
