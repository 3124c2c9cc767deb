"""Builds the lane emulator's libraries (tests/emu/*.cpp: the kernel sources of smelter_amd/csrc compiled for the CPU with SMR_EMU) — test
infrastructure.  SMR_EMU_ASAN=1 (the inner runs of tests/test_emu_asan.py, under LD_PRELOAD of the compiler's AddressSanitizer runtime)
builds them instrumented: LDS arrays (function-local statics here), the dynamic LDS block, register arrays and every host-side buffer get
red zones."""
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


def build(name, source, deps=()):
    """-> path of tests/emu/_build/lib<name>[_asan].so, rebuilt when its source, a kernel header or an emulator header is newer."""
    asan = bool(os.environ.get("SMR_EMU_ASAN"))
    out_dir = os.path.join(EMU, "_build")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, f"lib{name}{'_asan' if asan else ''}.so")
    srcs = [os.path.join(EMU, source), os.path.join(EMU, "emu_device.h"), os.path.join(EMU, "emu_guard.h"), os.path.join(EMU, "shim/hip/hip_runtime.h"),
            os.path.join(CSRC, "smr_internal.h")] + [os.path.join(CSRC, d) for d in deps]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in srcs):
        flags = ["-std=c++17", "-fPIC", "-shared", "-DSMR_EMU=1", "-ffp-contract=off", "-Wno-unused-function"]
        flags += ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address", "-shared-libasan", "-DSMR_EMU_ASAN=1"] if asan else ["-O2"]
        cmd = [CLANG] + flags + ["-I", os.path.join(EMU, "shim"), "-I", EMU, "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", lib, srcs[0], "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    return lib
