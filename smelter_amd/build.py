"""Builds libsmr_hip.so (gfx950) in-tree with hipcc.  `python -m smelter_amd.build [--force]`."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libsmr_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the general kernels keep every multiply/add individually rounded like the
# WGSL source (and the oracle); hot loops that want FMA say so with __builtin_fmaf.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
if os.environ.get("SMR_LAB"):  # a laboratory build: A/B knobs from the environment, the fused-conversion builds of k_ingest_wave (tools/variant.sh -DSMR_LAB is the usual way)
    FLAGS.append("-DSMR_LAB")
if os.environ.get("SMR_ABLATION_BUILDS"):  # profiling only: extra instantiations of the ingest kernel with phases compiled out
    FLAGS.append("-DSMR_ABLATION_BUILDS")


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "host", "*.cpp")))


def _deps_mtime():
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "host", "*.h")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    newest = max(os.path.getmtime(src), _deps_mtime())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def kernels_sha256(lib: str = LIB) -> str:
    """Identity of the library's DEVICE code: sha256 of its .hip_fatbin section (the gfx950 code objects of every kernel).  Profiles that
    quote per-kernel counters record it (tools/traffic_json.py, tools/issue_json.py); bench.py refuses to quote a profile whose hash is not
    the hash of the library it is timing."""
    import hashlib
    import struct
    with open(lib, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise ValueError(f"{lib}: not an ELF64 file")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sect(i):
        name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size
    _n, stroff, _strsize = sect(shstrndx)
    for i in range(shnum):
        name, off, size = sect(i)
        end = data.index(b"\0", stroff + name)
        if data[stroff + name:end] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()
    raise ValueError(f"{lib}: no .hip_fatbin section")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
