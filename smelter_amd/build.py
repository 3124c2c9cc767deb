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


def _elf_sections(data: bytes) -> dict:
    import struct
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise ValueError("not an ELF64 image")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)

    def sect(i):
        name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size
    _n, stroff, _strsize = sect(shstrndx)
    out = {}
    for i in range(shnum):
        name, off, size = sect(i)
        end = data.index(b"\0", stroff + name)
        out[data[stroff + name:end].decode()] = (off, size)
    return out


def fatbin_sha256(lib: str = LIB) -> str:
    """sha256 of the library's .hip_fatbin section as it is.  NOT an identity of the device code across checkouts: the offload bundles carry
    a compilation-unit id derived from the source file's absolute path (the same sources built in another directory hash differently)."""
    import hashlib
    with open(lib, "rb") as f:
        data = f.read()
    off, size = _elf_sections(data)[".hip_fatbin"]
    return hashlib.sha256(data[off:off + size]).hexdigest()


def kernels_sha256(lib: str = LIB) -> str:
    """Identity of the library's DEVICE code: sha256 over the machine code (.text) and the kernel descriptors / constants (.rodata) of every
    gfx950 code object in its .hip_fatbin section, in order — what the GPU executes, and nothing that depends on where the sources were built
    (the same sources give the same value in any directory; `fatbin_sha256` does not).  Profiles that quote per-kernel counters record it
    (tools/traffic_json.py, tools/issue_json.py); bench.py refuses to quote a profile whose hash is not the hash of the library it is timing."""
    import hashlib
    import struct
    with open(lib, "rb") as f:
        data = f.read()
    secs = _elf_sections(data)
    if ".hip_fatbin" not in secs:
        raise ValueError(f"{lib}: no .hip_fatbin section")
    off, size = secs[".hip_fatbin"]
    fb = data[off:off + size]
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    h = hashlib.sha256()
    pos, objects = 0, 0
    while True:
        pos = fb.find(magic, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", fb, pos + len(magic))
        p = pos + len(magic) + 8
        for _ in range(n):
            eoff, esize, tlen = struct.unpack_from("<QQQ", fb, p)
            p += 24
            triple = fb[p:p + tlen]
            p += tlen
            if b"amdgcn" in triple and esize:
                elf = fb[pos + eoff:pos + eoff + esize]
                es = _elf_sections(elf)
                for name in (".text", ".rodata"):
                    if name in es:
                        h.update(name.encode())
                        h.update(elf[es[name][0]:es[name][0] + es[name][1]])
                objects += 1
        pos += len(magic)
    if not objects:
        raise ValueError(f"{lib}: no gfx code object in .hip_fatbin (a compressed bundle?)")
    return h.hexdigest()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
