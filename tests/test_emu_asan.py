"""The three kernels of the default route on the lane emulator, INSTRUMENTED: tests/emu built with -fsanitize=address,undefined (tests/emu_build.py,
SMR_EMU_ASAN) and the parity tests rerun in child processes under the compiler's AddressSanitizer runtime.  What guard pages
(emu_guard.h) cannot see, red zones do: an index past an LDS array (function-local statics here) or past the dynamic LDS block, a
register array indexed out of range, a read one element past a weight band — the emulated "device" buffers at their exact sizes.
(Found when this was written: the converter reads — and ignores — a dword past a chroma row that fills its pitch; frames of such widths
were being kept off the block converter by the host for exactly that reason: tests/test_gpu_parity.py::
test_chroma_rows_that_fill_their_pitch_take_the_block_converter.)  Test infrastructure only."""
import os
import subprocess
import sys

import pytest

from tests import emu_build

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

RUNS = {  # module -> selection (the instrumented build runs several times slower: the slowest parametrisations stay with the plain runs)
    "test_emu_convert.py": "block_converter or runs_of_blocks or equal_shares",
    "test_emu_wave.py": "rgb12 or single_tile or alpha_builds",
    "test_emu_compose.py": "zoo or transition or longer or seams",
}


def test_emulated_kernels_under_address_sanitizer():
    rt = emu_build.asan_runtime()
    if rt is None:
        pytest.skip("the emulator's compiler has no shared AddressSanitizer runtime")
    if os.environ.get("SMR_EMU_ASAN"):
        pytest.skip("this is the inner run")
    env = dict(os.environ, SMR_EMU_ASAN="1", LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    # -s: a report goes to the child's stderr as it is (pytest's capture would swallow it when the process dies)
    children = {mod: subprocess.Popen([sys.executable, "-m", "pytest", os.path.join(HERE, mod), "-x", "-q", "-s", "-p", "no:cacheprovider", "-k", sel],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT) for mod, sel in RUNS.items()}
    for mod, child in children.items():
        out, _ = child.communicate(timeout=2400)
        report = out[out.index("ERROR: AddressSanitizer"):][:4000] if "ERROR: AddressSanitizer" in out else out[-2500:]
        assert child.returncode == 0, f"{mod}: rc {child.returncode}\n{report}"
        assert " passed" in out and "AddressSanitizer" not in out and "runtime error" not in out, f"{mod}\n{report}"
