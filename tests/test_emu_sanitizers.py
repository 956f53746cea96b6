"""The kernel library's own sources compiled for the HOST and run under AddressSanitizer + UndefinedBehaviorSanitizer
(tests/emu: a stand-in for <hip/hip_runtime.h>, one active lane per wave, blocks and threads one after the other).
Not a CPU path of the product -- the product has none -- but a sanitizer harness for the kernels' indexing: every
out-of-bounds access to a per-lane array (frame stack, level state, sorted lists, walk stacks), to the ray queues, the
node pool or a table shows up here as a report instead of a GPU memory fault.  The driver renders the scene with the
one-launch kernel and with a launch per generation of rays, counting variants included, at several queue / node
capacities, and compares the frames bit for bit.  The kernel pipeline of large scenes (rt_wf2.h: ballots, shuffles, a
wave's records put together in LDS) runs with 64 host threads per wave in lockstep, every collective a rendezvous."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG) or shutil.which("make") is None:
        pytest.skip("no host clang++ to build the emulation with")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "emu"), "-j8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return os.path.join(ROOT, "build", "emu", "emu_main")


@pytest.mark.parametrize("args", [("synthetic", "24", "16", "1500", "quick"), ("csg", "48", "27")])
def test_kernels_are_clean_under_asan_and_ubsan(emu, args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([emu, *args], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
