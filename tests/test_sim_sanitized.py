"""The simulator sources under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile `sanitize`: g++ -O1
-fsanitize=address,undefined -fno-sanitize-recover=all): the whole invariant suite -- both formulations, i.e. the one-env-per-lane
core AND the body-per-lane kernel the product launches, the latter through the host lane emulation -- runs against that build in
a child interpreter with the ASan runtime preloaded.  Any report aborts the child."""
import os
import subprocess
import sys

from conftest import REPO


def test_simulator_sources_are_clean_under_asan_and_ubsan():
    here = os.path.join(REPO, "oracle")
    subprocess.check_call(["make", "-C", here, "sanitize"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["g++", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), "libasan.so of the host compiler not found"
    warn = open(os.path.join(here, "_build", "sim_host_O3_warnings.txt")).read()
    assert "warning" not in warn, warn                   # -O3 -Wall -Wextra -Wuninitialized -Wmaybe-uninitialized is silent
    env = dict(os.environ, PARC_SIM_HOST_LIB=os.path.join(here, "_build", "libparc_sim_host_asan.so"), LD_PRELOAD=libasan,
               # (clear_shadow_mmap_threshold: ASan re-maps the shadow of a fiber's stack on every swapcontext when the stack is large;
               #  the lane emulation switches fibers ~1e6 times, so keep that a memset)
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:clear_shadow_mmap_threshold=1000000000", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_sim_invariants.py"), "-x", "-q", "-p", "no:cacheprovider",
                          "-m", "not gpu", "-k", "not unwritten_work_memory"], capture_output=True, text=True, env=env, cwd=REPO, timeout=600)
    tail = (res.stdout + res.stderr)[-4000:]
    assert res.returncode == 0, tail
    assert "runtime error" not in res.stdout + res.stderr and "AddressSanitizer" not in res.stdout + res.stderr, tail
    assert " passed" in res.stdout
