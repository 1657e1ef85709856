"""CPU-side sanitizer leg ("ASan-enabled host build + parity tests"): the plain-C++ half of the product —
generator drop-in (gen_ethz.cpp), host write-combining split and shard function (hj_host.cpp), the driver's option
parsing and generate-only mode (bench_main.cpp, -DHJ_HOST_ONLY) — and the oracle's C files are rebuilt with
-fsanitize=address,undefined (`make asan`) and the CPU tests that drive them run against those builds in a child
process (libasan preloaded into the interpreter).  Never run on the GPU box: marked as a CPU test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what the sanitizer leg runs: every CPU test that enters gen_ethz.cpp, hj_host.cpp, bench_main.cpp or oracle/*.c
LEG = ["tests/test_generator.py", "tests/test_oracle_gen.py", "tests/test_oracle_join.py",
       "tests/test_abi.py::test_host_write_combining_split_parity", "tests/test_abi.py::test_host_one_pass_block_split_parity",
       "tests/test_abi.py::test_host_radix_join_parity"]


def asan_env():
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    libubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan.so")
    env = dict(os.environ, HJ_ASAN="1", LD_PRELOAD=libasan + (":" + libubsan if os.path.isabs(libubsan) else ""),
               # the interpreter itself is not instrumented and "leaks" by design; everything else is fatal
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=97",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98", OMP_NUM_THREADS="4")
    return env


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "icde2019-gpu-join_amd", "csrc"), "asan"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + LEG,
                       cwd=ROOT, env=asan_env(), capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    log = os.environ.get("HJ_ASAN_LOG")
    if log:
        open(log, "w").write("$ HJ_ASAN=1 LD_PRELOAD=libasan.so:libubsan.so python -m pytest -x -q -m 'not gpu' " + " ".join(LEG) + "\n" + out)
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
    assert r.returncode == 0, out[-6000:]
    assert " passed" in out
