"""CPU: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer.  tests/cpp/oracle_sanitize.cpp compiles the oracle's
translation units into one program with -fsanitize=address,undefined -fno-sanitize-recover and drives every stage of the path (all three
arithmetic modes, all match filters, every RANSAC metric, GROR, the edge inputs of the parity tests) on a small pair; any report aborts.
GPU AddressSanitizer is not available on the MI355X pool, so the CPU side -- the checker of every parity claim -- is where sanitizers run."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_and_ubsan(tmp_path):
    exe = os.path.join(str(tmp_path), "oracle_sanitize")
    src = [os.path.join(ROOT, "oracle", "src", f) for f in sorted(os.listdir(os.path.join(ROOT, "oracle", "src"))) if f.endswith(".cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "cpp", "oracle_sanitize.cpp"), *src, "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    assert "oracle_sanitize: 0 failures" in out.stdout
