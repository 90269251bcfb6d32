"""lgr_seqsum (csrc/lgr_seqsum.h: the sequential float sum of n copies of an increment without the loop -- the value of a finished SPFH
bin) against the loop itself, on the CPU: tests/cpp/seqsum_test.cpp is compiled with g++ -ffp-contract=off (the header is
host / device code) and walks every FPFH increment 100 / (k - 1), k - 1 <= 1500, with every count, plus 200 000 random increments
(ties included)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_seqsum_equals_the_loop(tmp_path):
    exe = str(tmp_path / "seqsum_test")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "cpp", "seqsum_test.cpp")])
    out = subprocess.run([exe, "1500"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
