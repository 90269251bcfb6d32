"""The header-only C++ shim (host/lgr_compat.hpp) that keeps the reference's call surface: it must compile and link
against liblgr_hip.so with plain g++ (CPU check), and run end to end on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lidar-global-registration_amd", "csrc")


def build(tmp_path):
    exe = os.path.join(str(tmp_path), "shim_smoke")
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "shim_smoke.cpp"), "-o", exe,
                           "-L", CSRC, "-llgr_hip", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def test_shim_compiles_and_links(tmp_path):
    assert os.path.exists(build(tmp_path))


@pytest.mark.gpu
def test_shim_runs_alignPointClouds(tmp_path):
    exe = build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "converged=1" in out.stdout
