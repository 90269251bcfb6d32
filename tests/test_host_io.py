"""The C++ host's format header (lidar-global-registration_amd/host/lgr_io.hpp: loadPLYFile / savePLYFile* / CSVRow / split /
getTransformation / saveTransformation / readCorrespondencesFromCSV / saveCorrespondencesToCSV, the reference's names) against
the Python host's (lgr_amd/formats.py): files written by one are read by the other and re-written byte for byte, and the C++
CSVRow is held to the reference's own tokeniser (oracle/_ref, when built) on the same lines.  CPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
from lgr_amd import formats, synthetic  # noqa: E402

CSV_LINES = ("reading,gT00,gT01\n" "a.ply,1,2\n" "b.ply,,3,\n" ",,\n" "\n" "quoted \"x,y\",7\n" "crlf,1,2\r\n" "last,no,newline")


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    d = tmp_path_factory.mktemp("io")
    exe = str(d / "io_roundtrip")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "cpp", "io_roundtrip.cpp"), "-o", exe])
    rng = np.random.default_rng(5)
    pts = synthetic.make_points(rng.normal(size=(70001, 3)) * 40.0, intensity=2.5)     # > one 65536-point read chunk
    pts[:, 4:7] = rng.normal(size=(70001, 3))
    pts[:, 9] = rng.uniform(0, 1, 70001)
    pts[0, :3] = [1e-7, -3.4e38, 1.17549435e-38]
    formats.write_ply(str(d / "py_bin.ply"), pts, binary=True)
    formats.write_ply(str(d / "py_ascii.ply"), pts[:3000], binary=False)
    with open(d / "foreign_ascii.ply", "w") as f:      # double coordinates, short-named normals, colour, a face element with a list
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty double x\nproperty double y\nproperty double z\n"
                "property uchar red\nproperty float nx\nproperty float ny\nproperty float nz\nproperty float scalar_intensity\n"
                "element face 1\nproperty list uchar int vertex_indices\nend_header\n"
                "1.5 2.5 -3.5 255 0 0 1 7\n-1 0 1e-3 0 1 0 0 9\n3 0 1 1\n")
    with open(d / "foreign_be.ply", "wb") as f:        # big endian, a list element before the vertices, an integer label
        f.write(b"ply\nformat binary_big_endian 1.0\nelement face 1\nproperty list uchar int vertex_indices\nelement vertex 2\n"
                b"property float x\nproperty float y\nproperty float z\nproperty ushort label\nproperty double curvature\nend_header\n")
        f.write(bytes([3]) + np.array([0, 1, 1], ">i4").tobytes())
        f.write(np.array([(1, 2, 3, 4, 0.25), (5, 6, 7, 8, 0.5)], dtype=[("x", ">f4"), ("y", ">f4"), ("z", ">f4"), ("l", ">u2"), ("c", ">f8")]).tobytes())
    A = synthetic.random_se3(np.random.default_rng(1))
    B = synthetic.random_se3(np.random.default_rng(2))
    formats.save_transformation(str(d / "py_t.csv"), "a.ply", A)
    formats.save_transformation(str(d / "py_t.csv"), "b.ply", B)
    corr = np.zeros(500, dtype=[("index_query", "<i4"), ("index_match", "<i4"), ("distance", "<f4"), ("threshold", "<f4")])
    corr["index_query"] = rng.integers(0, 70001, 500)
    corr["index_match"] = rng.integers(0, 70001, 500)
    corr["distance"] = rng.uniform(0, 3000, 500)
    corr["threshold"] = rng.uniform(0, 1, 500)
    formats.save_correspondences(str(d / "py_corr.csv"), pts, pts, corr)
    (d / "lines.txt").write_bytes(CSV_LINES.encode())
    out = subprocess.run([exe, str(d)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    return d, out.stdout, pts


def test_ply_written_by_python_is_reread_and_rewritten_identically(run):
    d, log, pts = run
    assert "py_bin: 70001 points, normals=1" in log and "py_ascii: 3000 points, normals=1" in log
    assert "missing file -> -1, not a ply -> -1" in log
    # binary: the same bytes back; ascii: the same text back (both hosts print 9 significant digits)
    assert (d / "cpp_py_bin_bin.ply").read_bytes() == (d / "py_bin.ply").read_bytes()
    assert (d / "cpp_py_ascii_ascii.ply").read_bytes() == (d / "py_ascii.ply").read_bytes()
    # and across: what C++ wrote, python reads bit for bit
    got, fields = formats.read_ply(str(d / "cpp_py_bin_ascii.ply"))
    assert np.array_equal(got.view(np.uint32), pts.view(np.uint32)) and formats.has_normals(fields)


def test_foreign_ply_layouts_agree(run):
    d, log, _ = run
    assert "foreign_ascii: 2 points, normals=1, fields=x y z normal_x normal_y normal_z intensity" in log
    assert "foreign_be: 2 points, normals=0, fields=x y z curvature" in log
    for name in ("foreign_ascii", "foreign_be"):
        want, _ = formats.read_ply(str(d / (name + ".ply")))
        got, _ = formats.read_ply(str(d / ("cpp_%s_bin.ply" % name)))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    want, _ = formats.read_ply(str(d / "foreign_be.ply"))
    assert np.array_equal(want[:, 9], [0.25, 0.5]) and np.array_equal(want[:, :3], [[1, 2, 3], [5, 6, 7]])


def test_csv_files_agree(run):
    d, _, _ = run
    py = (d / "py_t.csv").read_text().splitlines()
    cpp = (d / "cpp_t.csv").read_text().splitlines()
    assert cpp[:3] == py[:3] and cpp[0] == formats.TRANSFORMATION_HEADER            # text -> float -> the same text
    rel = np.array([float(v) for v in cpp[3].split(",")[1:]], np.float32).reshape(4, 4)
    want = formats.get_relative_transformation(str(d / "py_t.csv"), "a.ply", "b.ply")
    np.testing.assert_allclose(rel, want, rtol=0, atol=2e-5)                        # float cofactor inverse vs double, then 6 digits
    assert (d / "cpp_corr.csv").read_bytes() == (d / "py_corr.csv").read_bytes()
    back = formats.read_correspondences(str(d / "cpp_corr.csv"))
    assert len(back) == 500


def test_cpp_csvrow_cuts_lines_like_python_and_the_reference(run, oracle):
    d, _, _ = run
    got = [ln.split("|")[1:] for ln in (d / "cpp_tokens.txt").read_bytes().decode().split("\n")[:-1]]
    counts = [int(ln.split("|")[0]) for ln in (d / "cpp_tokens.txt").read_bytes().decode().split("\n")[:-1]]
    with open(d / "lines.txt", newline="\n") as f:
        want = [formats.csv_row(ln) for ln in f]
    assert got == want and counts == [len(r) for r in want]
    ref = oracle.ref_utils()
    if ref is not None:     # the reference's own CSVRow, where oracle/_ref is built
        import ctypes as C
        buf = C.create_string_buffer(1 << 16)
        assert ref.ref_csv_rows(CSV_LINES.encode(), buf, len(buf)) > 0
        assert [row.split("\x1f") for row in buf.value.decode().split("\x1e")] == got
