"""On-disk formats: PLY as loadPLYFile<PointN> reads it, the reference's CSV schemas."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lidar-global-registration_amd"))
from lgr_amd import formats, synthetic  # noqa: E402


def test_ply_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    pts = synthetic.make_points(rng.normal(size=(500, 3)), intensity=2.5)
    pts[:, 4:7] = rng.normal(size=(500, 3))
    pts[:, 9] = rng.uniform(0, 1, 500)
    for binary in (True, False):
        p = str(tmp_path / ("a_%d.ply" % binary))
        formats.write_ply(p, pts, binary=binary)
        got, fields = formats.read_ply(p)
        assert np.array_equal(got.view(np.uint32), pts.view(np.uint32))
        assert formats.has_normals(fields) and "intensity" in fields
    p = str(tmp_path / "nonormals.ply")
    formats.write_ply(p, pts, with_normals=False)
    got, fields = formats.read_ply(p)
    assert not formats.has_normals(fields) and np.all(got[:, 4:7] == 0) and np.array_equal(got[:, :3], pts[:, :3])


def test_ply_foreign_layouts(tmp_path):
    """double coordinates, short-named normals, extra colour properties, a face element with a list, big endian."""
    p = str(tmp_path / "ascii.ply")
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty double x\nproperty double y\nproperty double z\n"
                "property uchar red\nproperty float nx\nproperty float ny\nproperty float nz\nproperty float scalar_intensity\n"
                "element face 1\nproperty list uchar int vertex_indices\nend_header\n"
                "1.5 2.5 -3.5 255 0 0 1 7\n-1 0 1e-3 0 1 0 0 9\n3 0 1 1\n")
    pts, fields = formats.read_ply(p)
    assert pts.shape == (2, 12) and np.allclose(pts[0, :3], [1.5, 2.5, -3.5]) and pts[0, 3] == 1
    assert np.array_equal(pts[:, 4:7], [[0, 0, 1], [1, 0, 0]]) and np.array_equal(pts[:, 8], [7, 9]) and formats.has_normals(fields)
    p = str(tmp_path / "be.ply")
    with open(p, "wb") as f:
        f.write(b"ply\nformat binary_big_endian 1.0\nelement face 1\nproperty list uchar int vertex_indices\nelement vertex 2\n"
                b"property float x\nproperty float y\nproperty float z\nproperty ushort label\nend_header\n")
        f.write(bytes([3]) + np.array([0, 1, 1], ">i4").tobytes())
        f.write(np.array([(1, 2, 3, 4), (5, 6, 7, 8)], dtype=[("x", ">f4"), ("y", ">f4"), ("z", ">f4"), ("l", ">u2")]).tobytes())
    pts, fields = formats.read_ply(p)
    assert np.array_equal(pts[:, :3], [[1, 2, 3], [5, 6, 7]]) and fields == ["x", "y", "z"]
    with pytest.raises(ValueError):
        bad = str(tmp_path / "bad.ply")
        open(bad, "w").write("plx\n")
        formats.read_ply(bad)


def test_transformation_csv(tmp_path):
    p = str(tmp_path / "t.csv")
    T1 = np.arange(16, dtype=np.float32).reshape(4, 4) / 3
    T2 = synthetic.random_se3(np.random.default_rng(1))
    formats.save_transformation(p, "a.ply", T1)
    formats.save_transformation(p, "b.ply", T2)
    lines = open(p).read().splitlines()
    assert lines[0] == formats.TRANSFORMATION_HEADER and len(lines) == 3
    assert lines[1].startswith("a.ply,0,0.333333,0.666667,1,")          # 6 significant digits like ostream << float
    assert np.allclose(formats.get_transformation(p, "b.ply"), T2, atol=1e-5)
    rel = formats.get_relative_transformation(p, "b.ply", "b.ply")
    assert np.allclose(rel, np.eye(4), atol=1e-5)
    assert formats.get_relative_transformation(p, "b.ply", "missing.ply") is None
    with pytest.raises(KeyError):
        formats.get_transformation(p, "missing")


def test_correspondences_csv(tmp_path):
    pr = synthetic.make_correspondence_problem(n_pts=200, c=50, seed=3)
    p = str(tmp_path / "c.csv")
    formats.save_correspondences(p, pr["src"], pr["tgt"], pr["corr"])
    lines = open(p).read().splitlines()
    assert lines[0] == formats.CORRESPONDENCES_HEADER and len(lines) == 51 and lines[1].count(",") == 9
    back = formats.read_correspondences(p)
    assert np.array_equal(back["index_query"], pr["corr"]["index_query"]) and np.array_equal(back["index_match"], pr["corr"]["index_match"])
    assert np.allclose(back["distance"], pr["corr"]["distance"], rtol=1e-5) and np.allclose(back["threshold"], 0.05)


def test_results_row():
    cols = formats.RESULTS_HEADER.split(",")
    assert len(cols) == 38 and cols[0] == "version" and cols[-1] == "converged"
    row = formats.results_row(version="x", descriptor="fpfh", testname="a_b", metric=0.5, correspondences=120, iteration=1000, converged=True,
                              distance_thr=0.1, time_cs=0.081, time_te=0.009)
    tok = row.split(",")
    assert len(tok) == 38 and tok[1] == "fpfh" and tok[3] == "0.5" and tok[-1] == "1" and tok[cols.index("feature_radius")] == ""
    with pytest.raises(KeyError):
        formats.results_row(nope=1)
