"""CPU: what the oracle's documented deviations from PCL 1.12.1's arithmetic do to the north-star observables.

The oracle has a measurement-only ARITH_PCL mode (oracle/src/orc_features.cpp: pcl::eigen33 closed-form normals, libm acosf swap
test and atan2f in computePairFeatures, weightPointSPFHSignature's neighbour order / rounding steps).  The BASELINE configs[1]
profile is run in both modes on one synthetic pair (tools/pcl_order_report.py) and the stage-by-stage differences are bounded here;
DESIGN.md section 6 tables the measured values at 100 k and 1 M points.

VERDICT r2 asked for |dT| <= 1e-4 between the two modes.  That bound does NOT hold and cannot: RANSAC's sample stream maps its
draws onto the correspondence LIST (index = draw mod C), so ONE correspondence more or fewer re-deals every triple; already the
weighting-order piece alone (2 of 200 000 match indices, 1 of 16 166 correspondences differ) moves the result by 3e-3 -- the same
size as either result's distance to the ground truth (6e-3 / 8e-3, noise 5 mm).  What is asserted: every stage differs only at its
rounding level, the correspondence sets overlap by > 99 %, and both modes register the pair equally well."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(*argv):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pcl_order_report.py"), *argv], text=True)
    return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


@pytest.fixture(scope="module")
def rep():
    return report("--points", "60000")


def test_stage_differences_are_at_rounding_level(rep):
    for side in ("src", "tgt"):
        n = rep["normals_" + side]
        assert n["differ_in_any_bit"] > 0.5 * n["points"]                     # eigen33 and the Jacobi solver really are different code
        assert n["max_abs_component_diff"] < 5e-4 and n["differ_by_more_than_1e-5"] < 1e-3 * n["points"]
        assert n["max_abs_curvature_diff"] < 1e-5
        f = rep["fpfh_" + side]
        assert f["nan_rows_differ"] == 0
        assert f["mean_abs_bin_diff"] < 2e-3                                  # of bins that sum to 100 per block
        assert f["rows_with_a_bin_moved_by_more_than_0.5"] < 6e-3 * f["rows"]   # a pair feature on a bin edge changes one histogram count
    m = rep["match"]
    assert m["src_to_tgt_indices_differ"] + m["tgt_to_src_indices_differ"] < 0.01 * m["queries"]
    c = rep["correspondences"]
    assert c["in_both"] > 0.985 * max(c["canonical"], c["pcl_order"])


def test_both_arithmetics_register_the_pair_equally_well(rep):
    r = rep["ransac"]
    assert r["canonical"]["converged"] == 1 and r["pcl_order"]["converged"] == 1
    e = r["max_abs_err_vs_ground_truth"]
    assert e["canonical"] < 0.03 and e["pcl_order"] < 0.03
    assert r["max_abs_dT"] < 0.03                                             # not 1e-4: see the module docstring
    assert abs(r["canonical"]["inliers"] - r["pcl_order"]["inliers"]) < 0.05 * r["canonical"]["inliers"]


def test_weighting_order_alone_changes_almost_nothing_but_still_moves_T():
    """the deviation that puts the FPFH weighting on the matrix cores (one fmaf chain in grid order instead of PCL's ascending-distance
    mul-then-add): every row differs in some bit, no bin by more than 1e-3 of its block, a handful of matches -- and RANSAC still
    lands on a different (equally good) transform, which is the sampling sensitivity named above."""
    rep = report("--points", "60000", "--mode", "4")
    for side in ("src", "tgt"):
        f = rep["fpfh_" + side]
        assert f["rows_with_a_bin_moved_by_more_than_0.5"] == 0 and f["max_abs_bin_diff"] < 1e-2
        assert rep["normals_" + side]["differ_in_any_bit"] == 0
    m = rep["match"]
    assert m["src_to_tgt_indices_differ"] + m["tgt_to_src_indices_differ"] < 1e-3 * m["queries"]
    assert rep["ransac"]["max_abs_err_vs_ground_truth"]["pcl_order"] < 0.03
