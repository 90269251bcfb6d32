"""CPU: the distance between the arithmetic the HIP library restates and PCL 1.12.1's own, piece by piece.

The oracle's arithmetic mode is a bit mask (oracle/lgr_oracle.h): eigen33 normals, acosf swap test, atan2f, and the three pieces of
pcl::FPFHEstimation::weightPointSPFHSignature (neighbour order, rounded product, running normaliser).  tools/pcl_order_report.py runs the
BASELINE configs[1] profile in two masks on one synthetic pair and reports the stage-by-stage differences;
profiles/r5_pcl_order_by_piece_*.json tables every piece alone at 100 k and 1 M points.

Rounds 1-4 restated NONE of the pieces (mask 0): 0.5 % of the match indices and |dT| ~ 1e-2 away from PCL's arithmetic at 1 M -- and the
by-piece table shows that the Jacobi normals and the argument-comparison swap test carried all of it.  Since round 5 the default mode
(ARITH_CANONICAL = eigen33 | acosf | atan2f, with glibc 2.35's float routines restated op for op: tests/test_oracle_libm.py) IS PCL's
arithmetic for the normals and pair features; what is left is the weighting's rounding (the fused chain that runs on the matrix cores),
and LGR_ARITH_PCL removes that too (tests/test_gpu_pcl_arith.py: HIP(PCL) == oracle(PCL) bit for bit).  Asserted here, at 60 k points:
  * default vs PCL: normals identical, no FPFH bin moves by 1e-2, a handful of match indices at most, both register the pair;
  * rounds 1-4's mask 0 vs PCL: the old rounding-level bounds still hold (the measurement mode is kept);
  * |dT| <= 1e-4 between two arithmetics cannot be promised in general (ONE different correspondence re-deals every RANSAC triple, because the
    sample stream indexes the correspondence LIST), which is why parity is claimed mode by mode against the oracle, never across modes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(*argv):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pcl_order_report.py"), *argv], text=True, stderr=subprocess.DEVNULL)
    return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


@pytest.fixture(scope="module")
def rep_default():
    return report("--points", "60000")                    # base = ARITH_CANONICAL (round-5 default), other = ARITH_PCL


@pytest.fixture(scope="module")
def rep_round4():
    return report("--points", "60000", "--base", "0")     # base = rounds 1-4's canonical orders, other = ARITH_PCL


def test_default_mode_is_pcl_arithmetic_up_to_the_weightings_rounding(rep_default):
    rep = rep_default
    assert rep["base_mode_bits"] == 35 and all(rep["pcl_order_pieces"].values())
    for side in ("src", "tgt"):
        assert rep["normals_" + side]["differ_in_any_bit"] == 0           # eigen33 in both
        f = rep["fpfh_" + side]
        assert f["nan_rows_differ"] == 0 and f["rows_with_a_bin_moved_by_more_than_0.5"] == 0 and f["max_abs_bin_diff"] < 1e-2
        assert f["rows_differ_in_any_bit"] > 0.9 * f["rows"]              # ... and the weighting really is a different rounding sequence
    m = rep["match"]
    assert m["src_to_tgt_indices_differ"] + m["tgt_to_src_indices_differ"] <= 1e-4 * m["queries"]
    c = rep["correspondences"]
    assert c["only_canonical"] + c["only_pcl_order"] <= 2
    r = rep["ransac"]
    assert r["canonical"]["converged"] == 1 and r["pcl_order"]["converged"] == 1
    e = r["max_abs_err_vs_ground_truth"]
    assert e["canonical"] < 0.03 and e["pcl_order"] < 0.03


def test_round4_orders_differ_from_pcl_at_rounding_level_only(rep_round4):
    rep = rep_round4
    for side in ("src", "tgt"):
        n = rep["normals_" + side]
        assert n["differ_in_any_bit"] > 0.5 * n["points"]                     # eigen33 and the Jacobi solver really are different code
        assert n["max_abs_component_diff"] < 5e-4 and n["differ_by_more_than_1e-5"] < 1e-3 * n["points"]
        assert n["max_abs_curvature_diff"] < 1e-5
        f = rep["fpfh_" + side]
        assert f["nan_rows_differ"] == 0
        assert f["mean_abs_bin_diff"] < 2e-3                                  # of bins that sum to 100 per block
        assert f["rows_with_a_bin_moved_by_more_than_0.5"] < 6e-3 * f["rows"]   # a pair feature on a bin edge / a swapped pair changes histogram counts
    m = rep["match"]
    assert m["src_to_tgt_indices_differ"] + m["tgt_to_src_indices_differ"] < 0.01 * m["queries"]
    c = rep["correspondences"]
    assert c["in_both"] > 0.985 * max(c["canonical"], c["pcl_order"])
    r = rep["ransac"]
    assert r["canonical"]["converged"] == 1 and r["pcl_order"]["converged"] == 1
    e = r["max_abs_err_vs_ground_truth"]
    assert e["canonical"] < 0.03 and e["pcl_order"] < 0.03
    assert r["max_abs_dT"] < 0.03                                             # not 1e-4: see the module docstring


def test_default_is_much_closer_to_pcl_than_round4_was(rep_default, rep_round4):
    d = rep_default["match"]; r = rep_round4["match"]
    assert 20 * (d["src_to_tgt_indices_differ"] + d["tgt_to_src_indices_differ"]) < r["src_to_tgt_indices_differ"] + r["tgt_to_src_indices_differ"]
