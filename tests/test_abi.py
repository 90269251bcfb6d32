"""The C-ABI library loads and exports every symbol include/lgr.h declares; struct layouts seen by the ctypes
binding equal the C compiler's.  No compute call is made (this file runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lgr.h")


@pytest.fixture(scope="module")
def capi():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    from lgr_amd import capi
    return capi


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lgr_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(capi):
    lib = capi.lib()
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_layout_matches_c_compiler(capi, tmp_path):
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lgr.h"\nint main(){'
                    'printf("%zu %zu %zu ", sizeof(lgr_params), sizeof(lgr_result), sizeof(lgr_corr));'
                    'printf("%zu %zu %zu %zu ", offsetof(lgr_params, alignment_id), offsetof(lgr_params, vp_src), offsetof(lgr_params, ransac_batch), offsetof(lgr_params, seed));'
                    'printf("%zu %zu %zu\\n", offsetof(lgr_result, n_correspondences), offsetof(lgr_result, time_cs), offsetof(lgr_result, stage_ms));return 0;}')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    P, R = capi.Params, capi.Result
    want = [C.sizeof(P), C.sizeof(R), 16, P.alignment_id.offset, P.vp_src.offset, P.ransac_batch.offset, P.seed.offset,
            R.n_correspondences.offset, R.time_cs.offset, R.stage_ms.offset]
    assert got == want
    assert capi.CORR_DTYPE.itemsize == 16


def test_host_only_entry_points(capi):
    lib = capi.lib()
    assert lib.lgr_version() == 5
    p = capi.default_params()
    # defaults of src/common.cpp:216-223, 335-413 / include/common.h:38-57
    assert (p.feature_nr_points, p.normal_nr_points, p.bf_block_size, p.cluster_k, p.randomness, p.n_samples) == (352, 30, 10000, 40, 1, 3)
    assert abs(p.edge_thr_coef - 0.95) < 1e-7 and abs(p.confidence - 0.999) < 1e-7 and p.scale_factor == 2.0
    assert (p.matching_id, p.metric_id, p.score_id, p.alignment_id) == (capi.MATCH_CLUSTER, capi.METRIC_UNIFORMITY, capi.SCORE_MSE, capi.ALIGN_RANSAC)
    assert p.fix_seed == 1 and p.seed == 566
    # invalid arguments are reported, not crashed on (no device is touched)
    assert lib.lgr_ctx_create(0, None, None) == -1
    assert lib.lgr_ctx_sync(None) == -1 and lib.lgr_match_bf_dev(None, None, 0, None, 0, 1, None, None) == -1


def test_update_hypotheses_matches_oracle(capi, oracle):
    """include/hypotheses.h:10-12 is host bookkeeping in the product too; same decisions as the oracle."""
    import numpy as np
    from lgr_amd import synthetic
    rng = np.random.default_rng(3)
    buf = np.zeros((32, 16), np.float32); met = np.zeros(32, np.float32); n = 0
    otn, om = [], []
    for step in range(40):
        T = synthetic.random_se3(rng) if step % 3 else np.eye(4) + np.diag([0, 0, 0, 0])
        if step % 3 == 0:
            T = np.eye(4); T[:3, 3] = rng.normal(0, 0.02, 3)
        m = float(rng.uniform(0, 1))
        t16 = np.ascontiguousarray(T.astype(np.float32).T.reshape(16))
        n = capi.lib().lgr_update_hypotheses(buf.ctypes.data_as(C.c_void_p), met.ctypes.data_as(C.c_void_p), n, 32,
                                             t16.ctypes.data_as(C.c_void_p), C.c_float(m), C.c_float(0.1))
        assert n >= 0
        otn, om = oracle.update_hypotheses(otn, om, T, m, 0.1, cap=32)
        assert n == len(om)
        np.testing.assert_allclose(met[:n], om, rtol=0, atol=0)
