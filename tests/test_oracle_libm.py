"""CPU: oracle/src/orc_libm.h -- glibc 2.35's float acosf / atanf / atan2f / sinf / cosf restated op for op -- against the libm.so.6 this
process runs on.  PCL 1.12.1 calls exactly these in pcl::computePairFeatures (features/src/pfh.cpp: acos of |angle| for the swap test,
atan2 for f1; call site include/common.h:326-331) and in pcl::computeRoots under pcl::eigen33 (common/impl/eigen.hpp: atan2, cos, sin;
call site src/common.cpp:646-653).  "PCL's own arithmetic" (ORC_ARITH_PCL, LGR_ARITH_PCL) is defined with THESE routines; this file
is the pin: on a glibc 2.35 host every comparison below is an equality of bits.  On another libm the restatement is still what both
sides of the GPU parity tests use; the comparison is then skipped (the named version is the definition, not the host's)."""
import ctypes
import platform

import numpy as np
import pytest

import oracle as o

GLIBC = platform.libc_ver()
pytestmark = pytest.mark.skipif(GLIBC[0] != "glibc" or GLIBC[1] not in ("2.35", "2.31"), reason="the restatement names glibc 2.35 (2.31 carries the same sources); this host runs %s %s" % GLIBC)

ONE = 0x3f800000


def test_acosf_every_float_of_minus_one_to_one():
    assert o.libm_check_range(o.LIBM_ACOSF, 0, ONE) == 0                      # [+0, 1]: the swap test's arguments are |angle|
    assert o.libm_check_range(o.LIBM_ACOSF, 0x80000000, 0x80000000 + ONE) == 0   # [-1, -0]
    assert o.libm_check_range(o.LIBM_ACOSF, ONE + 1, ONE + 4096) == 0         # just above 1: NaN on both sides


def test_atanf_every_float():
    assert o.libm_check_range(o.LIBM_ATANF, 0, 0xffffffff) == 0


def test_sinf_cosf_every_float_of_the_range_eigen33_uses():
    # theta = atan2(sqrt(-q), half_b) / 3 lies in [0, pi / 3]; checked on [0, 2] and [-2, -0] (the x86-64 ifunc picks an FMA build of
    # these two on this CPU: the restatement is the plain build and still agrees on every input)
    for fn in (o.LIBM_SINF, o.LIBM_COSF):
        assert o.libm_check_range(fn, 0, 0x40000000) == 0
        assert o.libm_check_range(fn, 0x80000000, 0xc0000000) == 0


def test_atan2f_samples_and_special_cases():
    rng = np.random.default_rng(566)
    n = 40_000_000
    y = rng.uniform(-1.5, 1.5, n).astype(np.float32)
    x = rng.uniform(-1.5, 1.5, n).astype(np.float32)
    y[::8] *= np.float32(1e-4); x[3::16] *= np.float32(1e-5)
    x[5::1024] = 0; y[7::1024] = 0; x[9::1024] = 1; x[11::1024] = -0.0; y[13::1024] = -0.0
    assert o.libm_check_atan2(y, x) == 0
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-38, -1e-38, 3e38, -3e38, 1e-45, 2.0 ** 61, 2.0 ** -61], np.float32)
    yy, xx = [a.ravel() for a in np.meshgrid(sp, sp)]
    assert o.libm_check_atan2(yy, xx) == 0
    # eigen33's call: atan2(sqrt(-q) >= 0, half_b), |arguments| <= ~1e-1 down to denormals
    y = np.abs(rng.standard_normal(n).astype(np.float32)) * np.float32(10.0) ** rng.integers(-30, 0, n).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 0, n).astype(np.float32)
    assert o.libm_check_atan2(y, x) == 0


def test_eval_matches_the_running_libm_elementwise():
    libm = ctypes.CDLL("libm.so.6")
    libm.acosf.restype = ctypes.c_float; libm.acosf.argtypes = [ctypes.c_float]
    xs = np.linspace(0, 1, 1001, dtype=np.float32)
    got = o.libm_eval(o.LIBM_ACOSF, xs)
    want = np.array([libm.acosf(float(v)) for v in xs], np.float32)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
