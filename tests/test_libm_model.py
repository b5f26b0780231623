"""The device's logf/sinf/cosf (csrc/vspg_libm.h) must equal the HOST libm bit for bit, because
the CPU reference path calls the host libm and a single ulp re-seeds the shadow-ray RNG.  Here
the same header is compiled for the host and compared with the running libm on >10^7 arguments
drawn from the ranges the path produces.  (GPU build of the same header: test_gpu_parity.py.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def shim(libm_shim):
    return libm_shim


def run(lib, name, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    fp = C.POINTER(C.c_float)
    getattr(lib, name)(x.shape[0], x.ctypes.data_as(fp), y.ctypes.data_as(fp))
    return y


def check(lib, fn, x):
    a, b = run(lib, "model_" + fn, x), run(lib, "libm_" + fn, x)
    bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
    assert bad.size == 0, "%s: %d mismatches, e.g. x=%r model=%r libm=%r" % (
        fn, bad.size, x[bad[:3]], a[bad[:3]], b[bad[:3]])


def test_logf_equals_host_libm(shim):
    rng = np.random.default_rng(0)
    # SampleExponential: log(1-u), u = k*2^-32 rounded to float; resampling: log(1-vsp)
    u = np.minimum((rng.integers(0, 2 ** 32, 6_000_000, dtype=np.uint64).astype(np.float32) * np.float32(2.0 ** -32)),
                   np.float32(float.fromhex('0x1.fffffep-1')))
    check(shim, "logf", np.float32(1) - u)
    check(shim, "logf", rng.uniform(1e-3, 1.0, 2_000_000).astype(np.float32))
    check(shim, "logf", np.exp(rng.uniform(-80, 80, 2_000_000)).astype(np.float32))
    # every float in a few binades around 1 where the table index changes
    bits = np.arange(0x3f000000, 0x3f000000 + 3_000_000, dtype=np.uint32)
    check(shim, "logf", bits.view(np.float32))
    check(shim, "logf", np.array([1.0, 0.5, 2.0, 2.0 ** -126, 3.4e38, float.fromhex('0x1.fffffep-1')], dtype=np.float32))


def test_double_log_equals_host_libm(shim):
    """log_host_exact (double) vs the running libm's log: the arguments the path produces are
    1.0 - (double)u with u a float in [0, 1); plus both branches of the algorithm on wide ranges."""
    rng = np.random.default_rng(3)
    dp = C.POINTER(C.c_double)

    def check_d(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        a, b = np.empty_like(x), np.empty_like(x)
        shim.model_log(x.shape[0], x.ctypes.data_as(dp), a.ctypes.data_as(dp))
        shim.libm_log(x.shape[0], x.ctypes.data_as(dp), b.ctypes.data_as(dp))
        bad = np.nonzero(a.view(np.uint64) != b.view(np.uint64))[0]
        assert bad.size == 0, "log: %d mismatches, e.g. x=%r model=%r libm=%r" % (bad.size, x[bad[:3]], a[bad[:3]], b[bad[:3]])

    u = np.minimum(rng.integers(0, 2 ** 32, 4_000_000, dtype=np.uint64).astype(np.float32) * np.float32(2.0 ** -32),
                   np.float32(float.fromhex('0x1.fffffep-1')))
    check_d(1.0 - u.astype(np.float64))
    # u * tpStep products: small arguments -> the near-1 branch
    check_d(1.0 - (u * rng.random(u.shape[0], dtype=np.float32)).astype(np.float64))
    check_d(1.0 - np.ldexp(rng.random(1_000_000), rng.integers(-24, 0, 1_000_000)).astype(np.float32).astype(np.float64))
    # every float u in [0, 2^-10) step 7, and the floats just below 1
    bits = np.arange(0, 0x3a800000, 7 * 4093, dtype=np.uint32)
    check_d(1.0 - bits.view(np.float32).astype(np.float64))
    bits = np.arange(0x3f7fffff - 2_000_000, 0x3f800000, dtype=np.uint32)
    check_d(1.0 - bits.view(np.float32).astype(np.float64))
    # generic positive normal doubles, both branches and the branch boundaries
    check_d(np.exp(rng.uniform(-700, 700, 2_000_000)))
    check_d(rng.uniform(0.9, 1.1, 2_000_000))
    check_d(np.array([1.0, 1.0 - 2.0 ** -4, np.nextafter(1.0 - 2.0 ** -4, 0), 1.0 + float.fromhex('0x1.09p-4'),
                      np.nextafter(1.0 + float.fromhex('0x1.09p-4'), 0), 2.0 ** -24, 2.0 ** -1022, 1.7e308, 0.5, 2.0]))


def test_powf_equals_host_libm(shim):
    """powf_host_exact vs the running libm's powf: the NDS+ arguments (x = sigma_t/sigma_maj in [0, 1],
    y = 1/(1+Tr) in [0.5, 1]), wide random ranges, and every special-case class of e_powf.c."""
    rng = np.random.default_rng(5)
    fp = C.POINTER(C.c_float)

    def check2(x, y):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        a, b = np.empty_like(x), np.empty_like(x)
        shim.model_powf(x.shape[0], x.ctypes.data_as(fp), y.ctypes.data_as(fp), a.ctypes.data_as(fp))
        shim.libm_powf(x.shape[0], x.ctypes.data_as(fp), y.ctypes.data_as(fp), b.ctypes.data_as(fp))
        both_nan = np.isnan(a) & np.isnan(b)
        bad = np.nonzero((a.view(np.uint32) != b.view(np.uint32)) & ~both_nan)[0]
        assert bad.size == 0, "powf: %d mismatches, e.g. x=%r y=%r model=%r libm=%r" % (
            bad.size, x[bad[:3]], y[bad[:3]], a[bad[:3]], b[bad[:3]])

    n = 4_000_000
    check2(rng.uniform(0, 1, n), np.float32(1) / (np.float32(1) + rng.uniform(0, 1, n).astype(np.float32)))
    check2(rng.uniform(0, 1.0001, n), rng.uniform(0.4, 1.1, n))
    check2(np.exp(rng.uniform(-88, 88, n)), rng.uniform(-3, 3, n))
    check2(np.exp(rng.uniform(-104, -80, n)), rng.uniform(-1.5, 1.5, n))        # subnormal x, results near under/overflow
    check2(np.exp(rng.uniform(-5, 5, n)), rng.uniform(-40, 40, n))              # |y log2 x| around 126..150
    check2(-np.exp(rng.uniform(-5, 5, n)), rng.integers(-40, 40, n).astype(np.float32))   # negative x, integer y
    check2(-np.exp(rng.uniform(-5, 5, 1000)), rng.uniform(-4, 4, 1000))         # negative x, non-integer y -> NaN
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, 1e-45, -1e-45,
                   3.4e38, -3.4e38, 2.0 ** -126, 2.0 ** 24, 2.0 ** 24 + 2, float.fromhex('0x1.fffffep-1')], dtype=np.float32)
    X, Y = np.meshgrid(sp, sp)
    check2(X.ravel(), Y.ravel())
    # every float x in one binade against a few exponents (table-index boundaries of log2_inline)
    bits = np.arange(0x3f000000, 0x3f800000, 7, dtype=np.uint32)
    for yv in (0.5, 0.75, 1.0, float.fromhex('0x1.fffffep-1')):
        check2(bits.view(np.float32), np.full(bits.shape, yv, dtype=np.float32))


@pytest.mark.parametrize("fn", ["sinf", "cosf"])
def test_sincos_equal_host_libm(shim, fn):
    rng = np.random.default_rng(1)
    two_pi = np.float32(2) * np.float32(np.pi)
    # phi = 2*Pi*u (HG / sphere sampling), theta in [-pi/4, 3pi/4] (concentric disk)
    check(shim, fn, two_pi * rng.random(6_000_000, dtype=np.float32))
    check(shim, fn, rng.uniform(-np.pi / 4, 3 * np.pi / 4, 4_000_000).astype(np.float32))
    check(shim, fn, rng.uniform(-119.9, 119.9, 2_000_000).astype(np.float32))
    check(shim, fn, (10.0 ** rng.uniform(-8, 0, 1_000_000)).astype(np.float32))
    edge = np.array([0.0, -0.0, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi, 2.0 ** -12, float.fromhex('0x1.fffffep-13'), 119.99], dtype=np.float32)
    check(shim, fn, np.concatenate([edge, -edge, np.nextafter(edge, np.float32(10)), np.nextafter(edge, np.float32(-10))]))


@pytest.mark.parametrize("fn", ["sinf", "cosf"])
def test_sincos_exhaustive(shim, fn):
    """every float in [2^-30, 8] and its negative (the path's angles lie in [-pi, 2 pi]); below 2^-30
    sampled.  ~6e8 arguments per function, a few seconds each in C."""
    lo, hi = np.float32(2.0 ** -30).view(np.uint32), np.float32(8.0).view(np.uint32)
    step = 1 << 24
    for start in range(int(lo), int(hi) + 1, step):
        bits = np.arange(start, min(start + step, int(hi) + 1), dtype=np.uint32)
        x = bits.view(np.float32)
        check(shim, fn, x)
        check(shim, fn, -x)
    tiny = (2.0 ** np.random.default_rng(3).uniform(-126, -30, 2_000_000)).astype(np.float32)
    check(shim, fn, tiny)
    check(shim, fn, -tiny)


def test_atanhf_log1pf_equal_host_libm(shim):
    """std::atanh(float) of SampleVisibleWavelengths (util/sampling.h:169-171) -- the wavelengths a temperature grid's blackbody
    emission is evaluated at -- and the log1pf under it.  The arguments the path produces are 0.85691062 - 1.82750197 * up with
    up in [0, 1]: that whole interval is covered float by float (~2.5e7 arguments), the rest of (-1, 1) sampled."""
    lo, hi = np.float32(0.85691062) - np.float32(1.82750197), np.float32(0.85691062)
    for a, b in ((np.float32(2.0 ** -30), hi), (np.float32(2.0 ** -30), -lo)):
        b0, b1 = int(np.float32(a).view(np.uint32)), int(np.float32(b).view(np.uint32))
        step = 1 << 23
        for start in range(b0, b1 + 1, step):
            x = np.arange(start, min(start + step, b1 + 1), dtype=np.uint32).view(np.float32)
            check(shim, "atanhf", x if a > 0 and b == hi else -x)
    rng = np.random.default_rng(5)
    u = rng.random(4_000_000, dtype=np.float32)
    check(shim, "atanhf", np.float32(0.85691062) - np.float32(1.82750197) * u)
    check(shim, "atanhf", rng.uniform(-1, 1, 2_000_000).astype(np.float32))
    check(shim, "atanhf", np.array([0.0, -0.0, 0.5, -0.5, 2.0 ** -28, 2.0 ** -29, float.fromhex('0x1.fffffep-1'), float.fromhex('-0x1.fffffep-1')], dtype=np.float32))
    check(shim, "log1pf", rng.uniform(-0.999, 40.0, 4_000_000).astype(np.float32))
    check(shim, "log1pf", (10.0 ** rng.uniform(-12, 6, 2_000_000)).astype(np.float32))
    check(shim, "log1pf", -(10.0 ** rng.uniform(-12, -0.001, 2_000_000)).astype(np.float32))
