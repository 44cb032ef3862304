"""CPU test of the host logic that decides WHICH cells' singlet sums muxgl_fmx_prepare recomputes in the reference's
arithmetic (popscle_amd/csrc/score_exact.hpp, settle_with; cmd_cram_freemux2.cpp:183-189, sc_drop_seq.h:190-198): given
sums that deviate from the reference's by rounding noise, the cells it asks exact sums for must be enough to make the
sorted order the reference's -- ties by index included -- and no more than the cells within reach of a neighbour.

The header is compiled into a small shared object with hipcc (host code only; no device is touched)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    so = str(tmp_path_factory.mktemp("probe") / "score_order_probe.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-shared", "-fPIC",
                        "-I", os.path.join(ROOT, "popscle_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "csrc", "score_order_probe.hip"), "-o", so],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = C.CDLL(so)
    lib.probe_settle.restype = C.c_int
    return lib


def ref_order(l0, l2):
    """the reference's sort: score descending, ties by index descending"""
    s = l2 - l0
    return np.lexsort((-np.arange(s.size), -s))


def settle(lib, d0, d2, x0, x2):
    Cn = d0.size
    l0, l2 = d0.copy(), d2.copy()
    n_exact, calls = C.c_int64(), C.c_int32()
    was = np.zeros(Cn, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib.probe_settle(C.c_int64(Cn), p(l0), p(l2), p(x0), p(x2), C.byref(n_exact), C.byref(calls), p(was))
    assert rc == 0
    assert n_exact.value == int(was.sum())
    return l0, l2, was.astype(bool), calls.value


@pytest.mark.parametrize("seed", range(8))
def test_settled_order_is_the_exact_order(probe, seed):
    r = np.random.default_rng(seed)
    Cn = int(r.choice([2, 3, 50, 2000, 20000]))
    # exact sums: groups of cells with the same score (droplets: 0), scores a few ulps apart, and well separated ones
    x0 = -r.uniform(1, 3000, Cn)
    base = r.choice([0.0, 0.0, 5.0, -3.25]) + r.choice([0, 0, 1, 2, 3], Cn) * r.choice([0.0, 1e-13, 1e-6, 1.0])
    kind = r.random(Cn)
    score = np.where(kind < 0.5, base, r.normal(0, 50, Cn))
    score = np.where(kind < 0.1, score + r.integers(-3, 4, Cn) * 1e-14, score)   # rounding-level differences
    x2 = x0 + score
    # what the device returns: the same to ~1e-13 relative
    d0 = x0 * (1 + r.uniform(-1, 1, Cn) * 2e-13)
    d2 = x2 * (1 + r.uniform(-1, 1, Cn) * 2e-13)
    l0, l2, was, calls = settle(probe, d0, d2, x0, x2)
    assert np.array_equal(ref_order(l0, l2), ref_order(x0, x2)), "the settled sums do not sort as the reference's"
    # cells that were not asked for keep the device's sums; the asked ones hold the exact sums
    assert np.array_equal(l0[~was], d0[~was]) and np.array_equal(l2[~was], d2[~was])
    assert np.array_equal(l0[was], x0[was]) and np.array_equal(l2[was], x2[was])
    # every cell whose exact score is within reach (1e-9 x magnitude) of another cell's was asked for ...
    s = x2 - x0
    o = np.argsort(s)
    mag = np.maximum(1.0, np.maximum(np.abs(x0), np.abs(x2)))
    gap = np.diff(s[o])
    close = gap <= 0.5e-9 * np.maximum(mag[o][1:], mag[o][:-1])
    must = np.zeros(Cn, dtype=bool)
    must[o[1:][close]] = True
    must[o[:-1][close]] = True
    assert was[must].all()
    # ... and nobody whose score is far (1e-8 x magnitude) from every other cell's
    far = np.ones(Cn, dtype=bool)
    lim = 2e-9 * np.maximum(mag[o][1:], mag[o][:-1]) + 1e-9
    far[o[1:][gap <= lim]] = False
    far[o[:-1][gap <= lim]] = False
    assert not was[far].any()
    assert calls <= 3


def test_nothing_to_settle(probe):
    x0 = -np.arange(1.0, 101.0)
    x2 = x0 + np.arange(100) * 0.37
    l0, l2, was, calls = settle(probe, x0.copy(), x2.copy(), x0, x2)
    assert calls == 0 and not was.any()


def test_non_finite_scores_are_left_alone(probe):
    x0 = np.array([-1.0, -np.inf, -2.0])
    x2 = np.array([-1.0, -3.0, -2.0])
    l0, l2, was, calls = settle(probe, x0.copy(), x2.copy(), x0, x2)
    assert calls == 0 and not was.any()
