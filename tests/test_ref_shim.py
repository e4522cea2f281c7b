"""The host execution model under oracle/_ref (oracle/ref_shim/cuda_on_host.h) checked BY ITSELF, on kernels written for the purpose
(tests/ref_shim_check/shim_kernels.cu): if the fibers, barriers, shared memory, atomics, overloads or CUB stand-ins were wrong, the
bit-for-bit agreement of tests/test_ref_parity.py would mean nothing.  Needs only g++ (no reference sources)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(HERE, "ref_shim_check", "shim_kernels.cu")
    so = os.path.join(HERE, "ref_shim_check", "libshimcheck.so")
    text = build_ref.launch_syntax_to_cxx(open(src).read())          # the one substitution the reference's sources get, too
    subprocess.run(["g++", "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-fopenmp", "-w", "-I", build_ref.SHIM, "-shared", "-o", so, "-"],
                   input=text.encode(), check=True)
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("threads", [1, 4])
def test_blocks_as_fibers_shared_memory_barriers_and_votes(lib, threads):
    lib.shim_set_threads(threads)
    gy, n = 5, 5 * 3 * 256 - 100                       # the last block is ragged
    x = np.random.default_rng(0).normal(size=n).astype(np.float32)
    out = np.zeros(3 * gy, np.float32)
    votes = np.zeros(3 * gy, np.int32)
    lib.shim_block_sum(_p(x), _p(out), _p(votes), n, gy)
    pad = np.concatenate([x, np.zeros(3 * gy * 256 - n, np.float32)]).reshape(3 * gy, 256)
    # the kernel's own summation order: 16 column sums of 16 strided values, then their sum -- all in fp32
    col = np.zeros((3 * gy, 16), np.float32)
    for k in range(16):
        col += pad[:, 16 * k:16 * k + 16]
    want = np.zeros(3 * gy, np.float32)
    for k in range(16):
        want += col[:, k]
    assert np.array_equal(out, want)
    assert np.array_equal(votes, (pad > 0).sum(1).astype(np.int32))
    lib.shim_set_threads(1)


@pytest.mark.parametrize("threads", [1, 4])
def test_flat_launch_and_float_atomics(lib, threads):
    lib.shim_set_threads(threads)
    n = 10007
    x = np.random.default_rng(1).integers(-8, 9, n).astype(np.float32)     # exactly representable sums: order cannot matter
    bins = np.zeros(7, np.float32)
    lib.shim_scatter_add(_p(x), _p(bins), n)
    want = np.array([x[k::7].sum() for k in range(7)], np.float32)
    assert np.array_equal(bins, want)
    lib.shim_set_threads(1)


def test_cuda_min_max_overloads(lib):
    u = np.zeros(2, np.uint32)
    d = np.zeros(2, np.float64)
    lib.shim_minmax(_p(u), _p(d))
    assert list(u) == [0, 10]
    assert d[0] == 1e-6 and d[1] == float(np.float32(1e-6))


def test_cub_contracts(lib):
    rng = np.random.default_rng(2)
    n = 5000
    keys = (rng.integers(0, 40, n).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 6, n).astype(np.uint64) | (rng.integers(0, 3, n).astype(np.uint64) << np.uint64(45))
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = np.zeros_like(keys), np.zeros_like(vals)
    end_bit = 32 + 6                                     # bits above end_bit must be ignored, ties keep their input order (stable)
    lib.shim_sort_pairs(_p(keys), _p(ko), _p(vals), _p(vo), n, end_bit)
    masked = keys & np.uint64((1 << end_bit) - 1)
    order = np.argsort(masked, kind="stable")
    assert np.array_equal(vo, vals[order]) and np.array_equal(ko, keys[order])
    x = rng.integers(0, 9, n).astype(np.uint32)
    y = np.zeros_like(x)
    lib.shim_inclusive_sum(_p(x), _p(y), n)
    assert np.array_equal(y, np.cumsum(x, dtype=np.uint64).astype(np.uint32))
