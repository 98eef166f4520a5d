"""The SOURCE of `long_chain_gram` (graphvite_amd/csrc/gvk_kernels.hip; GVK_TUNE_HOT_GRAM, off by default), compiled for the host as
written over a stand-in for one wave64 workgroup (tests/hostdev/simt.h, tests/simt_build.py: one host thread per lane, the
matrix instruction / DPP / shuffles / ballot as rendezvous of a wavefront's 64 threads) and run against the oracle's tasks-of-16
form (`gvo_hot_unit_chains`, cap 16, 64 tasks).  tests/test_gram_chain_cpu.py checks a hand-written twin of the algorithm; this
one checks the device function's own text — what is left to the GPU is whether the hardware's lane maps are the documented ones."""
import ctypes as C
import os

import numpy as np
import pytest

import simt_build
from oracle_lib import Oracle
from test_gram_chain_cpu import oracle_chain

F = np.float32


@pytest.fixture(scope="module")
def simt():
    if not (os.path.exists(simt_build.CLANG) or os.path.exists(simt_build.OUT)):
        pytest.skip("no host clang++ to build the stand-in with")
    lib = C.CDLL(simt_build.build())
    fp, up = np.ctypeslib.ndpointer(np.float32, flags="C"), np.ctypeslib.ndpointer(np.uint32, flags="C")
    lib.simt_long_chain_gram.restype = C.c_int
    lib.simt_long_chain_gram.argtypes = [C.c_int, fp, fp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, up, fp, fp, C.c_float, C.c_float,
                                         C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]
    return lib


@pytest.mark.parametrize("dim,n", [(128, 8), (128, 17), (128, 250), (128, 257), (128, 1024), (128, 1500), (32, 100), (64, 257), (96, 40)])
def test_device_source_of_the_gram_form_matches_the_oracle(simt, dim, n):
    rng = np.random.default_rng(77 * dim + n)
    oracle = Oracle()
    kv, kc, rows = 6, 5, 400
    vertex = (rng.standard_normal((rows, dim)) * 0.35).astype(F)
    context = (rng.standard_normal((rows, dim)) * 0.35).astype(F)
    lr, wd, nw = F(0.025), F(0.005), F(5.0)
    for chain in (2, kv + 3):  # a head row's chain and a context row's
        partners = rng.integers(0, rows, n).astype(np.uint32)
        hub = rng.random(n) < 0.25  # a quarter of the entries name hub rows (read from the mirror)
        partners[hub] = rng.integers(0, kc if chain < kv else kv, hub.sum())
        labels = (rng.random(n) < 0.4).astype(np.uint32)
        junk = rng.integers(kv + kc, rows, 37).astype(np.uint32)  # the chain before this one owns the first 37 entries
        entries = np.concatenate([junk, partners | labels << 31]).astype(np.uint32)
        chain_start = np.zeros(kv + kc + 1, np.uint32)
        chain_start[chain:] = 37
        chain_start[chain + 1:] = 37 + n
        mirror = np.ascontiguousarray(np.concatenate([vertex[:kv], context[:kc]]))
        to = mirror.copy()
        rc = simt.simt_long_chain_gram(dim, vertex, context, kv, kc, wd, nw, entries, mirror, to, lr,
                                       F(np.log2(1.0 - float(lr) * float(wd))), F(np.log2(1.0 - float(lr) * float(nw) * float(wd))),
                                       chain, 37, n)
        assert rc == 0
        if n <= 1024:
            ov, oc = oracle_chain(oracle, dim, vertex, context, lr, wd, nw, kv, kc, chain_start, entries, 16, 64)
            want = ov[chain] if chain < kv else oc[chain - kv]
        else:  # segments of 1024 entries, one after the other: the second starts from the row the first left
            first = chain_start.copy()
            first[chain + 1:] = 37 + 1024
            ov, oc = oracle_chain(oracle, dim, vertex, context, lr, wd, nw, kv, kc, first, entries, 16, 64)
            second = chain_start.copy()
            second[chain:] = 37 + 1024
            second[chain + 1:] = 37 + n
            # hub partners are read from the mirror the unit started with: only the own row moves on
            v2, c2 = vertex.copy(), context.copy()
            (v2 if chain < kv else c2)[chain if chain < kv else chain - kv] = (ov if chain < kv else oc)[chain if chain < kv else chain - kv]
            ov, oc = oracle_chain(oracle, dim, v2, c2, lr, wd, nw, kv, kc, second, entries, 16, 64)
            want = ov[chain] if chain < kv else oc[chain - kv]
        np.testing.assert_allclose(to[chain], want, rtol=1e-4, atol=1e-5)
        others = np.ones(kv + kc, bool)
        others[chain] = False
        assert (to[others] == mirror[others]).all()  # a chain stores its own row and nothing else
