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


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dim,n", [(128, 17), (128, 257), (128, 600), (128, 1500), (32, 100), (96, 40), (64, 300)])
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


LANES_PER_CHAIN = {32: 8, 64: 16, 96: 8, 128: 16, 256: 16, 512: 32}  # default_lanes, gvk_kernels.hip


def unit_lists(rng, rows, kv, kc, samples, k):
    """A unit of `samples` samples with skewed hub rows on both sides, its work lists by the oracle (gvo_hot_lists) and the two
    record lists in the layout hot_list_kernel writes (long: {chain, first, n, -} from word 4; short: 16 words {chain, n, -, -,
    entries} from word 16; word 0 of each = the number of records)."""
    def column(hot):
        ids = rng.integers(hot, rows, samples)
        pick = rng.random(samples) < 0.5
        ids[pick] = np.minimum((rng.pareto(0.9, pick.sum()) * 1.5).astype(np.int64), hot - 1)
        return ids
    batch = np.stack([column(kc), column(kv)], 1).astype(np.uint32)  # records are {tail, head}
    negatives = column(kc).astype(np.uint32).reshape(samples, k)
    start, entries = Oracle().hot_lists(batch, negatives, kv, kc)
    return batch, negatives, start, entries


def records(start, entries, chains, cap):
    lengths = np.diff(start.astype(np.int64))
    long_chains = [c for c in range(chains) if lengths[c] > cap]
    short_chains = [c for c in range(chains) if 0 < lengths[c] <= cap]
    long_list = np.zeros(4 * (1 + chains), np.uint32)
    long_list[0] = len(long_chains)
    for j, c in enumerate(long_chains):
        long_list[4 + 4 * j:8 + 4 * j] = (c, start[c], lengths[c], 0)
    short_list = np.zeros(16 * (1 + chains), np.uint32)
    short_list[0] = len(short_chains)
    for j, c in enumerate(short_chains):
        short_list[16 + 16 * j], short_list[17 + 16 * j] = c, lengths[c]
        short_list[20 + 16 * j:20 + 16 * j + lengths[c]] = entries[start[c]:start[c + 1]]
    return long_list, short_list, long_chains, short_chains, lengths


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dim,gram", [(128, 0), (128, 1), (32, 0), (32, 1), (64, 0), (64, 1), (96, 0), (96, 1), (256, 0), (512, 0)])
def test_chain_side_of_a_unit_from_the_device_source(simt, dim, gram):
    """train_long_chains (the shipped steps, GRAM = 0, and the Gram form, GRAM = 1) and train_short_chains as train_hot_kernel runs
    them — record lists, workgroup loops, composition in LDS — against gvo_hot_unit_chains on the same lists."""
    fp, up = np.ctypeslib.ndpointer(np.float32, flags="C"), np.ctypeslib.ndpointer(np.uint32, flags="C")
    simt.simt_unit_chains.restype = C.c_int
    simt.simt_unit_chains.argtypes = [C.c_int, C.c_int, fp, fp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, up, up, up, up, C.c_uint32,
                                      C.c_uint32, fp, fp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
    rng = np.random.default_rng(31 * dim + gram)
    oracle = Oracle()
    kv, kc, rows, samples, k, cap = 30, 44, 300, 700, 1, 7
    vertex = (rng.standard_normal((rows, dim)) * 0.3).astype(F)
    context = (rng.standard_normal((rows, dim)) * 0.3).astype(F)
    lr, wd, nw = F(0.025), F(0.005), F(5.0)
    batch, negatives, start, entries = unit_lists(rng, rows, kv, kc, samples, k)
    chains = kv + kc
    long_list, short_list, long_chains, short_chains, lengths = records(start, entries, chains, cap)
    G = LANES_PER_CHAIN[dim]
    # the stand-in cannot run a shuffle that only some lane groups of a wavefront take part in (train_short_chains reads its record
    # under `mine`): whole wavefronts of short chains only — the chains beyond are left out of the run and of the comparison
    whole = len(short_chains) // (64 // G) * (64 // G)
    left_out, short_chains = short_chains[whole:], short_chains[:whole]
    short_list[0] = whole
    assert lengths.max() > cap * (256 // G) and len(short_chains) >= 64 // G and len(long_chains) >= 5  # steps beyond seven per task, too
    mirror = np.ascontiguousarray(np.concatenate([vertex[:kv], context[:kc]]))
    to = mirror.copy()
    entries = np.ascontiguousarray(np.concatenate([entries, np.zeros(64, np.uint32)]))
    rc = simt.simt_unit_chains(dim, gram, vertex, context, kv, kc, wd, nw, np.ascontiguousarray(start, np.uint32), entries, long_list,
                               short_list, chains, cap, mirror, to, lr, F(np.log2(1.0 - float(lr) * float(wd))),
                               F(np.log2(1.0 - float(lr) * float(nw) * float(wd))), len(long_chains), -(-len(short_chains) // (256 // G)))
    assert rc == 0
    ov, oc = oracle_chain(oracle, dim, vertex, context, lr, wd, nw, kv, kc, np.ascontiguousarray(start, np.uint32), entries,
                          16 if gram else cap, 64 if gram else 256 // G)
    want = np.concatenate([ov[:kv], oc[:kc]])
    for chain in range(chains):
        if lengths[chain] == 0 or chain in left_out:
            assert (to[chain] == mirror[chain]).all()  # rows without entries are copy_idle_rows' business
        else:
            np.testing.assert_allclose(to[chain], want[chain], rtol=1e-4, atol=1e-5, err_msg="chain %d of %d entries" % (chain, lengths[chain]))
