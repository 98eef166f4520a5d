"""The SOURCE of the hub chains' device functions (graphvite_amd/csrc/gvk_chains.hip: train_long_chains / entry_sums,
train_short_chains), compiled for the host as written over a stand-in for one wave64 workgroup (tests/hostdev/simt.h,
tests/simt_build.py: one host thread per lane, DPP / shuffles / ballot as rendezvous of a wavefront's 64 threads, __syncthreads
as a barrier of 256) and run against the oracle's chains (`gvo_hot_unit_chains`: chains of up to 7 entries in sequence, a longer
chain as up to 256 / lanes tasks side by side, composed — in one round, or in rounds of GVK_HOT_ROUND_STEPS = 4 entries per task (form GVK_HOT_ROUNDS)).  What is left to the GPU is whether the hardware's lane maps are the
documented ones (tests/test_hub_chains_gpu.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import simt_build
from oracle_lib import Oracle

F = np.float32


@pytest.fixture(scope="module")
def simt():
    if not (os.path.exists(simt_build.CLANG) or os.path.exists(simt_build.OUT)):
        pytest.skip("no host clang++ to build the stand-in with")
    return C.CDLL(simt_build.build())


def oracle_chain(oracle, dim, vertex, context, lr, wd, nw, kv, kc, chain_start, entries, cap, max_tasks, long_task=0, round_steps=0):
    fn = oracle.lib.gvo_hot_unit_chains
    fp, up = np.ctypeslib.ndpointer(np.float32, flags="C"), np.ctypeslib.ndpointer(np.uint32, flags="C")
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, up, up, C.c_uint32, C.c_uint32, C.c_int]
    oracle.lib.gvo_set_long_task(int(long_task))
    oracle.lib.gvo_set_round_steps(int(round_steps))
    v, c = vertex.copy(), context.copy()
    assert fn(dim, v, c, lr, wd, nw, kv, kc, chain_start, entries, cap, max_tasks, 1) == 0
    return v, c


LANES_PER_CHAIN = {32: 8, 64: 16, 96: 8, 128: 16, 256: 16, 512: 32}  # default_lanes, gvk_tuning.h


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
@pytest.mark.parametrize("dim,round_steps", [(128, 4), (128, 0), (128, 2), (128, 5), (128, 8), (32, 4), (32, 0), (64, 4), (64, 0), (96, 4), (96, 0), (256, 4), (256, 0), (512, 4), (512, 0)])
def test_chain_side_of_a_unit_from_the_device_source(simt, dim, round_steps):
    """train_long_chains and train_short_chains as train_hot_kernel runs them — record lists, workgroup loops, the tasks of a long
    chain over the lane groups, composition in LDS — against gvo_hot_unit_chains on the same lists."""
    fp, up = np.ctypeslib.ndpointer(np.float32, flags="C"), np.ctypeslib.ndpointer(np.uint32, flags="C")
    simt.simt_unit_chains.restype = C.c_int
    simt.simt_unit_chains.argtypes = [C.c_int, fp, fp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, up, up, up, up, C.c_uint32,
                                      C.c_uint32, C.c_uint32, fp, fp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
    rng = np.random.default_rng(31 * dim)
    oracle = Oracle()
    kv, kc, rows, samples, k, cap = 30, 44, 300, 2600, 1, 7
    # rows of the same norm at every dim: with |c|^2 = 46 (0.3 per coordinate at dim 512) a step moves a logit by more than itself and
    # rounding differences between two correct implementations grow from round to round
    scale = 0.3 * min(1.0, (128.0 / dim) ** 0.5)
    vertex = (rng.standard_normal((rows, dim)) * scale).astype(F)
    context = (rng.standard_normal((rows, dim)) * scale).astype(F)
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
    assert lengths.max() > 2 * G * (256 // G) and len(short_chains) >= 64 // G and len(long_chains) >= 5  # tasks beyond two windows of entries, too
    mirror = np.ascontiguousarray(np.concatenate([vertex[:kv], context[:kc]]))
    to = mirror.copy()
    entries = np.ascontiguousarray(np.concatenate([entries, np.zeros(64, np.uint32)]))
    rc = simt.simt_unit_chains(dim, vertex, context, kv, kc, wd, nw, np.ascontiguousarray(start, np.uint32), entries, long_list,
                               short_list, chains, cap, round_steps, mirror, to, lr, F(np.log2(1.0 - float(lr) * float(wd))),
                               F(np.log2(1.0 - float(lr) * float(nw) * float(wd))), len(long_chains), -(-len(short_chains) // (256 // G)))
    assert rc == 0
    ov, oc = oracle_chain(oracle, dim, vertex, context, lr, wd, nw, kv, kc, np.ascontiguousarray(start, np.uint32), entries,
                          cap, 256 // G, round_steps=round_steps)
    want = np.concatenate([ov[:kv], oc[:kc]])
    for chain in range(chains):
        if lengths[chain] == 0 or chain in left_out:
            assert (to[chain] == mirror[chain]).all()  # rows without entries are copy_idle_rows' business
        else:
            # fp32 sums of up to a thousand entries: the tolerance follows the row's scale (the oracle adds in double)
            np.testing.assert_allclose(to[chain], want[chain], rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(want[chain]).max())),
                                       err_msg="chain %d of %d entries" % (chain, lengths[chain]))
