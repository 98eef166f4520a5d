"""The CPU oracle pinned against (a) the committed golden vectors that the REFERENCE's own host-compiled
arithmetic produced (tests/golden/make_golden.py), (b) that reference build itself when it is present
(oracle/_ref), and (c) published known-answer vectors (Philox)."""
import os

import numpy as np
import pytest

from oracle_lib import ADAGRAD, ADAM, MOMENTUM, RMSPROP, SGD, Reference
from util import conflict_free_batch, init_tables, random_batch

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_arithmetic.npz"))
HP = {SGD: (0, 0, 0), MOMENTUM: (0.9, 0, 0), ADAGRAD: (0, 0, 1e-10), RMSPROP: (0.99, 0, 1e-8), ADAM: (0.9, 0.99, 1e-8)}
CASES = [(dim, SGD) for dim in (32, 64, 96, 128, 256, 512)] + \
        [(dim, opt) for dim in (32, 128) for opt in (MOMENTUM, ADAGRAD, RMSPROP, ADAM)]


@pytest.mark.parametrize("dim,opt", CASES)
def test_train_matches_reference_golden_bit_exact(oracle, dim, opt):
    key = "d%d_o%d" % (dim, opt)
    v, c = GOLDEN[key + "_v_in"].copy(), GOLDEN[key + "_c_in"].copy()
    moments = [GOLDEN[key + "_m%d_in" % i].copy() if key + "_m%d_in" % i in GOLDEN else None for i in range(4)]
    loss = oracle.train(v, c, GOLDEN[key + "_pairs"], GOLDEN[key + "_negs"], 0.025, 0.005, 5.0, opt, moments, HP[opt])
    # same operations in the same order in IEEE fp32 -> identical bits
    assert (v == GOLDEN[key + "_v_out"]).all()
    assert (c == GOLDEN[key + "_c_out"]).all()
    assert (loss == GOLDEN[key + "_loss"]).all()
    for i, m in enumerate(moments):
        if m is not None:
            assert (m == GOLDEN[key + "_m%d_out" % i]).all()
    assert (oracle.predict(v, c, GOLDEN[key + "_pairs"]) == GOLDEN[key + "_logits"]).all()


def test_sigmoid_and_schedule_match_reference_golden(oracle):
    ys = np.array([oracle.sigmoid(float(x)) for x in GOLDEN["sigmoid_x"]], np.float32)
    assert (ys == GOLDEN["sigmoid_y"]).all()
    ids = GOLDEN["lr_batch_id"]
    assert ([oracle.lr(0.025, True, int(i), 1000) for i in ids] == GOLDEN["lr_linear"]).all()
    assert ([oracle.lr(0.025, False, int(i), 1000) for i in ids] == GOLDEN["lr_constant"]).all()
    assert oracle.lr(0.025, True, 99999, 1000) == np.float32(0.025) * np.float32(1e-4)  # floor, optimizer.h:78


@pytest.mark.parametrize("opt", [SGD, MOMENTUM, ADAGRAD, RMSPROP, ADAM])
def test_oracle_equals_live_reference_build(oracle, reference, opt):
    """Fresh random inputs against oracle/_ref/libgvref.so (skipped where neither it nor /root/reference exists)."""
    rng = np.random.default_rng(opt)
    N, B, k, dim = 100, 300, 3, 96
    v, c = init_tables(rng, N, N, dim)
    v *= 50
    c *= 50
    pairs, negs = random_batch(rng, N, N, B, k)
    nm = 0 if opt == SGD else (2 if opt == ADAM else 1)
    m1 = [np.full((N, dim), 1e-3, np.float32) if i < 2 * nm else None for i in range(4)]
    m2 = [None if m is None else m.copy() for m in m1]
    v1, c1, v2, c2 = v.copy(), c.copy(), v.copy(), c.copy()
    l1 = oracle.train(v1, c1, pairs, negs, 0.01, 0.001, 3.0, opt, m1, HP[opt])
    l2 = reference.train(v2, c2, pairs, negs, 0.01, 0.001, 3.0, opt, m2, HP[opt])
    assert (v1 == v2).all() and (c1 == c2).all() and (l1 == l2).all()


def test_philox_known_answers(oracle):
    """Random123 kat_vectors for philox4x32-10."""
    assert [hex(x) for x in oracle.philox([0, 0, 0, 0], [0, 0])] == \
        ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in oracle.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                          [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_host_uniforms_are_uniform_and_reproducible(oracle):
    a = oracle.host_uniforms(5, 3, 0, 20001)
    assert (oracle.host_uniforms(5, 3, 7, 100) == a[7:107]).all()       # random access into the stream
    assert (oracle.host_uniforms(5, 4, 0, 100) != a[:100]).any()        # other stream
    assert a.min() >= 0 and a.max() < 1
    assert abs(a.mean() - 0.5) < 0.01 and abs(a.var() - 1 / 12) < 0.005


ALIAS_GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_alias.npz"))
ALIAS_CASES = sorted(k[:-2] for k in ALIAS_GOLDEN.files if k.endswith("_w"))


@pytest.mark.parametrize("case", ALIAS_CASES)
def test_alias_table_matches_reference_golden_bit_exact(oracle, case):
    """gvo_alias_build / gvo_alias_sample against what the reference's own AliasTable<float, uint32_t / size_t>
    (include/base/alias_table.cuh:84-152, compiled as written over an emulated CUDA runtime) produced."""
    w = ALIAS_GOLDEN[case + "_w"]
    for index_bytes in (4, 8):
        prob, alias = oracle.alias_build(w, index_bytes)
        assert np.array_equal(prob, ALIAS_GOLDEN[case + "_prob"], equal_nan=True)
        assert (alias == ALIAS_GOLDEN[case + "_alias"]).all()
    prob, alias = oracle.alias_build(w, 4)
    draws = [oracle.alias_sample(prob, alias, float(r1), float(r2)) for r1, r2 in ALIAS_GOLDEN[case + "_rand"]]
    assert draws == ALIAS_GOLDEN[case + "_draws"].tolist()


def test_alias_table_equals_live_reference_build(oracle, reference):
    if reference.alias_lib is None:
        pytest.skip("oracle/_ref/libgvref_alias.so is not built")
    rng = np.random.default_rng(5)
    for trial in range(60):
        n = int(rng.integers(1, 4000))
        w = [rng.pareto(1.2, n) + 1e-3, np.floor(rng.pareto(1.5, n) + 1) ** 0.75, rng.integers(1, 4, n),
             np.where(rng.random(n) < 0.7, rng.random(n), 0)][trial % 4].astype(np.float32)
        if not w.any():
            w[0] = 1
        for index_bytes in (4, 8):
            rp, ra = reference.alias_build(w, index_bytes)
            op, oa = oracle.alias_build(w, index_bytes)
            assert np.array_equal(rp, op, equal_nan=True) and (ra == oa).all()


def test_alias_table_known_answers(oracle):
    # uniform weights: every slot keeps itself with probability 1
    prob, alias = oracle.alias_build(np.ones(5, np.float32))
    assert (prob == 1).all() and (alias == np.arange(5)).all()
    # hand-worked Vose with FIFO queues: w = [1, 2, 3, 2] -> mean 2 -> p = [.5, 1, 1.5, 1]
    #   little = [0], large = [1, 2, 3]; pop 0 & 1: alias[0] = 1, p[1] = .5 -> little = [1], large = [2, 3]
    #   pop 1 & 2: alias[1] = 2, p[2] = 1.0 -> large = [3, 2]; leftovers alias themselves
    prob, alias = oracle.alias_build(np.array([1, 2, 3, 2], np.float32))
    assert prob.tolist() == [0.5, 0.5, 1.0, 1.0] and alias.tolist() == [1, 2, 2, 3]
    assert oracle.alias_sample(prob, alias, 0.1, 0.4) == 0 and oracle.alias_sample(prob, alias, 0.1, 0.6) == 1
    assert oracle.alias_sample(prob, alias, 0.99, 0.999) == 3


def test_alias_table_distribution(oracle):
    rng = np.random.default_rng(0)
    w = rng.pareto(1.2, 200).astype(np.float32) + 0.01
    prob, alias = oracle.alias_build(w)
    # exact marginal implied by the table
    p = prob.astype(np.float64).clip(max=1) / len(w)
    implied = p.copy()
    np.add.at(implied, alias, 1 / len(w) - p)
    np.testing.assert_allclose(implied, w / w.sum(), rtol=2e-5, atol=1e-8)
    # 8-byte index variant builds the same table
    prob8, alias8 = oracle.alias_build(w, 8)
    assert (prob8 == prob).all() and (alias8 == alias).all()


def test_partition_and_schedule_properties(oracle):
    rng = np.random.default_rng(1)
    w = np.floor(rng.pareto(1.5, 1001)).astype(np.float32)
    for P in (1, 2, 4, 8):
        part, local, sizes = oracle.partition(w, P)
        assert sizes.sum() == len(w) and sizes.max() - sizes.min() <= 1
        order = np.lexsort((np.arange(len(w)), -w))
        zig = np.minimum(np.arange(len(w)) % (2 * P), 2 * P - 1 - np.arange(len(w)) % (2 * P))
        assert (part[order] == zig).all()
        for p in range(P):  # local ids are 0..size-1 in sorted order
            assert (local[order][zig == p] == np.arange(sizes[p])).all()
        sums = np.array([w[part == p].sum() for p in range(P)])
        assert sums.max() - sums.min() <= max(w.max(), 1) * 2  # degree-balanced
    for P, W in ((1, 1), (2, 2), (4, 4), (8, 8), (4, 2), (8, 4), (16, 8)):
        sch = oracle.schedule(P, W)
        seen = set()
        for step in sch:
            heads, tails = step[:, 0], step[:, 1]
            assert len(set(heads)) == len(heads) and len(set(tails)) == len(tails)  # orthogonal blocks
            seen.update(map(tuple, step.tolist()))
        assert len(seen) == P * P and len(sch) * sch.shape[1] == P * P  # every block exactly once per episode
        if P == W and W > 1:
            assert (sch[1][:, 0] == (np.arange(W) + 1) % W).all() and (sch[1][:, 1] == np.arange(W)).all()


def test_conflict_free_batch_is_order_independent(oracle):
    """The premise of the GPU parity protocol: on a conflict-free batch any processing order gives the same bits."""
    rng = np.random.default_rng(2)
    v, c = init_tables(rng, 500, 500, 64)
    pairs, negs = conflict_free_batch(rng, 500, 500, 100, 2)
    v1, c1, v2, c2 = v.copy(), c.copy(), v.copy(), c.copy()
    l1 = oracle.train(v1, c1, pairs, negs, 0.025, 0.005, 5.0)
    perm = rng.permutation(100)
    l2 = oracle.train(v2, c2, pairs[perm], negs[perm], 0.025, 0.005, 5.0)
    assert (v1 == v2).all() and (c1 == c2).all() and (l1[perm] == l2).all()


# ---- the reference's own solver front end (tests/golden/reference_solver.npz, oracle/ref_solver_harness.cpp) ----------
SOLVER_GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_solver.npz"))
SOLVER_CONFIGS = sorted(k[4:-5] for k in SOLVER_GOLDEN.files if k.startswith("cfg_") and k.endswith("_args"))


def equal_up_to_ties(part_a, local_a, part_b, local_b, weights, P):
    """Two partitions deal the same weight to every (partition, position): they differ at most in which of several
    equally heavy vertices sits where (the reference leaves that to std::sort, solver.h:878-882)."""
    S = int(max(local_a.max(), local_b.max())) + 1
    a, b = np.full((P, S), -1.0), np.full((P, S), -1.0)
    a[part_a, local_a] = weights
    b[part_b, local_b] = weights
    return np.array_equal(a, b)


@pytest.mark.parametrize("name", SOLVER_CONFIGS)
def test_partition_and_schedule_match_the_reference_solver(oracle, name):
    G = SOLVER_GOLDEN
    weighted, undirected, W, spw, P_arg, B, episode = [int(x) for x in G["cfg_%s_args" % name]]
    P = int(G["cfg_%s_info" % name][3])
    vw = G["cfg_%s_vertex_weights" % name]
    part, local, _ = oracle.partition(vw, P)
    gpart, glocal = G["cfg_%s_part" % name], G["cfg_%s_local" % name]
    if weighted:  # distinct vertex weights: the partition is determined
        assert (part == gpart).all() and (local == glocal).all()
    else:
        assert equal_up_to_ties(part, local, gpart, glocal, vw, P)
    want = G["cfg_%s_schedule" % name]
    got = oracle.schedule(P, int(G["cfg_%s_info" % name][7]))
    assert got.shape == want.shape and (got == want).all()


@pytest.mark.parametrize("name", [n for n in SOLVER_CONFIGS if not n.startswith("auto")])
def test_edge_sampler_matches_the_reference_sampler_draw_for_draw(oracle, name):
    """SamplerMixin::sample (solver.h:1012-1055) run as written on the oracle's uniform streams fills exactly the pools
    gvo_sample_edges fills."""
    G = SOLVER_GOLDEN
    key = "cfg_%s_" % name
    P, episode, num_sampler = [int(G[key + "info"][i]) for i in (3, 4, 6)]
    B = int(G[key + "args"][5])
    n = episode * B
    prob, alias = oracle.alias_build(G[key + "edge_weights"], 8)
    assert (prob == G[key + "edge_prob"]).all() and (alias == G[key + "edge_alias"]).all()
    pools = [np.zeros((n, 2), np.uint32) for _ in range(P * P)]
    work = (n + num_sampler - 1) // num_sampler
    for t in range(num_sampler):
        rnd = oracle.host_uniforms(int(G["seed"]), t, 0, 40 * n + 1000)
        oracle.sample_edges(G[key + "uv"], prob, alias, G[key + "part"], G[key + "local"], P, pools, work * t,
                            min(work * (t + 1), n), 4000, rnd)
    want = G[key + "edge_pools"]
    for hp in range(P):
        for tp in range(P):
            assert (pools[hp * P + tp] == want[hp, tp]).all(), (hp, tp)


def test_solver_golden_equals_live_reference_solver(oracle):
    """Where oracle/_ref/libgvref_solver.so is built (this container): the reference's solver run now reproduces the
    committed fixture — partition, schedule, edge pools."""
    from oracle_lib import ReferenceSolver
    if not ReferenceSolver.available():
        pytest.skip("oracle/_ref/libgvref_solver.so is not built")
    G = SOLVER_GOLDEN
    weighted, undirected, W, spw, P, B, episode = [int(x) for x in G["cfg_w_p4_args"]]
    rs = ReferenceSolver(oracle, int(G["seed"]), G["edges"], G["weights"], undirected, W, spw, P, 1, B, episode)
    labels, part, local, vw = rs.partition()
    assert (part == G["cfg_w_p4_part"]).all() and (local == G["cfg_w_p4_local"]).all()
    assert (rs.schedule() == G["cfg_w_p4_schedule"]).all()
    assert (rs.sample("LINE", 1) == G["cfg_w_p4_edge_pools"]).all()


# ---- hub rows trained by chains (the oracle's restatement of gvk_hot_build / gvk_train_episode_hot) -------------------------

def _hub_batch(rng, N, B, k, kv, kc):
    """Heads / tails: 40 % hub rows (skewed), the rest distinct non-hub rows; negatives: 30 % hub rows, the rest distinct."""
    def column(hot, lo):
        ids = lo + rng.permutation(N // 4)[:B]
        pick = rng.random(B) < 0.4
        ids[pick] = np.minimum((rng.pareto(1.0, pick.sum()) * 2).astype(np.int64), hot - 1)
        return ids
    batch = np.stack([column(kc, N // 2), column(kv, N // 4)], 1).astype(np.uint32)
    negs = (3 * N // 4 + rng.permutation(N // 4)[:B * k]).reshape(B, k)
    pick = rng.random((B, k)) < 0.3
    negs[pick] = rng.integers(0, kc, pick.sum())
    return batch, negs.astype(np.uint32)


def test_hub_work_lists_hold_every_update_of_a_hub_row_once(oracle):
    rng = np.random.default_rng(3)
    N, B, k, kv, kc = 16384, 1500, 2, 20, 30
    batch, negs = _hub_batch(rng, N, B, k, kv, kc)
    start, entries = oracle.hot_lists(batch, negs, kv, kc)
    assert start[0] == 0 and (np.diff(start.astype(np.int64)) >= 0).all() and start[-1] == len(entries)
    for row in range(kv):  # a head row's chain: the targets of its samples, sample by sample, negatives first
        mine = np.nonzero(batch[:, 1] == row)[0]
        want = np.concatenate([np.concatenate([negs[s], [batch[s, 0] | 0x80000000]]) for s in mine]) if len(mine) else []
        assert (entries[start[row]:start[row + 1]] == np.asarray(want, np.uint32)).all()
    for row in range(kc):  # a context row's chain: the head of every sample the row is a target of, with the label
        got = np.sort(entries[start[kv + row]:start[kv + row + 1]])
        want = np.sort(np.concatenate([batch[batch[:, 0] == row, 1] | 0x80000000] +
                                      [batch[negs[:, j] == row, 1] for j in range(k)]).astype(np.uint32))
        assert (got == want).all()


def test_hub_chains_without_hub_rows_are_the_plain_batch(oracle):
    rng = np.random.default_rng(4)
    N, B, k, dim = 512, 300, 2, 32
    batch = rng.integers(0, N, (B, 2)).astype(np.uint32)
    negs = rng.integers(0, N, (B, k)).astype(np.uint32)
    v, c = init_tables(rng, N, N, dim)
    v1, c1, v2, c2 = v.copy(), c.copy(), v.copy(), c.copy()
    loss1 = oracle.train(v1, c1, batch, negs, 0.025, 0.005, 5.0)
    loss2 = oracle.train_hot(v2, c2, batch, negs, 0.025, 0.005, 5.0, 0, 0, np.zeros(1, np.uint32), np.zeros(0, np.uint32), 256)
    assert (v1 == v2).all() and (c1 == c2).all() and (loss1 == loss2).all()


def test_hub_chains_keep_every_update_of_a_hub_row(oracle):
    """A hub row that is the head of m samples moves as m sequential updates move it (the chain), the rows of the other
    samples as in the plain batch; a chain cut into parts receives the sum of the parts' deltas."""
    rng = np.random.default_rng(5)
    N, m, dim = 256, 40, 32
    v, c = init_tables(rng, N, N, dim)
    v *= 30
    c *= 30
    batch = np.stack([100 + np.arange(m), np.zeros(m, np.int64)], 1).astype(np.uint32)  # head row 0, distinct tails
    negs = (200 + np.arange(m)).astype(np.uint32).reshape(m, 1)
    start, entries = oracle.hot_lists(batch, negs, 1, 0)
    assert len(entries) == 2 * m
    want_v, want_c = v.copy(), c.copy()
    oracle.train(want_v, want_c, batch, negs, 0.025, 0.005, 5.0)           # sequential: what the reference's loop does
    got_v, got_c = v.copy(), c.copy()
    oracle.train_hot(got_v, got_c, batch, negs, 0.025, 0.005, 5.0, 1, 0, start, entries, 256)
    # the chain reads the context rows as the batch found them, the sequential loop updates them as it goes: the head row
    # agrees to first order in the learning rate, and it moved far beyond what one update moves it
    one_v, one_c = v.copy(), c.copy()
    oracle.train(one_v, one_c, batch[:1], negs[:1], 0.025, 0.005, 5.0)
    moved, single = np.linalg.norm(got_v[0] - v[0]), np.linalg.norm(one_v[0] - v[0])
    assert moved > 3 * single  # m random directions: about sqrt(m) single steps
    assert np.linalg.norm(got_v[0] - want_v[0]) < 0.05 * moved
    # the partner rows were trained against the hub row as the batch found it: the same step to first order
    assert np.linalg.norm(got_c[100:100 + m] - want_c[100:100 + m]) < 0.5 * np.linalg.norm(want_c[100:100 + m] - c[100:100 + m])
    parts_v, parts_c = v.copy(), c.copy()
    oracle.train_hot(parts_v, parts_c, batch, negs, 0.025, 0.005, 5.0, 1, 0, start, entries, 20)  # 4 parts of 10 samples
    assert np.linalg.norm(parts_v[0] - got_v[0]) < 0.1 * moved


def test_chain_families_side_by_side_update_a_hub_pair_from_its_old_values(oracle):
    """A sample between two hub rows belongs to two chains.  The reference updates both rows from their old values
    (model/graph.h:47-58): the chains of a unit — both families — start from the hub rows as the unit found them, so a
    positive sample between two hub rows is exactly the reference's step (one family after the other would give the second
    row a step that already contains the sample's own, DESIGN.md §3.1.2: the second ordering fact)."""
    rng = np.random.default_rng(11)
    N, dim = 8, 32
    v, c = init_tables(rng, N, N, dim)
    v *= 40
    c *= 40
    batch = np.array([[1, 0]], np.uint32)      # {tail 1, head 0}: both hub rows (kv = kc = 2); no negatives
    negs = np.zeros((1, 0), np.uint32)
    start, entries = oracle.hot_lists(batch, negs, 2, 2)
    assert start.tolist() == [0, 1, 1, 1, 2] and entries.tolist() == [1 | 0x80000000, 0 | 0x80000000]
    want_v, want_c = v.copy(), c.copy()
    oracle.train(want_v, want_c, batch, negs, 0.025, 0.005, 5.0)
    for lerp in (False, True):  # one sample: the chains' way has one point
        side_v, side_c = v.copy(), c.copy()
        oracle.train_hot(side_v, side_c, batch, negs, 0.025, 0.005, 5.0, 2, 2, start, entries, 256, lerp=lerp)
        assert (side_v == want_v).all() and (side_c == want_c).all()      # the reference's step, bit for bit


def test_long_chains_are_cut_into_at_most_max_tasks(oracle):
    """A chain of 64 entries with cap 4 is 16 tasks of 4; with max_tasks = 8 it is 8 tasks of 8 = cap 8 without a limit."""
    rng = np.random.default_rng(12)
    N, dim, m = 256, 32, 64
    v, c = init_tables(rng, N, N, dim)
    v *= 20
    c *= 20
    batch = np.stack([100 + np.arange(m), np.zeros(m, np.int64)], 1).astype(np.uint32)
    negs = np.zeros((m, 0), np.uint32)
    start, entries = oracle.hot_lists(batch, negs, 1, 0)
    out = {}
    for name, cap, max_tasks in (("cap 4", 4, 0), ("cap 4, 8 tasks", 4, 8), ("cap 8", 8, 0), ("cap 8, 16 tasks", 8, 16)):
        tv, tc = v.copy(), c.copy()
        oracle.train_hot(tv, tc, batch, negs, 0.025, 0.005, 5.0, 1, 0, start, entries, cap, max_tasks)
        out[name] = tv[0].copy()
    assert (out["cap 4, 8 tasks"] == out["cap 8"]).all() and (out["cap 8, 16 tasks"] == out["cap 8"]).all()
    assert not (out["cap 4"] == out["cap 8"]).all()
