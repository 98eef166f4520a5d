"""-m gpu: the HIP kernels (through the C ABI, graphvite_amd/libgvk.so) against the CPU oracle."""
import numpy as np
import pytest
import torch

from graphvite_amd import kernels as K
from oracle_lib import ADAGRAD, ADAM, MOMENTUM, RMSPROP, SGD
from util import conflict_free_batch, init_tables, power_law_weights, random_batch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
# fp32 tolerance of the parity protocol: only the dot-product summation order and
# FMA contraction differ between the wave64 kernel and the sequential oracle.
RTOL, ATOL = 1e-5, 1e-7

OPTS = {
    "SGD": (SGD, K.OptimizerSpec("SGD", 0.025, 0.005), (0, 0, 0)),
    "Momentum": (MOMENTUM, K.OptimizerSpec("Momentum", 0.025, 0.005, hp0=0.9), (0.9, 0, 0)),
    "AdaGrad": (ADAGRAD, K.OptimizerSpec("AdaGrad", 0.025, 0.005, epsilon=1e-10), (0, 0, 1e-10)),
    "RMSprop": (RMSPROP, K.OptimizerSpec("RMSprop", 0.025, 0.005, hp0=0.99, epsilon=1e-8), (0.99, 0, 1e-8)),
    "Adam": (ADAM, K.OptimizerSpec("Adam", 0.025, 0.005, hp0=0.9, hp1=0.99, epsilon=1e-8), (0.9, 0.99, 1e-8)),
}


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.view(dtype)
    return t.to(DEV)


def run_hip(hip, v, c, pairs, negs, spec, neg_weight, moments=None, **kw):
    tv, tc = dev(v), dev(c)
    tm = None if moments is None else [None if m is None else dev(m) for m in moments]
    tp = dev(pairs.view(np.int32))
    tn = None if negs is None else dev(negs.view(np.int32))
    loss = torch.zeros(pairs.shape[0], dtype=torch.float32, device=DEV)
    k = kw.pop("k", 0 if negs is None else negs.shape[1])
    hip.train(tv, tc, tp, loss, spec, k, neg_weight, negatives=tn, moments=tm, **kw)
    torch.cuda.synchronize()
    out_m = None if tm is None else [None if m is None else m.cpu().numpy() for m in tm]
    return tv.cpu().numpy(), tc.cpu().numpy(), loss.cpu().numpy(), out_m


@pytest.mark.parametrize("dim", [32, 64, 96, 128, 256, 512])
@pytest.mark.parametrize("k", [0, 1, 3])
def test_sgd_conflict_free_matches_oracle(hip, oracle, dim, k):
    rng = np.random.default_rng(dim * 10 + k)
    N, B = 4096, 777
    v, c = init_tables(rng, N, N, dim)
    v *= 20  # logits away from 0 so that sigmoid / log are exercised off the linear regime
    c *= 20
    pairs, negs = conflict_free_batch(rng, N, N, B, k)
    ov, oc = v.copy(), c.copy()
    oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
    hv, hc, hloss, _ = run_hip(hip, v, c, pairs, negs if k else None, OPTS["SGD"][1], 5.0, k=k)
    np.testing.assert_allclose(hv, ov, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hc, oc, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hloss, oloss, rtol=RTOL, atol=1e-6)
    # rows no pair touched are bit-identical
    touched_v = np.zeros(N, bool)
    touched_v[pairs[:, 1]] = True
    assert (hv[~touched_v] == v[~touched_v]).all()


@pytest.mark.parametrize("name", ["Momentum", "AdaGrad", "RMSprop", "Adam"])
@pytest.mark.parametrize("dim", [96, 128])
def test_moment_optimizers_match_oracle(hip, oracle, name, dim):
    opt_id, spec, hp = OPTS[name]
    rng = np.random.default_rng(7)
    N, B, k = 2048, 500, 2
    v, c = init_tables(rng, N, N, dim)
    v *= 20
    c *= 20
    nm = spec.num_moment
    moments = [rng.uniform(0, 1e-3, (N, dim)).astype(np.float32) if i < 2 * nm else None for i in range(4)]
    pairs, negs = conflict_free_batch(rng, N, N, B, k)
    ov, oc, om = v.copy(), c.copy(), [None if m is None else m.copy() for m in moments]
    oloss = oracle.train(ov, oc, pairs, negs, spec.lr, spec.weight_decay, 5.0, opt_id, om, hp)
    hv, hc, hloss, hm = run_hip(hip, v, c, pairs, negs, spec, 5.0, moments=moments)
    # sqrt / divide chains amplify the last-bit differences of the dot product a little more than SGD
    np.testing.assert_allclose(hv, ov, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(hc, oc, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(hloss, oloss, rtol=1e-5, atol=1e-6)
    for a, b in zip(hm, om):
        if a is not None:
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-8)  # moments pass through zero


@pytest.mark.parametrize("lanes", [8, 16, 32, 64])
def test_lane_group_variants_agree(hip, oracle, lanes):
    if not hip.has_ab_builds:
        pytest.skip("non-default lane groups exist in the A/B library only (GVK_LIBRARY=.../libgvk_ab.so)")
    rng = np.random.default_rng(lanes)
    N, B, k, dim = 4096, 1000, 1, 128
    v, c = init_tables(rng, N, N, dim)
    v *= 20
    c *= 20
    pairs, negs = conflict_free_batch(rng, N, N, B, k)
    ov, oc = v.copy(), c.copy()
    oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
    hip.set_lanes_per_pair(lanes)
    try:
        hv, hc, hloss, _ = run_hip(hip, v, c, pairs, negs, OPTS["SGD"][1], 5.0)
    finally:
        hip.set_lanes_per_pair(0)
    np.testing.assert_allclose(hv, ov, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hc, oc, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hloss, oloss, rtol=RTOL, atol=1e-6)


def test_reference_shape_variant_is_also_correct(hip, oracle):
    """The A/B baseline kernel (the reference's launch shape on wave64) computes the same thing."""
    if not hip.has_ab_builds:
        pytest.skip("the reference-shape kernel exists in the A/B library only (GVK_LIBRARY=.../libgvk_ab.so)")
    rng = np.random.default_rng(41)
    N, B, k, dim = 4096, 1500, 1, 128
    v, c = init_tables(rng, N, N, dim)
    v *= 20
    c *= 20
    pairs, negs = conflict_free_batch(rng, N, N, B, k)
    ov, oc = v.copy(), c.copy()
    oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
    hip.set_variant(3)
    try:
        hv, hc, hloss, _ = run_hip(hip, v, c, pairs, negs, OPTS["SGD"][1], 5.0)
    finally:
        hip.set_variant(0)
    np.testing.assert_allclose(hv, ov, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hc, oc, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hloss, oloss, rtol=RTOL, atol=1e-6)


def test_pair_sees_its_own_negative_update(hip, oracle):
    """negative == positive tail (and repeated negatives): the pair must read its own updated row, as the
    reference's sequential warp does (include/instance/gpu/graph.cuh:62-88)."""
    rng = np.random.default_rng(3)
    N, B, k, dim = 1024, 100, 3, 128
    v, c = init_tables(rng, N, N, dim)
    v *= 30
    c *= 30
    pairs, negs = conflict_free_batch(rng, N, N, B, k)
    negs[:, 2] = pairs[:, 0]      # last negative is the tail itself
    negs[::2, 1] = negs[::2, 0]   # adjacent duplicate negatives
    ov, oc = v.copy(), c.copy()
    oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
    hv, hc, hloss, _ = run_hip(hip, v, c, pairs, negs, OPTS["SGD"][1], 5.0)
    np.testing.assert_allclose(hv, ov, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hc, oc, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hloss, oloss, rtol=RTOL, atol=1e-6)


def _run_batch(rng, N, B, k, run_lengths):
    """A batch whose heads come in adjacent runs of the given lengths (cycled); tails and negatives all distinct."""
    pairs, negs = conflict_free_batch(rng, N, N, B, k)
    heads, i, r = pairs[:, 1].copy(), 0, 0
    while i < B:
        n = min(run_lengths[r % len(run_lengths)], B - i)
        heads[i:i + n] = heads[i]
        i, r = i + n, r + 1
    pairs[:, 1] = heads
    return pairs, negs


def _whole_runs(pairs, segment):
    """(first index, last index, inside one `segment`-aligned window) of every run of adjacent same-head pairs."""
    B = len(pairs)
    first = np.flatnonzero(np.r_[True, pairs[1:, 1] != pairs[:-1, 1]])
    last = np.r_[first[1:], B] - 1
    return first, last, first // segment == last // segment


def _check_runs(pairs, negs, v, c, ov, oc, oloss, hv, hc, hloss, segment):
    """Runs that lie inside one segment must equal the sequential oracle; a run cut by a segment boundary is trained
    by two lane groups / wavefronts concurrently and is only checked through its context rows' neighbours."""
    first, last, whole = _whole_runs(pairs, segment)
    rows = pairs[first[whole], 1]
    np.testing.assert_allclose(hv[rows], ov[rows], rtol=RTOL, atol=ATOL)
    in_whole = np.repeat(whole, last - first + 1)
    np.testing.assert_allclose(hloss[in_whole], oloss[in_whole], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(hc[pairs[in_whole, 0]], oc[pairs[in_whole, 0]], rtol=RTOL, atol=ATOL)
    if negs is not None:
        np.testing.assert_allclose(hc[negs[in_whole].ravel()], oc[negs[in_whole].ravel()], rtol=RTOL, atol=ATOL)
    assert whole.sum() > len(whole) // 3


@pytest.mark.parametrize("dim", [32, 64, 96, 128, 256, 512])
@pytest.mark.parametrize("steps", [0, 1, 2, 4])
@pytest.mark.parametrize("explicit", [True, False])
def test_segment_kernel_trains_runs_in_sequence(hip, oracle, dim, steps, explicit):
    """train_segment_kernel (GVK_TUNE_SEGMENT_STEPS; the A/B alternative to train_runs_kernel for SGD with one negative):
    adjacent pairs that share a head row and lie in one wavefront's segment are one run, trained one after the other on one register copy of the row (the reference's warp does that with
    consecutive iterations of its loop, gpu/graph.cuh:54-94).  With distinct context rows the result equals the
    SEQUENTIAL oracle — no update of the head row is lost — and every sample keeps its own negative and loss slot."""
    if steps and not hip.has_ab_builds:
        pytest.skip("train_segment_kernel exists in the A/B library only (GVK_LIBRARY=.../libgvk_ab.so)")
    rng = np.random.default_rng(dim + steps)
    N, B, k = 8192, 1531, 1
    v, c = init_tables(rng, N, N, dim)
    v *= 20
    c *= 20
    pairs, negs = _run_batch(rng, N, B, k, [1, 2, 1, 5, 3, 1, 1, 9, 16, 2, 1, 1, 4, 33])
    hip.set_segment_steps(steps)
    try:
        name = hip.describe_train(dim, "SGD", k, explicit, B, N)
        if steps == 0:  # the default: runs of same-head samples on tables below 16 MiB, the per-pair kernel above
            assert ("train_runs_kernel" in name) == (N * dim * 4 < 16 << 20), name
            assert "train_kernel<" in hip.describe_train(dim, "SGD", k, explicit, B, 1 << 20)
        if "train_segment_kernel" not in name:
            pytest.skip("no %d-step build at dim %d: %s" % (steps, dim, name))
        segment = int(name.split("> ")[1].split()[0])
        if explicit:
            hv, hc, hloss, _ = run_hip(hip, v, c, pairs, negs, OPTS["SGD"][1], 5.0)
        else:  # in-kernel draw: feed the oracle the negatives of the RNG contract; drop pairs whose draw collides
            w = rng.uniform(0.5, 1.5, N).astype(np.float32)  # a flat table: few draws collide in one batch
            prob, alias, packed = K.alias_build(w)
            negs = oracle.negatives(prob, alias, 9, 4, B, k)
            tv, tc, loss = dev(v), dev(c), torch.zeros(B, device=DEV)
            hip.train(tv, tc, dev(pairs.view(np.int32)), loss, OPTS["SGD"][1], k, 5.0,
                      table=K.packed_to_device(packed, DEV), seed=9, batch_id=4)
            torch.cuda.synchronize()
            hv, hc, hloss = tv.cpu().numpy(), tc.cpu().numpy(), loss.cpu().numpy()
    finally:
        hip.set_segment_steps(0)
    if explicit:
        ov, oc = v.copy(), c.copy()
        oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
        _check_runs(pairs, negs, v, c, ov, oc, oloss, hv, hc, hloss, segment)
    else:
        # drawn negatives repeat (hubs) and hit tails: compare the samples whose context rows nobody else touches
        ov, oc = v.copy(), c.copy()
        oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
        ctx = np.concatenate([pairs[:, :1], negs], 1)
        ids, count = np.unique(ctx, return_counts=True)
        shared = set(ids[count > 1].tolist())
        first, last, whole = _whole_runs(pairs, segment)
        clean_run = np.array([w and not (set(ctx[f:l + 1].ravel().tolist()) & shared) for f, l, w in zip(first, last, whole)])
        assert clean_run.sum() > len(first) // 8
        rows = pairs[first[clean_run], 1]
        np.testing.assert_allclose(hv[rows], ov[rows], rtol=RTOL, atol=ATOL)
        in_clean = np.repeat(clean_run, last - first + 1)
        np.testing.assert_allclose(hloss[in_clean], oloss[in_clean], rtol=RTOL, atol=1e-6)


@pytest.mark.parametrize("dim,k", [(128, 1), (128, 3), (96, 1), (32, 2), (512, 1), (128, 0)])
def test_runs_kernel_trains_runs_in_sequence(hip, oracle, dim, k):
    """train_runs_kernel (GVK_TUNE_VARIANT 4, the A/B build in which one lane group trains a whole run): same property,
    any num_negative, runs cut at multiples of the run cap."""
    rng = np.random.default_rng(dim + k)
    N, B = 8192, 1500
    v, c = init_tables(rng, N, N, dim)
    v *= 20
    c *= 20
    pairs, negs = _run_batch(rng, N, B, k, [1, 2, 1, 5, 3, 1, 1, 9, 16, 2])
    ov, oc = v.copy(), c.copy()
    oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
    hip.set_variant(4)
    hip.set_run_cap(16)
    try:
        assert "train_runs_kernel" in hip.describe_train(dim, "SGD", k, True, B)
        hv, hc, hloss, _ = run_hip(hip, v, c, pairs, negs if k else None, OPTS["SGD"][1], 5.0, k=k)
    finally:
        hip.set_run_cap(0)
        hip.set_variant(0)
    _check_runs(pairs, negs if k else None, v, c, ov, oc, oloss, hv, hc, hloss, 16)
    first, last, whole = _whole_runs(pairs, 16)
    assert (~whole).sum() > 0


@pytest.mark.parametrize("variant", [0, 2])
def test_small_head_table_trains_in_several_launches(hip, oracle, variant):
    """A head table with fewer than batch / 2 rows: the batch is trained as consecutive launches of at most 2 samples per
    row (4 here; GVK_TUNE_SPLIT_HITS).  Launch boundaries add ordering and nothing else: with every head row's
    samples adjacent (runs of 16 = the run cap) and distinct context rows the whole batch equals the SEQUENTIAL oracle,
    and every sample keeps its own negative and loss slot.  variant 2: the per-pair kernel with every head row once per
    launch — a row's four samples are trained by four launches, each starting from what the one before wrote."""
    rng = np.random.default_rng(7 + variant)
    dim, k = 128, 1
    n_vertex, n_context = 96, 8192
    B = 1536 if variant == 0 else 384       # 16 / 4 samples per head row -> 4 launches / 1 launch of 4 per row
    v, c = init_tables(rng, n_vertex, n_context, dim)
    v *= 20
    c *= 20
    ctx = rng.permutation(n_context)[:2 * B].astype(np.uint32)
    if variant == 0:
        heads = np.repeat(np.arange(n_vertex, dtype=np.uint32), B // n_vertex)   # runs of 16, in row order
        hip.set_run_cap(16)
        hip.set_split_hits(4)
    else:   # a row at most once per launch of 96 * 4 / 4 = 96 ... here: 4 launches of 96 samples, each a permutation
        heads = np.concatenate([rng.permutation(n_vertex) for _ in range(B // n_vertex)]).astype(np.uint32)
        hip.set_split_hits(1)
    pairs = np.stack([ctx[:B], heads], 1).astype(np.uint32)
    negs = np.ascontiguousarray(ctx[B:].reshape(B, 1))
    ov, oc = v.copy(), c.copy()
    oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
    hip.set_variant(variant)
    try:
        name = hip.describe_train(dim, "SGD", k, True, B, n_vertex)
        assert name.endswith("in 4 launches per batch") and ("train_runs_kernel" in name) == (variant == 0), name
        hv, hc, hloss, _ = run_hip(hip, v, c, pairs, negs, OPTS["SGD"][1], 5.0)
    finally:
        hip.set_variant(0)
        hip.set_split_hits(2)
        hip.set_run_cap(0)
    np.testing.assert_allclose(hv, ov, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hc, oc, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hloss, oloss, rtol=RTOL, atol=1e-6)


@pytest.mark.parametrize("name", ["Momentum", "Adam"])
def test_runs_with_moment_optimizers(hip, oracle, name):
    opt_id, spec, hp = OPTS[name]
    rng = np.random.default_rng(11)
    N, B, k, dim = 4096, 600, 1, 128
    v, c = init_tables(rng, N, N, dim)
    v *= 20
    c *= 20
    nm = spec.num_moment
    moments = [rng.uniform(0, 1e-3, (N, dim)).astype(np.float32) if i < 2 * nm else None for i in range(4)]
    pairs, negs = _run_batch(rng, N, B, k, [4, 1, 2, 1])  # runs of 4, 1, 2, 1: none crosses a multiple of 8
    ov, oc, om = v.copy(), c.copy(), [None if m is None else m.copy() for m in moments]
    oloss = oracle.train(ov, oc, pairs, negs, spec.lr, spec.weight_decay, 5.0, opt_id, om, hp)
    hip.set_variant(4)
    hip.set_run_cap(8)
    try:
        hv, hc, hloss, hm = run_hip(hip, v, c, pairs, negs, spec, 5.0, moments=moments)
    finally:
        hip.set_run_cap(0)
        hip.set_variant(0)
    np.testing.assert_allclose(hv, ov, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(hc, oc, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(hloss, oloss, rtol=1e-5, atol=1e-6)
    for a, b in zip(hm, om):
        if a is not None:
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-8)


def test_run_cap_splits_long_runs(hip, oracle):
    """A run longer than the cap (a wavefront's segment) is trained by several lane groups (wavefronts), each from the
    row as the launch found it; the row ends up as ONE of their results (last store wins, as between two concurrent
    warps of the reference).  (cap, variant): the runs kernel with caps 4, 16, 1; the per-pair kernel; the segment
    kernel, whose segment is 8 pairs at dim 128."""
    rng = np.random.default_rng(5)
    N, B, k, dim = 4096, 64, 1, 128
    v, c = init_tables(rng, N, N, dim)
    v *= 20
    c *= 20
    pairs, negs = _run_batch(rng, N, B, k, [64])
    row = int(pairs[0, 1])
    for cap, variant in ((4, 4), (16, 4), (1, 4), (1, 2), (8, 0)):
        if variant == 0 and not hip.has_ab_builds:
            continue  # the segment kernel: A/B library only
        candidates = []
        for start in range(0, B, max(cap, 1)):
            ov, oc = v.copy(), c.copy()
            oracle.train(ov, oc, pairs[start:start + max(cap, 1)], negs[start:start + max(cap, 1)], 0.025, 0.005, 5.0)
            candidates.append(ov[row])
        hip.set_run_cap(cap if variant == 4 else 0)
        hip.set_variant(variant)
        try:
            if variant == 0:  # the segment kernel, 2 steps = 8 pairs per wavefront at dim 128
                hip.set_segment_steps(2)
                assert "8 pairs per wavefront" in hip.describe_train(dim, "SGD", k, True, B, N)
            hv, hc, hloss, _ = run_hip(hip, v, c, pairs, negs, OPTS["SGD"][1], 5.0)
        finally:
            hip.set_run_cap(0)
            hip.set_variant(0)
            hip.set_segment_steps(0)
        # the lane groups store the row concurrently — four of them in ONE store instruction of a wavefront, to the same
        # addresses — and a lane's 16 bytes are the unit the memory system arbitrates: every float4 of the row is the
        # matching float4 of ONE of their results (not necessarily the same one for all of them)
        for chunk in range(0, dim, 4):
            assert any(np.allclose(hv[row, chunk:chunk + 4], cand[chunk:chunk + 4], rtol=RTOL, atol=ATOL)
                       for cand in candidates), (cap, variant, chunk)
        assert np.isfinite(hloss).all()


def test_negative_draw_bit_exact(hip, oracle):
    """The on-device draw is integer work: bit-exact against the oracle's restatement of the RNG contract."""
    rng = np.random.default_rng(11)
    w = power_law_weights(rng, 5000)
    prob, alias, packed = K.alias_build(w)
    table = K.packed_to_device(packed, DEV)
    B, k, seed, batch_id = 300, 5, 0x1234567890ABCDEF, 77
    out = torch.zeros(B * k, dtype=torch.int32, device=DEV)
    hip.negative_draw(table, seed, batch_id, out, B, k)
    got = out.cpu().numpy().view(np.uint32).reshape(B, k)
    want = oracle.negatives(prob, alias, seed, batch_id, B, k)
    assert (got == want).all()


def _degree_weights(rng, n):
    return (np.sort(np.floor(rng.pareto(1.2, n) + 1))[::-1] ** 0.75).astype(np.float32)


def test_negative_draw_by_class_bit_exact(hip, oracle):
    """Negatives by weight class (gvk_class_entry table): integer work, bit-exact against the oracle's restatement."""
    rng = np.random.default_rng(13)
    w = _degree_weights(rng, 50000)
    classes = K.class_table_build(w)
    assert classes.size * 8 < w.size
    table = K.classes_to_device(classes, DEV)
    B, k, seed, batch_id = 3000, 5, 0x1234567890ABCDEF, 77
    out = torch.zeros(B * k, dtype=torch.int32, device=DEV)
    hip.negative_draw(table, seed, batch_id, out, B, k)
    got = out.cpu().numpy().view(np.uint32).reshape(B, k)
    want = oracle.negatives_by_class(oracle.class_table(w), seed, batch_id, B, k)
    assert (got == want).all() and got.max() < w.size


@pytest.mark.parametrize("by_class", [False, True])
@pytest.mark.parametrize("dim,k", [(128, 2), (32, 1), (64, 3), (512, 1)])
def test_fused_draw_equals_explicit_negatives(hip, oracle, by_class, dim, k):
    """gvk_train with negatives == NULL trains on exactly the negatives gvk_negative_draw reports — from the row table
    and from the class table: on the pairs whose rows nobody else in the batch touches, the result equals the sequential
    oracle fed those negatives."""
    rng = np.random.default_rng(12)
    N, B = 60000, 512
    v, c = init_tables(rng, N, N, dim)
    v *= 20
    c *= 20
    pairs, _ = conflict_free_batch(rng, N, N, B, 0)
    seed, batch_id = 99, 5
    if by_class:
        w = _degree_weights(rng, N)
        table = K.classes_to_device(K.class_table_build(w), DEV)
        negs = oracle.negatives_by_class(oracle.class_table(w), seed, batch_id, B, k)
    else:
        prob, alias, packed = K.alias_build(power_law_weights(rng, N))
        table = K.packed_to_device(packed, DEV)
        negs = oracle.negatives(prob, alias, seed, batch_id, B, k)
    ov, oc = v.copy(), c.copy()
    oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
    tv, tc = dev(v), dev(c)
    loss = torch.zeros(B, device=DEV)
    hip.train(tv, tc, dev(pairs.view(np.int32)), loss, OPTS["SGD"][1], k, 5.0, table=table, seed=seed,
              batch_id=batch_id)
    torch.cuda.synchronize()
    # pairs whose context rows (tail + negatives) are touched by no other pair: deterministic under Hogwild
    rows = np.concatenate([pairs[:, :1], negs], 1)
    ids, counts = np.unique(rows, return_counts=True)
    shared = set(ids[counts > 1].tolist())
    clean = np.array([not (set(r.tolist()) & shared) for r in rows])
    assert clean.sum() > B // 3
    hv, hc, hl = tv.cpu().numpy(), tc.cpu().numpy(), loss.cpu().numpy()
    np.testing.assert_allclose(hl[clean], oloss[clean], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(hv[pairs[clean, 1]], ov[pairs[clean, 1]], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hc[rows[clean].ravel()], oc[rows[clean].ravel()], rtol=RTOL, atol=ATOL)


def test_sample_pairs_bit_exact(hip, oracle):
    """Device-side positive sampling: integer work, bit-exact against the oracle's restatement."""
    rng = np.random.default_rng(31)
    n_edges, n = 7000, 20000
    prob, alias, packed = K.alias_build(power_law_weights(rng, n_edges))
    block_pairs = rng.integers(0, 1 << 20, (n_edges, 2)).astype(np.uint32)
    pool = torch.zeros(2 * n, dtype=torch.int32, device=DEV)
    seed, first = 0xABCDEF0123456789, (1 << 33) + 5
    hip.sample_pairs(K.packed_to_device(packed, DEV), dev(block_pairs.view(np.int32).reshape(-1)), seed, first, pool, n)
    got = pool.cpu().numpy().view(np.uint32).reshape(n, 2)
    assert (got == oracle.sample_pairs(prob, alias, block_pairs, seed, first, n)).all()
    # the packed form (gvk_sample_edges) draws the same pool
    packed_pool = torch.zeros(2 * n, dtype=torch.int32, device=DEV)
    edge_table = hip.pack_edge_table(K.packed_to_device(packed, DEV), dev(block_pairs.view(np.int32).reshape(-1)))
    hip.sample_edges(edge_table, seed, first, packed_pool, n)
    assert (packed_pool.cpu().numpy().view(np.uint32).reshape(-1, 2) == got).all()


@pytest.mark.parametrize("biased", [False, True])
def test_sample_walks_bit_exact(hip, oracle, biased):
    """Device-side random walks (DeepWalk / node2vec by rejection): bit-exact against the oracle's restatement."""
    import graphvite_amd as gv
    from graphvite_amd import hostlib, synthetic
    g = gv.graph.Graph()
    edges = synthetic.power_law_edges(3000, 30000, seed=4)
    w = np.random.default_rng(1).uniform(0.5, 2.0, len(edges)).astype(np.float32)
    # the directed variant has dead ends, which exercises the chain restart
    g.load([(str(a), str(b), float(c)) for (a, b), c in zip(edges, w)], as_undirected=not biased)
    part, local, _ = hostlib.partition(g.vertex_weights, 1)
    s = hostlib.Sampler(g, part, local, 1, seed=0)
    s.prepare("walk", num_thread=4)
    D = g.num_directed_edge
    nb_prob, nb_alias = [np.ascontiguousarray(a) for a in s.neighbor_tables(D)]
    edge_prob, edge_alias, edge_packed = K.alias_build(g.edge_weights)
    E, flat = g.edges, g.flat_offsets
    order = np.lexsort((E[:, 1], E[:, 0]))
    sorted_nb = np.ascontiguousarray(E[order, 1])
    entry = np.dtype([("prob", np.float32), ("alias", np.uint32)])
    nb = np.zeros(D, entry)
    nb["prob"], nb["alias"] = nb_prob, nb_alias
    walk = {"flat_offsets": dev(flat.astype(np.int64)), "edges_uv": dev(E.view(np.int32).reshape(-1)),
            "edge_table": K.packed_to_device(edge_packed, DEV), "neighbor_table": K.packed_to_device(nb, DEV),
            "local": dev(local.view(np.int32)), "sorted_neighbors": dev(sorted_nb.view(np.int32)), "biased": biased,
            "p": 0.5, "q": 2.0}
    L, aug, sb, pool_pairs, seed, first = 12, 3, 3, 3 * 33333, 77, (1 << 32) + 9
    pool = torch.zeros(2 * pool_pairs, dtype=torch.int32, device=DEV)
    hip.sample_walks(walk, seed, first, pool, pool_pairs, L, aug, sb)
    got = pool.cpu().numpy().view(np.uint32).reshape(-1, 2)
    want = oracle.sample_walks_device(flat, E, edge_prob, edge_alias, nb_prob, nb_alias, sorted_nb, local, biased, 0.5, 2.0,
                                      seed, first, pool_pairs, L, aug, sb)
    assert (got == want).all()


@pytest.mark.parametrize("biased", [False, True])
def test_sample_walks_blocks_matches_the_oracle_per_block(hip, oracle, biased):
    """Random walks binned per (head partition, tail partition) block on the device (gvk_sample_walks_blocks): the walks
    are those of gvk_sample_walks — pinned bit for bit above — so with pools large enough to hold everything, every
    block's pool must hold exactly the pairs of the oracle's walks that belong to it (as a multiset: slots are handed out
    by atomics), in local ids, and the counters the shares.  With small pools: full, and only pairs of the block."""
    import ctypes as C
    import graphvite_amd as gv
    from graphvite_amd import hostlib, synthetic
    g = gv.graph.Graph()
    edges = synthetic.power_law_edges(3000, 30000, seed=4)
    w = np.random.default_rng(1).uniform(0.5, 2.0, len(edges)).astype(np.float32)
    g.load([(str(a), str(b), float(c)) for (a, b), c in zip(edges, w)], as_undirected=not biased)
    P = 3
    part, local, _ = hostlib.partition(g.vertex_weights, P)
    s = hostlib.Sampler(g, part, local, P, seed=0)
    s.prepare("walk", num_thread=4)
    D = g.num_directed_edge
    nb_prob, nb_alias = [np.ascontiguousarray(a) for a in s.neighbor_tables(D)]
    edge_prob, edge_alias, edge_packed = K.alias_build(g.edge_weights)
    E, flat = g.edges, g.flat_offsets
    sorted_nb = np.ascontiguousarray(E[np.lexsort((E[:, 1], E[:, 0])), 1])
    nb = np.zeros(D, np.dtype([("prob", np.float32), ("alias", np.uint32)]))
    nb["prob"], nb["alias"] = nb_prob, nb_alias
    walk = {"flat_offsets": dev(flat.astype(np.int64)), "edges_uv": dev(E.view(np.int32).reshape(-1)),
            "edge_table": K.packed_to_device(edge_packed, DEV), "neighbor_table": K.packed_to_device(nb, DEV),
            "local": dev(local.view(np.int32)), "sorted_neighbors": dev(sorted_nb.view(np.int32)), "biased": biased,
            "p": 0.5, "q": 2.0}
    L, aug, seed, first, walks = 12, 3, 77, (1 << 32) + 9, 5000
    per_walk = aug * L - aug * (aug - 1) // 2
    identity = np.arange(g.num_vertex, dtype=np.uint32)
    want = oracle.sample_walks_device(flat, E, edge_prob, edge_alias, nb_prob, nb_alias, sorted_nb, identity, biased, 0.5,
                                      2.0, seed, first, walks * per_walk, L, aug, 1)  # {tail vertex, head vertex}
    block = part[want[:, 1]].astype(np.int64) * P + part[want[:, 0]]
    stripes, sb = 7, 3
    # pair i of a walk of wavefront w (64 walks) goes to stripe (w + (i % sb) * (stripes // sb)) % stripes: the pseudo shuffle's
    # parts (graph.cuh:362-364,439-441) chosen by the pair's index in its walk
    index = np.arange(len(want))
    stripe_of = (index // per_walk // 64 + index % per_walk % sb * (stripes // sb)) % stripes
    per_stripe = max(np.bincount(block[stripe_of == k], minlength=P * P).max() for k in range(stripes))
    capacity = (int(per_stripe) + 3) * stripes
    capacity += -capacity % (sb * stripes)
    where = np.arange(P * P, dtype=np.int64) * capacity
    where[4] = -1  # one block is not collected
    pools = torch.zeros(P * P * capacity * 2, dtype=torch.int32, device=DEV)
    counters = torch.zeros(P * P * stripes, dtype=torch.int32, device=DEV)
    desc = hip._walk_graph(walk, torch.device(DEV))
    part_dev, where_dev = dev(part.astype(np.int32)), dev(where)
    rc = hip.lib.gvk_sample_walks_blocks(None, C.byref(desc), part_dev.data_ptr(), P, seed, first, walks, pools.data_ptr(),
                                         where_dev.data_ptr(), counters.data_ptr(), capacity, stripes, L, aug, sb)
    assert rc == 0
    torch.cuda.synchronize()
    got = pools.cpu().numpy().view(np.uint32).reshape(P * P, capacity, 2)
    count = counters.cpu().numpy().reshape(P * P, stripes)
    for b in range(P * P):
        if b == 4:
            assert not count[b].any() and not got[b].any()
            continue
        for k in range(stripes):
            mine = want[(block == b) & (stripe_of == k)]
            assert count[b, k] == len(mine)
            stored = got[b][k * (capacity // stripes) + np.arange(len(mine))]
            expect = np.stack([local[mine[:, 0]], local[mine[:, 1]]], 1)
            assert sorted(map(tuple, stored.tolist())) == sorted(map(tuple, expect.tolist()))
    # thinned (gvk_sample_walks_blocks_thinned): every block keeps a pair with a probability of its own, decided by a hash of (walk, pair index in
    # the walk, seed) — not by when the pair arrives: exactly the oracle's pairs the rule keeps, per block and stripe
    from util import thinning_uniform
    accept = np.random.default_rng(5).uniform(0.2, 0.9, P * P).astype(np.float32)
    accept[0], accept[-1] = 1.0, 0.0
    keep = thinning_uniform(first + index // per_walk, index % per_walk, seed) < accept[block]
    pools.zero_(), counters.zero_()
    rc = hip.lib.gvk_sample_walks_blocks_thinned(None, C.byref(desc), part_dev.data_ptr(), P, seed, first, walks, pools.data_ptr(), where_dev.data_ptr(),
                                                 counters.data_ptr(), capacity, stripes, L, aug, sb, dev(accept).data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    got = pools.cpu().numpy().view(np.uint32).reshape(P * P, capacity, 2)
    count = counters.cpu().numpy().reshape(P * P, stripes)
    assert not count[P * P - 1].any() and count[0].sum() == np.count_nonzero(block == 0)
    for b in range(P * P):
        if b == 4:
            assert not count[b].any()
            continue
        for k in range(stripes):
            mine = want[(block == b) & (stripe_of == k) & keep]
            assert count[b, k] == len(mine), (b, k, count[b, k], len(mine))
            stored = got[b][k * (capacity // stripes) + np.arange(len(mine))]
            expect = np.stack([local[mine[:, 0]], local[mine[:, 1]]], 1)
            assert sorted(map(tuple, stored.tolist())) == sorted(map(tuple, expect.tolist()))
    # the wrapper: small pools, repeated until every collected pool is full; only pairs of the block, local ids
    small = 600
    pools = torch.full((P * P * small * 2,), -1, dtype=torch.int32, device=DEV)
    where = np.arange(P * P, dtype=np.int64) * small
    used = hip.sample_walks_blocks(walk, part_dev, P, seed, first, pools, dev(where), small, L, aug, sb)
    assert used >= P * P * small // per_walk
    got = pools.cpu().numpy().view(np.uint32).reshape(P * P, small, 2)
    sizes = np.bincount(part, minlength=P)
    for b in range(P * P):
        assert (got[b][:, 1] < sizes[b // P]).all() and (got[b][:, 0] < sizes[b % P]).all()


def test_alias_sample_matches_reference_semantics(hip, oracle):
    rng = np.random.default_rng(13)
    prob, alias, packed = K.alias_build(power_law_weights(rng, 1000))
    table = K.packed_to_device(packed, DEV)
    n = 4096
    rand = rng.random(2 * n)
    out = torch.zeros(n, dtype=torch.int32, device=DEV)
    hip.alias_sample(table, dev(rand), out)
    got = out.cpu().numpy().view(np.uint32)
    want = np.array([oracle.alias_sample_gpu(prob, alias, rand[2 * i], rand[2 * i + 1]) for i in range(n)])
    assert (got == want).all()


@pytest.mark.parametrize("dim", [32, 64, 96, 128, 256, 512])
def test_predict_matches_oracle(hip, oracle, dim):
    rng = np.random.default_rng(dim)
    N, B = 2000, 3333
    v, c = init_tables(rng, N, N, dim)
    pairs, _ = random_batch(rng, N, N, B, 0)
    logits = torch.zeros(B, device=DEV)
    hip.predict(dev(v), dev(c), dev(pairs.view(np.int32)), logits)
    np.testing.assert_allclose(logits.cpu().numpy(), oracle.predict(v, c, pairs), rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("dim", [32, 64, 96, 128, 256, 512])
def test_row_traffic_probe_touches_exactly_the_rows_of_a_batch(hip, dim):
    """gvk_probe_row_traffic (bench.py's `roofline.access_pattern`): every element of the head, tail and negative row of
    every pair read, bumped and written back, nothing else touched; bump 0 leaves the tables bit-identical."""
    rng = np.random.default_rng(dim)
    N, B = 4096, 777
    v, c = init_tables(rng, N, N, dim)
    pairs, negs = conflict_free_batch(rng, N, N, B, 1)
    tv, tc, tp, tn = dev(v), dev(c), dev(pairs.view(np.int32)), dev(negs.view(np.int32).reshape(-1))
    hip.probe_row_traffic(tv, tc, tp, tn)
    assert (tv.cpu().numpy() == v).all() and (tc.cpu().numpy() == c).all()
    hip.probe_row_traffic(tv, tc, tp, tn, bump=1.0)
    want_v, want_c = v.copy(), c.copy()
    want_v[pairs[:, 1]] += 1
    want_c[pairs[:, 0]] += 1
    want_c[negs[:, 0]] += 1
    assert (tv.cpu().numpy() == want_v).all() and (tc.cpu().numpy() == want_c).all()
    with pytest.raises(ValueError, match="dim"):
        hip.probe_row_traffic(torch.zeros((4, 48), device=DEV), torch.zeros((4, 48), device=DEV), tp, tn)


def test_hogwild_batch_statistics(hip, oracle):
    """Arbitrary batch with conflicts (T2): batch-mean loss and row norms track the sequential oracle."""
    rng = np.random.default_rng(21)
    N, B, k, dim = 2000, 20000, 1, 128
    v, c = init_tables(rng, N, N, dim)
    v *= 10
    c *= 10
    pairs, negs = random_batch(rng, N, N, B, k)
    ov, oc = v.copy(), c.copy()
    oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
    hv, hc, hloss, _ = run_hip(hip, v, c, pairs, negs, OPTS["SGD"][1], 5.0)
    # every row is hit ~10 times inside this one batch: the parallel kernel loses most same-row updates that
    # the sequential oracle applies one after another, so only aggregate agreement is expected
    assert abs(hloss.mean() - oloss.mean()) <= 1e-3 * abs(oloss.mean())
    assert abs(np.linalg.norm(hv) - np.linalg.norm(ov)) <= 3e-2 * np.linalg.norm(ov)
    assert abs(np.linalg.norm(hc) - np.linalg.norm(oc)) <= 3e-2 * np.linalg.norm(oc)


def test_benchmark_size_batch_against_oracle(hip, oracle):
    """BASELINE.json's size (1M rows x dim 128, one 100 000-pair batch, negatives drawn in-kernel from a power-law
    table, real conflicts): every pair whose three rows nobody else touches must equal the sequential oracle to fp32
    tolerance, every untouched row must be bit-identical, and the batch loss must agree."""
    rng = np.random.default_rng(2026)
    N, B, k, dim = 1000000, 100000, 1, 128
    v, c = init_tables(rng, N, N, dim)
    v *= 20
    c *= 20
    w = power_law_weights(rng, N)
    order = np.argsort(-w, kind="stable")  # local ids are degree ranks, as in the solver
    prob, alias, packed = K.alias_build(w[order])
    p = w[order].astype(np.float64)
    p /= p.sum()
    heads, tails = rng.choice(N, B, p=p), rng.choice(N, B, p=p)
    pairs = np.stack([tails, heads], 1).astype(np.uint32)
    seed, batch_id = 7, 123
    negs = oracle.negatives(prob, alias, seed, batch_id, B, k)
    ov, oc = v.copy(), c.copy()
    oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
    tv, tc = dev(v), dev(c)
    loss = torch.zeros(B, device=DEV)
    hip.train(tv, tc, dev(pairs.view(np.int32)), loss, OPTS["SGD"][1], k, 5.0, table=K.packed_to_device(packed, DEV),
              seed=seed, batch_id=batch_id)
    torch.cuda.synchronize()
    hv, hc, hl = tv.cpu().numpy(), tc.cpu().numpy(), loss.cpu().numpy()
    ctx_rows = np.concatenate([pairs[:, :1], negs], 1)
    cid, ccount = np.unique(ctx_rows, return_counts=True)
    hid, hcount = np.unique(pairs[:, 1], return_counts=True)
    shared_c, shared_h = set(cid[ccount > 1].tolist()), set(hid[hcount > 1].tolist())
    clean = np.array([h not in shared_h and not (set(r.tolist()) & shared_c)
                      for h, r in zip(pairs[:, 1].tolist(), ctx_rows)])
    assert clean.sum() > B // 20
    np.testing.assert_allclose(hl[clean], oloss[clean], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(hv[pairs[clean, 1]], ov[pairs[clean, 1]], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hc[ctx_rows[clean].ravel()], oc[ctx_rows[clean].ravel()], rtol=RTOL, atol=ATOL)
    untouched_v = np.ones(N, bool)
    untouched_v[pairs[:, 1]] = False
    untouched_c = np.ones(N, bool)
    untouched_c[ctx_rows.ravel()] = False
    assert (hv[untouched_v] == v[untouched_v]).all() and (hc[untouched_c] == c[untouched_c]).all()
    assert abs(hl.mean() - oloss.mean()) <= 5e-3 * abs(oloss.mean())


def test_empty_and_ragged_batches(hip, oracle):
    rng = np.random.default_rng(5)
    N, dim = 1024, 128
    v, c = init_tables(rng, N, N, dim)
    spec = OPTS["SGD"][1]
    # empty batch: nothing happens, no error
    tv, tc = dev(v), dev(c)
    hip.train(tv, tc, torch.zeros((0, 2), dtype=torch.int32, device=DEV), torch.zeros(1, device=DEV), spec, 1, 5.0,
              negatives=torch.zeros(0, dtype=torch.int32, device=DEV))
    torch.cuda.synchronize()
    assert (tv.cpu().numpy() == v).all() and (tc.cpu().numpy() == c).all()
    # batch sizes that do not fill a wavefront / a block
    for B in (1, 3, 63, 65, 257):
        pairs, negs = conflict_free_batch(rng, N, N, B, 1)
        ov, oc = v.copy(), c.copy()
        oloss = oracle.train(ov, oc, pairs, negs, 0.025, 0.005, 5.0)
        hv, hc, hloss, _ = run_hip(hip, v, c, pairs, negs, spec, 5.0)
        np.testing.assert_allclose(hv, ov, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(hc, oc, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(hloss, oloss, rtol=RTOL, atol=1e-6)


def test_errors_are_returned_not_aborted(hip):
    v = torch.zeros((8, 100), device=DEV)  # dim 100 is not instantiated
    with pytest.raises(ValueError):
        hip.train(v, v.clone(), torch.zeros((1, 2), dtype=torch.int32, device=DEV), torch.zeros(1, device=DEV),
                  OPTS["SGD"][1], 0, 5.0)
    v = torch.zeros((8, 128), device=DEV)
    with pytest.raises(ValueError):  # negatives requested but no source
        hip.train(v, v.clone(), torch.zeros((1, 2), dtype=torch.int32, device=DEV), torch.zeros(1, device=DEV),
                  OPTS["SGD"][1], 1, 5.0)
    with pytest.raises(ValueError):  # Adam without moment tables
        hip.train(v, v.clone(), torch.zeros((1, 2), dtype=torch.int32, device=DEV), torch.zeros(1, device=DEV),
                  OPTS["Adam"][1], 0, 5.0)


def test_rows_beyond_4_gib(hip, oracle):
    """Friendster-sized shards: row ids whose byte offsets need more than 32 bits (9.4M rows x 128 x 4 B = 4.8 GB per
    table).  The device tables hold the full row range; the oracle gets the touched rows compacted."""
    dim, N, B, k = 128, 9_400_000, 512, 2
    rng = np.random.default_rng(77)
    lo = (1 << 32) // (dim * 4)  # first row whose offset does not fit in 32 bits
    heads = (lo + rng.permutation(N - lo)[:B]).astype(np.uint32)
    ctx = (lo + rng.permutation(N - lo)[:B * (k + 1)]).astype(np.uint32)
    pairs = np.stack([ctx[:B], heads], 1).astype(np.uint32)
    negs = np.ascontiguousarray(ctx[B:].reshape(B, k))
    tv = torch.zeros((N, dim), dtype=torch.float32, device=DEV)
    tc = torch.zeros((N, dim), dtype=torch.float32, device=DEV)
    v_rows, c_rows = init_tables(rng, B, B * (k + 1), dim)
    v_rows *= 20
    c_rows *= 20
    tv[torch.from_numpy(heads.astype(np.int64)).to(DEV)] = dev(v_rows)
    tc[torch.from_numpy(ctx.astype(np.int64)).to(DEV)] = dev(c_rows)
    # compact ids for the oracle: head i -> i, context row ctx[j] -> j
    opairs = np.stack([np.arange(B), np.arange(B)], 1).astype(np.uint32)
    onegs = np.ascontiguousarray(np.arange(B, B * (k + 1)).reshape(B, k).astype(np.uint32))
    ov, oc = v_rows.copy(), c_rows.copy()
    oloss = oracle.train(ov, oc, opairs, onegs, 0.025, 0.005, 5.0)
    loss = torch.zeros(B, dtype=torch.float32, device=DEV)
    hip.train(tv, tc, dev(pairs.view(np.int32)), loss, OPTS["SGD"][1], k, 5.0, negatives=dev(negs.view(np.int32)))
    torch.cuda.synchronize()
    np.testing.assert_allclose(tv[torch.from_numpy(heads.astype(np.int64)).to(DEV)].cpu().numpy(), ov, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(tc[torch.from_numpy(ctx.astype(np.int64)).to(DEV)].cpu().numpy(), oc, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(loss.cpu().numpy(), oloss, rtol=RTOL, atol=1e-6)
    logits = torch.zeros(B, device=DEV)
    hip.predict(tv, tc, dev(pairs.view(np.int32)), logits)
    np.testing.assert_allclose(logits.cpu().numpy(), oracle.predict(ov, oc, opairs), rtol=1e-5, atol=1e-8)
    # nothing below the 4 GiB line moved
    assert float(tv[:lo].abs().max()) == 0.0 and float(tc[:lo].abs().max()) == 0.0


@pytest.mark.parametrize("nb,B,rows", [(1, 1, 2), (3, 1000, 777), (7, 4097, 1 << 20), (217, 100000, 1000000),
                                       (2, 5001, 1 << 24), (2, 3000, 2 ** 31), (1, 70000, 2 ** 32)])
def test_group_pairs_is_a_stable_per_batch_sort_on_head(hip, nb, B, rows):
    """1, 2, 3 and 4 counting passes (10 bits of the head row per pass); ties keep their input order."""
    rng = np.random.default_rng(nb * B)
    if rows > 1 << 20:  # heads all over the row range, with repeats
        heads = rng.integers(0, rows, max(nb * B // 3, 1), dtype=np.uint64)[rng.integers(0, max(nb * B // 3, 1), nb * B)]
    else:
        heads = (rng.pareto(1.2, nb * B) * 20).astype(np.uint64) % np.uint64(rows)
    pool = np.stack([rng.integers(0, 2 ** 32, nb * B, dtype=np.uint64).astype(np.uint32), heads.astype(np.uint32)], 1)
    tin, tout = dev(pool.view(np.int32)), torch.zeros((nb * B, 2), dtype=torch.int32, device=DEV)
    hip.group_pairs(tin, tout, B, nb, rows)
    torch.cuda.synchronize()
    got = tout.cpu().numpy().view(np.uint32).reshape(nb, B, 2)
    rec = pool.reshape(nb, B, 2)
    for i in range(0, nb, max(nb // 7, 1)):
        want = rec[i][np.argsort(rec[i, :, 1], kind="stable")]
        assert (got[i] == want).all()
    assert (tin.cpu().numpy().view(np.uint32) == pool).all()  # the input pool is left alone


def test_group_pairs_edge_cases(hip):
    from graphvite_amd import _lib
    import ctypes as C
    a = torch.zeros((8, 2), dtype=torch.int32, device=DEV)
    b = torch.full((8, 2), -1, dtype=torch.int32, device=DEV)
    hip.group_pairs(a, b, 4, 0, 10)  # no batches: nothing written
    torch.cuda.synchronize()
    assert (b.cpu().numpy() == -1).all()
    need = C.c_size_t(64)
    lib = _lib.lib()
    assert lib.gvk_group_pairs(None, a.data_ptr(), a.data_ptr(), a.data_ptr(), C.byref(need), 4, 2, 10) == _lib.GVK_EINVAL
    assert lib.gvk_group_pairs(None, a.data_ptr(), b.data_ptr(), a.data_ptr(), C.byref(need), 4, 2, 0) == _lib.GVK_EINVAL
    assert lib.gvk_group_pairs(None, None, None, None, None, 4, 2, 10) == _lib.GVK_EINVAL
    with pytest.raises(ValueError):
        hip.group_pairs(a, b, 4, 3, 10)


def test_spread_pairs_sends_consecutive_records_to_consecutive_units(hip):
    """gvk_spread_pairs: record i of the pool to place (i % units) * (n / units) + i / units — unit u holds the records u, u +
    units, ...: what sat side by side in a walk-ordered pool is trained by consecutive launches."""
    rng = np.random.default_rng(3)
    for n, units in ((1200, 8), (4096, 64), (1000, 1), (600, 600)):
        pool = rng.integers(0, 1 << 31, (n, 2)).astype(np.uint32)
        a = torch.from_numpy(pool.view(np.int32)).to(DEV)
        b = torch.zeros_like(a)
        hip.spread_pairs(a, b, n, units)
        torch.cuda.synchronize()
        want = pool.reshape(n // units, units, 2).transpose(1, 0, 2).reshape(n, 2)
        assert (b.cpu().numpy().view(np.uint32) == want).all()
    with pytest.raises(ValueError):
        hip.spread_pairs(a, b, 600, 7)
