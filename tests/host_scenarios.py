"""TEST INFRASTRUCTURE: scenarios that drive the solver ENGINE (graphvite_amd/csrc/gvx_engine.cpp, through graphvite_amd.solver)
on a machine without a GPU.  They run in a process of their own with GVK_LIBRARY = tests/hostdev/build/libgvk_host.so — the
engine's own sources over a host stand-in for the HIP runtime, its kernels being the CPU oracle — so that partitioning,
sampling, pool handling, batch-id / learning-rate accounting, the slot claim + all-gather exchange (in one process: several
workers; across processes: gloo through the engine's transport hook), the routing of walk pools and write-back can be
checked here.  tests/test_solver_cpu.py starts one process per scenario:

    GVK_LIBRARY=tests/hostdev/build/libgvk_host.so python tests/host_scenarios.py <scenario> [json arguments]
"""
import ctypes as C
import json
import logging
import os
import pickle
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
HOST_LIBRARY = os.path.join(ROOT, "tests", "hostdev", "build", "libgvk_host.so")

import graphvite_amd as gv  # noqa: E402
from graphvite_amd import _lib, hostlib, synthetic  # noqa: E402
from oracle_lib import link_prediction_auc  # noqa: E402


def make_graph(n=300, e=3000, seed=1):
    g = gv.graph.Graph()
    g.load(synthetic.power_law_edges(n, e, seed=seed))
    return g


def ascending_heads(s, rec):
    """Regrouped batches: the heads of every PART of a batch ascend (a part is what one launch trains, gvk_train_launches)."""
    parts = _lib.lib().gvk_train_launches(len(rec), s.partition_rows)
    heads = rec[:, 1].astype(np.int64).reshape(parts, -1)
    return bool((np.diff(heads, axis=1) >= 0).all())


class Launches(object):
    """Every batch the host kernels trained since clear(): batch ids and learning rates (tests/hostdev/host_kernels.cpp)."""

    def __init__(self):
        self.lib = _lib.lib()
        self.lib.gvh_launch_log.restype = C.c_size_t
        self.lib.gvh_launch_log.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]

    def clear(self):
        self.lib.gvh_launch_log_clear()

    def get(self):
        n = self.lib.gvh_launch_log(None, None, None, None, 0)
        ids, lrs = np.zeros(n, np.uint32), np.zeros(n, np.float32)
        self.lib.gvh_launch_log(ids.ctypes.data, lrs.ctypes.data, None, None, n)
        return ids.astype(np.int64), lrs


OBSERVER = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p)


class Batches(object):
    """Copies of the batches the host kernels are about to train (the observer hook of the host build)."""

    def __init__(self):
        self.batches = []
        self._callback = OBSERVER(self._see)
        _lib.lib().gvh_set_batch_observer(self._callback)

    def _see(self, pairs, batch_size, batch_id, vertex, context):
        rec = np.ctypeslib.as_array(C.cast(pairs, C.POINTER(C.c_uint32)), shape=(batch_size, 2)).copy()
        self.batches.append((int(batch_id), rec))

    def close(self):
        _lib.lib().gvh_set_batch_observer(OBSERVER())


# ---- one process ---------------------------------------------------------------------------------------------------------

def build_defaults():
    g = make_graph()
    s = gv.solver.GraphSolver(128, num_sampler_per_worker=2)
    s.build(g)
    assert s.optimizer.type == "SGD" and s.optimizer.init_lr == 0.025 and s.optimizer.weight_decay == 0.005
    assert s.optimizer.schedule.type == "linear"
    assert s.num_partition == 1 and s.num_negative == 1 and s.batch_size == 100000
    assert s.episode_size == 200  # max(300 * 175 / 1 / 1e5, 1) -> 1, single partition -> 2e7 / 1e5 (solver.h:426-436)
    assert s.vertex_embeddings.shape == (g.num_vertex, 128)
    s.build(g, optimizer=0.1, batch_size=500, episode_size=3)  # bare learning rate keeps the default optimizer
    assert s.optimizer.type == "SGD" and s.optimizer.init_lr == np.float32(0.1)
    s.build(g, optimizer=gv.optimizer.Adam(1e-3), batch_size=500, episode_size=3)
    assert s.optimizer.type == "Adam" and s.num_moment == 2
    for bad in (lambda: s.build(g, num_partition=-1), lambda: s.train(model="TransE"),
                lambda: gv.solver.GraphSolver(32, pair_order="sorted"), lambda: s.session(modle="LINE")):
        try:
            bad()
        except (ValueError, TypeError):
            continue
        raise AssertionError("an invalid argument was accepted")
    for bad in (lambda: gv.solver.GraphSolver(100), lambda: gv.solver.GraphSolver(128, float_type=gv.float64)):
        try:
            bad()
        except AttributeError:
            continue
        raise AssertionError("an instantiation that does not exist was accepted")


def hub_row_rules():
    """Which rows the engine hands to chains (GVX_HUB_ROWS / GVX_HUB_PARTS / GVX_FIDELITY).  The host build's
    kernels are sequential — what chains restore — so the trained tables must not depend on the choice."""
    g = make_graph(n=2000, e=30000)
    trained = {}
    for name, kw, model, want in (("default, LINE", {}, "LINE", None), ("off", dict(hub_rows=0), "DeepWalk", 0),
                                  ("default, DeepWalk: every row", {}, "DeepWalk", g.num_vertex),
                                  ("by expected hits", dict(hub_rows="auto"), "LINE", None), ("given", dict(hub_rows=100), "LINE", 100),
                                  ("more than there are", dict(hub_rows=10 ** 6), "LINE", g.num_vertex),
                                  ("fidelity", dict(fidelity="reference"), "LINE", None)):
        s = gv.solver.GraphSolver(32, num_sampler_per_worker=2, seed=1, **kw)
        s.build(g, batch_size=1000, episode_size=4)
        s.train(model=model, num_epoch=2, augmentation_step=1 if model == "LINE" else 2, log_frequency=1 << 30)
        if want is None:  # rows a batch (1000 samples) is expected to hit once or more: some of the 2000
            assert 0 < s.hub_rows < g.num_vertex, (name, s.hub_rows)
            trained.setdefault("expected hits", s.hub_rows)
            assert trained["expected hits"] == s.hub_rows
        else:
            assert s.hub_rows == want, (name, s.hub_rows, want)
        # ... among the runs that train the pools in the same order (a walk-ordered pool is spread over the launches unless
        # chains own every row, gvk_spread_pairs: the same samples in another order)
        trained.setdefault((model, s.pair_order), s.vertex_embeddings.copy())
        assert (trained[model, s.pair_order] == s.vertex_embeddings).all(), name
        assert s.pair_order == ("spread" if model == "DeepWalk" and s.hub_rows < g.num_vertex else "sampled"), (name, s.pair_order)
    # several workers / partitions with hub rows: the engine's bookkeeping (work lists per block visit, per-worker workspaces,
    # partitions with different numbers of hub rows) — the tables must still be those of the plain run
    plain = None
    for hub, parts, fidelity in ((0, 0, "throughput"), (40, 0, "throughput"), ("auto", 4, gv.auto), (None, 0, "reference")):
        s = gv.solver.GraphSolver(32, device_ids=[0, 0], num_sampler_per_worker=1, seed=4, hub_rows=hub, fidelity=fidelity)
        s.hub_parts = parts
        s.build(g, batch_size=1000, episode_size=3, num_partition=4)
        s.train(model="LINE", num_epoch=2, augmentation_step=1, log_frequency=1 << 30)
        assert (s.hub_rows > 0) == (hub != 0)
        plain = s.vertex_embeddings.copy() if plain is None else plain
        assert (plain == s.vertex_embeddings).all(), (hub, parts, fidelity)
    # several partitions of walk-ordered pools: hub rows by expected hits
    s = gv.solver.GraphSolver(32, num_sampler_per_worker=2, seed=1)
    s.build(g, batch_size=1000, episode_size=4, num_partition=2)
    s.train(model="DeepWalk", num_epoch=1, augmentation_step=2, log_frequency=1 << 30)
    assert 0 < s.hub_rows <= g.num_vertex // 2  # per partition (1000 rows each)
    # the moment optimizers have chains of their own (one sequential task per hub row: gvk_chains.hip train_moment_chains): hub rows by the
    # same rule, asked for explicitly honoured, fidelity="throughput" turns them off
    s = gv.solver.GraphSolver(32, num_sampler_per_worker=2, seed=1)
    s.build(g, optimizer=gv.optimizer.Adam(1e-3), batch_size=1000, episode_size=4)
    s.train(model="DeepWalk", num_epoch=1, augmentation_step=2, log_frequency=1 << 30)
    assert s.hub_rows > 0
    for kw, rows in ((dict(hub_rows=50), 50), (dict(fidelity="reference"), None), (dict(fidelity="throughput"), 0)):
        s = gv.solver.GraphSolver(32, num_sampler_per_worker=2, seed=1, **kw)
        s.build(g, optimizer=gv.optimizer.Adam(1e-3), batch_size=1000, episode_size=4)
        s.train(model="LINE", num_epoch=1, log_frequency=1 << 30)
        assert (s.hub_rows > 0) if rows is None else (s.hub_rows == rows), (kw, s.hub_rows)
    # a schedule computed by a callback: chains all the same (a call per batch, its learning rate from the host)
    tables = {}
    for name, schedule in (("linear", "linear"), ("callback", lambda batch_id, num_batch: max(1 - batch_id / num_batch, 1e-4))):
        s = gv.solver.GraphSolver(32, num_sampler_per_worker=2, seed=1, hub_rows=50)
        s.build(g, optimizer=gv.optimizer.SGD(0.025, 0.005, schedule), batch_size=1000, episode_size=4)
        s.train(model="LINE", num_epoch=2, log_frequency=1 << 30)
        assert s.hub_rows == 50
        tables[name] = s.vertex_embeddings.copy()
    assert np.abs(tables["linear"] - tables["callback"]).max() < 1e-6
    s = gv.solver.GraphSolver(32, num_sampler_per_worker=2, seed=1, hub_rows=50)
    s.hub_parts = 7  # not a divisor of the batch size: an error, not a silent fall-back to the rule
    s.build(g, batch_size=1000, episode_size=4)
    try:
        s.train(model="LINE", num_epoch=1, log_frequency=1 << 30)
    except ValueError as e:
        assert "must divide the batch size" in str(e)
    else:
        raise AssertionError("hub_parts = 7 does not divide a batch of 1000")
    # the embedding views keep the solver that owns their memory alive (the reference's binding does the same, bind.h:90-106)
    def views():
        t = gv.solver.GraphSolver(32, num_sampler_per_worker=1, seed=2)
        t.build(g, batch_size=1000, episode_size=2)
        t.train(model="LINE", num_epoch=1, log_frequency=1 << 30)
        return t.vertex_embeddings, t.context_embeddings
    import gc
    v, c = views()
    gc.collect()
    assert v.shape == (g.num_vertex, 32) and np.isfinite(v).all() and np.isfinite(c).all() and np.abs(c).max() > 0
    for bad in (lambda: gv.solver.GraphSolver(32, fidelity="exact"), lambda: gv.solver.GraphSolver(32, hub_rows=-5).build(g)):
        try:
            bad()
        except ValueError:
            continue
        raise AssertionError("an invalid argument was accepted")


def executor_simulator():
    """GVH_EXECUTOR (tests/hostdev/host_kernels.cpp): the host build trains hub rows the way a device executor would — unit by unit
    ("units"), the chains of unit u + 1 before the pairs of unit u write ("pipelined": the product's launch), the chains of a whole
    batch ahead of the pairs ("batchahead"), the pairs reading hub rows on a straight line across the batch ("batchlerp") —, so
    that a change of the device path can be judged by what it learns before it is written.  Every form trains the same samples:
    the tables stay close to the sequential ones, and differ from them (the forms are not the sequential loop)."""
    g = make_graph(n=2000, e=30000)
    trained = {}
    for executor in ("sequential", "units", "pipelined", "batchahead", "batchlerp", "pipelined, concurrent pairs"):
        os.environ["GVH_EXECUTOR"] = executor.split(",")[0]
        os.environ["GVH_PAIRS"] = "concurrent" if "concurrent" in executor else "in order"  # Hogwild inside a unit's pairs, modelled
        s = gv.solver.GraphSolver(32, num_sampler_per_worker=2, seed=1, hub_rows=200)
        s.hub_parts = 4
        s.build(g, batch_size=1000, episode_size=4)
        s.train(model="LINE", num_epoch=20, augmentation_step=1, log_frequency=1 << 30)
        assert s.hub_rows == 200
        trained[executor] = np.concatenate([s.vertex_embeddings.ravel(), s.context_embeddings.ravel()]).astype(np.float64)
        assert np.isfinite(trained[executor]).all(), executor
    os.environ.pop("GVH_EXECUTOR"), os.environ.pop("GVH_PAIRS")
    base = trained["sequential"]
    for executor, x in trained.items():
        cosine = float(x @ base / np.sqrt((x @ x) * (base @ base)))
        print("executor %-28s cosine to sequential %.5f" % (executor, cosine))
        assert cosine > 0.995, (executor, cosine)
        assert executor == "sequential" or not (x == base).all(), executor


def accounting_and_determinism():
    g = make_graph()
    log = Launches()
    runs = []
    for _ in range(2):
        log.clear()
        s = gv.solver.GraphSolver(64, num_sampler_per_worker=2, seed=3)
        s.build(g, batch_size=500, episode_size=6)
        view = s.vertex_embeddings
        s.train("LINE", num_epoch=5, log_frequency=7)
        assert np.shares_memory(view, s.vertex_embeddings)  # stable host buffers behind the numpy views
        runs.append((s.vertex_embeddings.copy(), s.context_embeddings.copy(), log.get(), s))
    v0, c0, (ids, lrs), s = runs[0]
    assert (v0 == runs[1][0]).all() and (c0 == runs[1][1]).all()  # same seed -> same pools, negatives, result
    # num_batch = num_epoch * |E| / batch (solver.h:611), overshoot to whole episodes (solver.h:629)
    assert s.num_batch == 5 * 3000 // 500 and s.augmentation_step == 3 and s.shuffle_base == 3
    assert ids.tolist() == list(range(30)) and s.batch_id == 30
    want = np.float32(0.025) * np.maximum(1 - ids / 30.0, 1e-4)  # lr = init_lr * max(1 - b / num_batch, 1e-4) (optimizer.h:77-79)
    np.testing.assert_allclose(lrs, want, rtol=1e-6)
    assert np.abs(c0).max() > 0 and np.isfinite(v0).all()
    before = s.vertex_embeddings.copy()  # resume continues the batch counter and does not re-initialise
    s.train("LINE", num_epoch=1, resume=True, log_frequency=1000)
    assert s.num_batch == 30 + 6 and s.batch_id == 36 and not (s.vertex_embeddings == before).all()


def grouped_pair_order():
    """pair_order="grouped": every batch trains the same multiset of pairs as with "sampled", heads adjacent."""
    g = make_graph()
    seen = {}
    for order in ("sampled", "grouped"):
        watch = Batches()
        s = gv.solver.GraphSolver(32, num_sampler_per_worker=2, seed=3, pair_order=order)
        s.build(g, batch_size=500, episode_size=3)
        s.train("LINE", num_epoch=2, augmentation_step=1, log_frequency=1 << 30)
        assert s.pair_order == order
        seen[order] = [rec for _, rec in watch.batches]
        watch.close()
    assert len(seen["sampled"]) == len(seen["grouped"]) > 0
    for a, b in zip(seen["sampled"], seen["grouped"]):
        assert (np.diff(b[:, 1].astype(np.int64)) >= 0).all() and not (a == b).all()
        key = lambda x: np.sort(x[:, 1].astype(np.int64) << 32 | x[:, 0].astype(np.int64))  # noqa: E731
        assert (key(a) == key(b)).all()


def models_and_samplers(model, aug):
    g = make_graph(200, 1500, seed=2)
    s = gv.solver.GraphSolver(32, num_sampler_per_worker=2)
    s.build(g, batch_size=300, episode_size=4, num_negative=2)
    s.train(model, num_epoch=2, augmentation_step=aug, random_walk_length=8, random_walk_batch_size=5, p=0.5, q=2.0,
            positive_reuse=2)
    assert s.batch_id % 8 == 0 and s.batch_id >= s.num_batch  # episodes of 4 x reuse 2
    assert np.abs(s.context_embeddings).max() > 0
    if model != "LINE":
        assert s.shuffle_base == 1 and s.pair_order in ("sampled", "spread")  # walk-ordered pools: as they come where chains own every row, else spread over the launches
    s.node2vec_table_limit = 10  # force the O(|E|)-memory sampler
    s.train("node2vec", num_epoch=2, augmentation_step=2, random_walk_length=8, random_walk_batch_size=5, p=0.5, q=2.0)
    assert s._mode == "biased_reject" and np.abs(s.context_embeddings).max() > 0


def device_sampling():
    """Positives drawn by the device samplers (oracle-backed here): every pair trained on is a real edge (walk pair) of the
    block being trained, training learns, the positive stream continues across resume."""
    g = make_graph(200, 2000, seed=8)
    nbrs = [set() for _ in range(g.num_vertex)]
    for u, v in g.edges.tolist():
        nbrs[u].add(v)
    for model, aug, P in (("LINE", 1, 1), ("LINE", 1, 3), ("DeepWalk", 3, 1), ("node2vec", 2, 3), ("LINE", 2, 1), ("LINE", 2, 3)):
        s = gv.solver.GraphSolver(32, num_sampler_per_worker=1, device_sampling=True, seed=4)
        s.build(g, batch_size=300, episode_size=4, num_partition=P)
        part, local = hostlib.partition(g.vertex_weights, P)[:2]
        inv = {(int(p), int(l)): v for v, (p, l) in enumerate(zip(part, local))}
        session = s.session(model=model, num_epoch=2, augmentation_step=aug, random_walk_length=8, p=0.5, q=2.0,
                            log_frequency=1 << 30)
        assert s._sampler is None and session.steps == P * P
        watch = Batches()
        session.fill(0)
        visited = set()
        for step in range(session.steps):
            hp, tp = session.block(step)
            visited.add((hp, tp))
            first = len(watch.batches)
            session.stage(step, 0, step & 1)
            session.train(step, 0, step & 1)
            session.exchange(step)
            for _, rec in watch.batches[first:]:
                for t_local, h_local in rec[::17].tolist():
                    h, t = inv[(hp, h_local)], inv[(tp, t_local)]  # KeyError = a pair binned into the wrong block
                    reach = {h} | nbrs[h]
                    for _ in range(aug - 1):
                        reach |= set().union(*[nbrs[x] for x in reach])
                    assert t in reach, "pair (%d, %d) is not within %d steps" % (h, t, aug)
        watch.close()
        assert visited == {(hp, tp) for hp in range(P) for tp in range(P)}
        session.close()
        assert np.abs(s.context_embeddings).max() > 0 and s.batch_id == P * P * 4
        s.train(model, num_epoch=2, augmentation_step=aug, random_walk_length=8, p=0.5, q=2.0, resume=True)
        assert s.batch_id >= s.num_batch


def session_equals_train():
    """solver.session(): the public step-by-step form of train() produces the same tables as train() itself."""
    g = make_graph(250, 2500, seed=3)
    kw = dict(model="LINE", num_epoch=2, augmentation_step=1, log_frequency=100000)
    a = gv.solver.GraphSolver(32, num_sampler_per_worker=2, seed=5)
    a.build(g, batch_size=500, episode_size=5, num_partition=2)
    a.train(**kw)
    for resident in (False, True):
        b = gv.solver.GraphSolver(32, num_sampler_per_worker=2, seed=5)
        b.build(g, batch_size=500, episode_size=5, num_partition=2)
        session = b.session(resident_pools=resident, **kw)
        assert session.steps == 4 and sorted(session.block(i) for i in range(4)) == [(0, 0), (0, 1), (1, 0), (1, 1)]
        current = 0
        session.fill(current)
        while b.batch_id < b.num_batch:
            for step in range(session.steps):
                session.stage(step, current, step & 1)
                session.train(step, current, step & 1, 0, 3)      # a block visit in two calls
                session.train(step, current, step & 1, 3, 2)
                session.exchange(step)
            current ^= 1
            session.fill(current)
        assert session.loss() > 0
        session.close()
        assert a.batch_id == b.batch_id
        assert (a.vertex_embeddings == b.vertex_embeddings).all() and (a.context_embeddings == b.context_embeddings).all()


def lists_prefetch():
    """The work lists of a visit's first chunk are built while the visit before it trains (gvx_engine.cpp prefetch_lists) under the assumption
    that the next call trains the visit stage() announced from its first batch: the tables must be those of a run that never builds ahead
    (GVX_LISTS_PREFETCH=0), the assumption must hold in the episode loop and in a session that walks the schedule as bench.py does, and a
    session that breaks it (a visit left early, a visit in pieces, logging batches that cut a chunk) must train what train() trains."""
    g = make_graph(n=2000, e=30000)
    kw = dict(model="LINE", num_epoch=3, augmentation_step=1)
    tables = {}
    for workers, partitions, log in ((1, 2, 1 << 30), (2, 4, 1 << 30), (1, 3, 7)):
        for knob in ("0", "1"):
            os.environ["GVX_LISTS_PREFETCH"] = knob
            s = gv.solver.GraphSolver(32, device_ids=[0] * workers, num_sampler_per_worker=1, seed=4, hub_rows=60)
            s.build(g, batch_size=1000, episode_size=3, num_partition=partitions)
            s.train(log_frequency=log, **kw)
            s._refresh()
            # every visit but an episode's first is announced while the visit before it trains; logging batches cut some first chunks short
            visits = s.num_batch // (3 * workers)
            episodes = -(-visits // (partitions * partitions // workers))
            assert s.lists_prefetched == (0 if knob == "0" else (visits - episodes) * workers), (workers, partitions, knob, s.lists_prefetched, visits, episodes)
            key = (workers, partitions, log)
            tables.setdefault(key, (s.vertex_embeddings.copy(), s.context_embeddings.copy()))
            assert (tables[key][0] == s.vertex_embeddings).all() and (tables[key][1] == s.context_embeddings).all(), (key, knob)
    del os.environ["GVX_LISTS_PREFETCH"]
    # a session that walks the schedule the way bench.py does: the next visit staged before this one trains; pools resident and reused
    reference = gv.solver.GraphSolver(32, num_sampler_per_worker=1, seed=4, hub_rows=60)
    reference.build(g, batch_size=1000, episode_size=4, num_partition=2)
    for walk in ("whole visits", "pieces and visits left early"):
        for knob in ("0", "1"):
            os.environ["GVX_LISTS_PREFETCH"] = knob
            b = gv.solver.GraphSolver(32, num_sampler_per_worker=1, seed=4, hub_rows=60)
            b.build(g, batch_size=1000, episode_size=4, num_partition=2)
            session = b.session(resident_pools=True, log_frequency=1 << 30, **kw)
            session.fill(0)
            session.stage(0, 0, 0)
            for visit in range(12):
                session.stage((visit + 1) % session.steps, 0, (visit + 1) & 1)
                step = visit % session.steps
                if walk == "whole visits":
                    session.train(step, 0, visit & 1, 0, 4)
                elif visit % 4 == 0:
                    session.train(step, 0, visit & 1, 0, 2)   # left after two batches: the next visit's lists were built for other ids
                elif visit % 4 == 1:
                    session.train(step, 0, visit & 1, 0, 1)   # in pieces: the second call continues the visit, the lists built ahead are dropped
                    session.train(step, 0, visit & 1, 1, 3)
                else:
                    session.train(step, 0, visit & 1, 0, 4)
                session.exchange(step)
            session.close()
            b._refresh()
            if walk == "whole visits":
                assert b.lists_prefetched == (11 if knob == "1" else 0), b.lists_prefetched
            else:
                assert b.lists_prefetched == (3 if knob == "1" else 0), b.lists_prefetched  # the whole visits that follow a whole visit
            tables.setdefault(walk, (b.vertex_embeddings.copy(), b.context_embeddings.copy()))
            assert (tables[walk][0] == b.vertex_embeddings).all() and (tables[walk][1] == b.context_embeddings).all(), (walk, knob)
    # a session closed while lists built ahead are outstanding, then train(resume): the next training refills the pools those lists were built from
    for knob in ("0", "1"):
        os.environ["GVX_LISTS_PREFETCH"] = knob
        b = gv.solver.GraphSolver(32, num_sampler_per_worker=1, seed=4, hub_rows=60)
        b.build(g, batch_size=1000, episode_size=4, num_partition=2)
        session = b.session(resident_pools=True, log_frequency=1 << 30, **kw)
        session.fill(0)
        session.stage(0, 0, 0)
        session.stage(1, 0, 1)
        session.train(0, 0, 0, 0, 4)
        session.exchange(0)
        session.close()
        b.train(resume=True, log_frequency=1 << 30, **kw)
        tables.setdefault("resumed", (b.vertex_embeddings.copy(), b.context_embeddings.copy()))
        assert (tables["resumed"][0] == b.vertex_embeddings).all() and (tables["resumed"][1] == b.context_embeddings).all(), knob
    del os.environ["GVX_LISTS_PREFETCH"]


def custom_schedule_and_optimizers():
    g = make_graph(150, 900, seed=4)
    log = Launches()
    s = gv.solver.GraphSolver(32, num_sampler_per_worker=1)
    s.build(g, optimizer=gv.optimizer.SGD(0.1, 0, lambda b, n: 0.5), batch_size=300, episode_size=2)
    log.clear()
    s.train("LINE", num_epoch=1, augmentation_step=1)
    ids, lrs = log.get()
    assert len(lrs) and np.allclose(lrs, 0.05)

    def broken(batch_id, num_batch):
        raise KeyError("schedule")
    s.build(g, optimizer=gv.optimizer.SGD(0.1, 0, broken), batch_size=300, episode_size=2)
    try:
        s.train("LINE", num_epoch=1, augmentation_step=1)
    except KeyError:
        pass
    else:
        raise AssertionError("the schedule's exception did not surface")
    for opt in (gv.optimizer.Momentum(0.01), gv.optimizer.AdaGrad(0.01), gv.optimizer.RMSprop(0.01), gv.optimizer.Adam(0.01)):
        s.build(g, optimizer=opt, batch_size=300, episode_size=2)
        s.train("LINE", num_epoch=1, augmentation_step=1)
        assert np.isfinite(s.vertex_embeddings).all() and np.abs(s.context_embeddings).max() > 0
        first = s.batch_id
        s.train("LINE", num_epoch=1, augmentation_step=1, resume=True)  # the moments travel with resume
        assert s.batch_id > first and np.isfinite(s.vertex_embeddings).all()


def workers_in_one_process(workers, partitions, model, aug, device_sampling, order):
    """Several workers in ONE process (device_ids = [0] * W; exchange by the copies carrier): the schedule interleaves the
    head groups, every block trains pairs of its own block only, every batch id is used exactly once, ids interleave over
    the workers, and the tables match a second run bit for bit."""
    g = make_graph(240, 2400, seed=6)
    nbrs = [set() for _ in range(g.num_vertex)]
    for u, v in g.edges.tolist():
        nbrs[u].add(v)
    part, local = hostlib.partition(g.vertex_weights, partitions)[:2]
    inv = {(int(p), int(l)): v for v, (p, l) in enumerate(zip(part, local))}
    W, P = workers, partitions
    tables = []
    for repeat in range(2):
        log = Launches()
        log.clear()
        s = gv.solver.GraphSolver(32, device_ids=[0] * W, num_sampler_per_worker=1, seed=9, pair_order=order,
                                  device_sampling=device_sampling)
        s.build(g, batch_size=400, episode_size=3, num_partition=P)
        assert s.num_worker == W and s.num_local_worker == W
        session = s.session(model=model, num_epoch=4, augmentation_step=aug, random_walk_length=6, random_walk_batch_size=4,
                            p=0.25, q=0.25, log_frequency=100000)
        assert s.transport == "device copies"
        blocks = [[session.block(step, w) for w in range(W)] for step in range(session.steps)]
        assert len({b for step in blocks for b in step}) == P * P                     # every block once per episode
        for step in blocks:
            assert len({h for h, _ in step}) == W and len({t for _, t in step}) == W   # orthogonal within a step
            assert len({h // W for h, _ in step}) == 1                                # one head group per step
        groups = [step[0][0] // W for step in blocks]
        if P > W:
            assert all(a != b for a, b in zip(groups, groups[1:]))  # consecutive steps never touch the same head group
        for w in range(W):  # a worker's context shards are pinned: tails w, W + w, ...
            assert {t for step in blocks for t in [step[w][1]]} == set(range(w, P, W))
        watch = Batches()
        current = 0
        session.fill(current)
        while s.batch_id < s.num_batch:
            for step in range(session.steps):
                first = len(watch.batches)
                session.stage(step, current, step & 1)
                session.train(step, current, step & 1)
                session.exchange(step)
                seen = watch.batches[first:]
                assert len(seen) == 3 * W
                for w in range(W):  # worker w's three batches of this step, in its block
                    hp, tp = blocks[step][w]
                    for batch_id, rec in seen[3 * w:3 * w + 3]:
                        assert batch_id % W == w
                        if s.pair_order == "grouped":
                            assert ascending_heads(s, rec)
                        for t_local, h_local in rec[::37].tolist():
                            h, t = inv[(hp, h_local)], inv[(tp, t_local)]  # KeyError = a pair routed to the wrong block
                            reach = nbrs[h] if aug == 1 else nbrs[h] | set().union(*[nbrs[x] for x in nbrs[h]])
                            assert t in reach, "pair (%d, %d) is not within %d steps" % (h, t, aug)
            current ^= 1
            if s.batch_id < s.num_batch:
                session.fill(current)
        watch.close()
        stats = s._exchange_stats()
        session.close()
        ids, lrs = log.get()
        assert np.sort(ids).tolist() == list(range(len(ids))) and len(ids) % (P * P * 3) == 0 and s.batch_id == len(ids)
        np.testing.assert_allclose(lrs, np.float32(0.025) * np.maximum(1 - ids / float(s.num_batch), 1e-4), rtol=1e-6)
        # one collective per schedule step: every worker sends its head shard (ceil(240 / P) rows of dim 32, fp32) to the others
        steps_run = len(ids) // (3 * W)
        shard = -(-240 // P) * 32 * 4
        assert stats == {"exchanges": steps_run, "bytes_sent_per_gpu": steps_run * shard * (W - 1)}
        assert np.abs(s.context_embeddings).max() > 0 and np.isfinite(s.vertex_embeddings).all()
        tables.append((s.vertex_embeddings.copy(), s.context_embeddings.copy()))
    assert (tables[0][0] == tables[1][0]).all() and (tables[0][1] == tables[1][1]).all()
    # train() itself walks the same episode
    t = gv.solver.GraphSolver(32, device_ids=[0] * W, num_sampler_per_worker=1, seed=9, pair_order=order,
                              device_sampling=device_sampling)
    t.build(g, batch_size=400, episode_size=3, num_partition=P)
    t.train(model, num_epoch=4, augmentation_step=aug, random_walk_length=6, random_walk_batch_size=4, p=0.25, q=0.25,
            log_frequency=100000)
    assert (t.vertex_embeddings == tables[0][0]).all() and (t.context_embeddings == tables[0][1]).all()


def walk_blocks_layout():
    """gvk_sample_walks_blocks as the CPU scenarios train on it (the host restatement; tests/test_kernel_gpu.py pins the device kernel to
    the same rule with the same arithmetic): per block and stripe the oracle's walk pairs, pair i of a walk of wavefront w in stripe
    (w + i % sb * (stripes // sb)) % stripes — and what the rule is for (DESIGN.md section 7.11 a): two consecutive pairs of one walk,
    the ones that share a row, never lie in the same part of a pool, whichever walks surround them."""
    from graphvite_amd import kernels as K
    from oracle_lib import Oracle
    oracle, lib = Oracle(), _lib.lib()
    g = gv.graph.Graph()
    edges = synthetic.power_law_edges(3000, 30000, seed=4)
    g.load(edges)
    for P, stripes, sb, L, aug in ((3, 8, 2, 12, 2), (2, 7, 3, 9, 3), (4, 16, 1, 10, 3)):
        part, local, _ = hostlib.partition(g.vertex_weights, P)
        s = hostlib.Sampler(g, part, local, P, seed=0)
        s.prepare("walk", num_thread=2)
        D = g.num_directed_edge
        nb_prob, nb_alias = [np.ascontiguousarray(a) for a in s.neighbor_tables(D)]
        edge_prob, edge_alias, edge_packed = K.alias_build(g.edge_weights)
        E, flat = np.ascontiguousarray(g.edges), np.ascontiguousarray(g.flat_offsets)  # uint64 offsets
        nb = np.zeros(D, np.dtype([("prob", np.float32), ("alias", np.uint32)]))
        nb["prob"], nb["alias"] = nb_prob, nb_alias
        local32, part32 = np.ascontiguousarray(local.astype(np.uint32)), np.ascontiguousarray(part.astype(np.int32))
        desc = _lib.WalkGraph(flat.ctypes.data, E.ctypes.data, edge_packed.ctypes.data, nb.ctypes.data, None, local32.ctypes.data,
                              g.num_vertex, D, 0, 1.0, 1.0)
        seed, first, walks = 77, (1 << 32) + 9, 1500
        per_walk = aug * L - aug * (aug - 1) // 2
        identity = np.arange(g.num_vertex, dtype=np.uint32)
        sorted_nb = np.ascontiguousarray(E[np.lexsort((E[:, 1], E[:, 0])), 1])
        want = oracle.sample_walks_device(flat, E, edge_prob, edge_alias, nb_prob, nb_alias, sorted_nb, identity, False, 1.0, 1.0,
                                          seed, first, walks * per_walk, L, aug, 1)  # {tail vertex, head vertex}
        block = part[want[:, 1]].astype(np.int64) * P + part[want[:, 0]]
        index = np.arange(len(want))
        apart = max(stripes // sb, 1)
        stripe_of = (index // per_walk // 64 + index % per_walk % sb * apart) % stripes
        per_stripe = max(np.bincount(block[stripe_of == k], minlength=P * P).max() for k in range(stripes))
        capacity = (int(per_stripe) + 3) * stripes
        capacity += -capacity % (sb * stripes)
        where = np.arange(P * P, dtype=np.uint64) * capacity
        pools = np.zeros((P * P, capacity, 2), np.uint32)
        counters = np.zeros((P * P, stripes), np.uint32)
        rc = lib.gvk_sample_walks_blocks(None, C.byref(desc), part32.ctypes.data, P, seed, first, walks, pools.ctypes.data,
                                         where.ctypes.data, counters.ctypes.data, capacity, stripes, L, aug, sb)
        assert rc == 0
        for b in range(P * P):
            for k in range(stripes):
                mine = want[(block == b) & (stripe_of == k)]
                assert counters[b, k] == len(mine)
                stored = pools[b][k * (capacity // stripes) + np.arange(len(mine))]
                expect = np.stack([local[mine[:, 0]], local[mine[:, 1]]], 1)
                assert sorted(map(tuple, stored.tolist())) == sorted(map(tuple, expect.tolist()))
        # thinned (gvk_sample_walks_blocks_thinned): every block keeps a pair with a probability of its own, decided by a hash of (walk, pair index, seed)
        from util import thinning_uniform
        accept = np.random.default_rng(P).uniform(0.2, 0.9, P * P).astype(np.float32)
        accept[0], accept[-1] = 1.0, 0.0
        keep = thinning_uniform(first + index // per_walk, index % per_walk, seed) < accept[block]
        assert abs(keep[accept[block] < 1].mean() - accept[block][accept[block] < 1].mean()) < 0.02 and keep[block == 0].all() and not keep[block == P * P - 1].any()
        pools[:], counters[:] = 0, 0
        rc = lib.gvk_sample_walks_blocks_thinned(None, C.byref(desc), part32.ctypes.data, P, seed, first, walks, pools.ctypes.data,
                                                 where.ctypes.data, counters.ctypes.data, capacity, stripes, L, aug, sb, accept.ctypes.data)
        assert rc == 0
        for b in range(P * P):
            for k in range(stripes):
                mine = want[(block == b) & (stripe_of == k) & keep]
                assert counters[b, k] == len(mine), (b, k, counters[b, k], len(mine))
                stored = pools[b][k * (capacity // stripes) + np.arange(len(mine))]
                expect = np.stack([local[mine[:, 0]], local[mine[:, 1]]], 1)
                assert sorted(map(tuple, stored.tolist())) == sorted(map(tuple, expect.tolist()))
        if sb > 1:  # pairs i and i + 1 of a walk: different parts of the pool — at least a part minus a stripe of records between them
            same_walk = index[:-1] // per_walk == index[1:] // per_walk
            gap = np.abs(stripe_of[1:] - stripe_of[:-1])[same_walk]
            gap = np.minimum(gap, stripes - gap)
            assert gap.min() >= min(apart, stripes - apart * (sb - 1)) >= 1, (gap.min(), apart)
            assert (gap.min() - 1) * (capacity // stripes) >= capacity // sb - 2 * (capacity // stripes) or stripes < 2 * sb


def streamed_partitions():
    """gpu_memory_limit below what the resident design needs: partitions travel through host memory (the reference's
    load_partition / write_back scheme); same accounting, the model learns."""
    edges = synthetic.community_edges(600, 12000, num_community=6, seed=2)
    train, (valid, test) = synthetic.link_prediction_split(edges, (100, 5, 5))
    g = gv.graph.Graph()
    g.load(train)
    n2i = g.name2id
    keep = [(n2i[str(h)], n2i[str(t)], y) for h, t, y in zip(*test) if str(h) in n2i and str(t) in n2i]
    aucs = {}
    for name, limit in (("resident", 0), ("streamed", 300 << 10)):  # the vertex table alone is 600 x 128 x 4 B = 300 KB
        s = gv.solver.GraphSolver(128, device_ids=[0, 0] if name == "streamed" else [0], num_sampler_per_worker=1, seed=1,
                                  gpu_memory_limit=limit)
        s.build(g, batch_size=500, episode_size=4)
        if name == "streamed":
            assert s.num_partition > 2 and s.gpu_memory_cost < limit
        s.train("LINE", num_epoch=100, augmentation_step=1, log_frequency=1 << 30)
        aucs[name] = link_prediction_auc(s.vertex_embeddings, s.context_embeddings, [k[0] for k in keep], [k[1] for k in keep],
                                         [k[2] for k in keep])
    print("AUC", aucs)
    assert aucs["resident"] > 0.8 and abs(aucs["streamed"] - aucs["resident"]) < 0.02
    try:
        gv.solver.GraphSolver(128, gpu_memory_limit=32 << 10).build(g, batch_size=500, episode_size=4)
    except MemoryError:
        return
    raise AssertionError("an impossible memory limit was accepted")


def auto_build_rules():
    """num_partition = auto and episode_size = auto as SolverMixin::build of the reference resolved them
    (solver.h:365-434; tests/golden/reference_solver.npz, one worker)."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "reference_solver.npz"))
    args, info = G["cfg_auto_1_args"], G["cfg_auto_1_info"]
    g = gv.graph.Graph()
    g.load(G["edges"].astype(np.int64), as_undirected=bool(args[1]))
    s = gv.solver.GraphSolver(128, num_sampler_per_worker=1)
    s.build(g, batch_size=int(args[5]))
    assert (s.num_vertex, s.num_edge) == (int(info[0]), int(info[1]))
    assert s.num_partition == int(info[3]) and s.episode_size == int(info[4])


def link_prediction_pipeline(tmp):
    edges = synthetic.power_law_edges(400, 6000, seed=5)
    train, (valid, test) = synthetic.link_prediction_split(edges, (100, 5, 5), seed=1024)
    assert len(train) + (valid[2] == 1).sum() + (test[2] == 1).sum() == len(edges)
    assert (test[2] == 1).sum() == (test[2] == 0).sum()
    app = gv.application.GraphApplication(dim=32)
    app.load(edge_list=train)
    app.build(batch_size=1000, episode_size=10)
    app.train(model="LINE", num_epoch=60, augmentation_step=1, log_frequency=100000)
    H, T, Y = test
    result = app.evaluate("link prediction", H=[str(h) for h in H], T=[str(t) for t in T], Y=Y.tolist(),
                          filter_H=[str(h) for h in train[:, 0]], filter_T=[str(t) for t in train[:, 1]])
    # the same number from the numpy restatement of the reference's AUC (application.py:433-449)
    n2i = app.graph.name2id
    in_train = {(n2i[str(h)], n2i[str(t)]) for h, t in train}
    keep = [(n2i[str(h)], n2i[str(t)], y) for h, t, y in zip(H, T, Y) if str(h) in n2i and str(t) in n2i]
    keep = [k for k in keep if (k[0], k[1]) not in in_train]  # filter_H / filter_T drop pairs seen in training
    want = link_prediction_auc(app.solver.vertex_embeddings, app.solver.context_embeddings, [k[0] for k in keep],
                               [k[1] for k in keep], [k[2] for k in keep])
    assert abs(result["AUC"] - want) < 1e-9 and result["AUC"] > 0.6
    pairs = np.array([[1, 2], [3, 4], [5, 5]])  # predict takes (v, c) pairs in global ids and returns dot products
    got = app.solver.predict(pairs)
    want = np.einsum("ij,ij->i", app.solver.vertex_embeddings[pairs[:, 0]], app.solver.context_embeddings[pairs[:, 1]])
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
    for bad in (np.zeros((3, 3), np.int64), np.array([[0, 400]])):
        try:
            app.solver.predict(bad)
        except ValueError:
            continue
        raise AssertionError("predict accepted %r" % (bad,))
    # save / load round trip maps nodes by name
    path = os.path.join(tmp, "model.pkl")
    app.save_model(path)
    saved = pickle.load(open(path, "rb"))
    assert saved["solver"]["vertex_embeddings"].shape == app.solver.vertex_embeddings.shape
    old = app.solver.vertex_embeddings.copy()
    app.solver.vertex_embeddings[:] = 0
    app.load_model(path)
    assert (app.solver.vertex_embeddings == old).all()
    try:
        app.evaluate("node clustering")
    except ValueError:
        pass
    else:
        raise AssertionError("an unknown task was accepted")
    # the file has the reference's layout (application.py:145-187): attribute access all the way down, the class the
    # reference pickles (easydict.EasyDict), and with save_hyperparameter its key set including solver.optimizer
    assert type(saved).__module__ == "easydict" and type(saved).__name__ == "EasyDict"
    assert saved.graph.name2id["%d" % train[0, 0]] == app.graph.name2id["%d" % train[0, 0]]
    assert saved.solver.vertex_embeddings is saved["solver"]["vertex_embeddings"]
    app.save_model(path, save_hyperparameter=True)
    full = pickle.load(open(path, "rb"))
    assert full.solver.optimizer.type == "SGD" and full.solver.optimizer.schedule == "linear"
    assert abs(full.solver.optimizer.lr - 0.025) < 1e-7 and full.solver.num_negative == 1
    assert full.graph.num_vertex == app.graph.num_vertex and full.solver.model == "LINE"
    assert full.solver.batch_size == 1000 and full.solver.random_walk_batch_size == 100
    # what the reference's load_model does with such a file (application.py:131-142, 288-291): attribute access only
    mapping = [full.graph.name2id[name] for name in app.graph.id2name]
    assert (full.solver.vertex_embeddings[mapping] == old).all()
    # and a file the way the reference writes it — object attributes gathered into nested EasyDicts, the name map an
    # EasyDict too — or as older versions of this package wrote it (plain dicts) loads here
    from graphvite_amd.application.application import easy_dict_class
    EasyDict = easy_dict_class()
    theirs = EasyDict()
    theirs.graph = EasyDict()
    theirs.graph["name2id"] = dict(app.graph.name2id)
    theirs.graph["id2name"] = list(app.graph.id2name)
    theirs.solver = EasyDict(vertex_embeddings=old * 2, context_embeddings=np.array(app.solver.context_embeddings))
    for record in (theirs, {"graph": dict(theirs.graph), "solver": dict(theirs.solver)}):
        with open(path, "wb") as fout:
            pickle.dump(record, fout, protocol=pickle.HIGHEST_PROTOCOL)
        app.solver.vertex_embeddings[:] = 0
        app.load_model(path)
        assert (app.solver.vertex_embeddings == old * 2).all()


def node_classification_and_cli(tmp):
    """The "next" rows around the path: node classification, `run config.yaml`, word2vec-format embeddings."""
    import argparse
    import yaml
    from graphvite_amd import cmd
    edges = synthetic.community_edges(300, 6000, num_community=3, seed=2)
    graph_file = os.path.join(tmp, "graph.txt")
    np.savetxt(graph_file, edges, fmt="%d")
    label_file = os.path.join(tmp, "label.txt")
    with open(label_file, "w") as f:
        for i in range(300):
            f.write("%d\tc%d\n" % (i, i // 100))
    config = {"application": "graph", "resource": {"dim": 32}, "format": {"delimiters": " \t\r\n", "comment": "#"},
              "graph": {"file_name": graph_file, "as_undirected": True},
              "build": {"optimizer": {"type": "SGD", "lr": 0.025, "weight_decay": 0.005}, "num_partition": "auto",
                        "num_negative": 1, "batch_size": 1000, "episode_size": 10},
              "train": {"model": "LINE", "num_epoch": 150, "augmentation_step": 1, "log_frequency": 100000},
              "evaluate": [{"task": "node classification", "file_name": label_file, "portions": [0.2], "times": 1}],
              "save": {"file_name": os.path.join(tmp, "model.pkl")}}
    config_file = os.path.join(tmp, "config.yaml")
    open(config_file, "w").write(yaml.safe_dump(config))
    app = cmd.run_main(argparse.Namespace(config=config_file, gpu=None, cpu=None, eval=True))
    assert app.solver.num_partition == 1 and app.solver.optimizer.type == "SGD" and os.path.exists(os.path.join(tmp, "model.pkl"))
    result = app.node_classification(file_name=label_file, portions=(0.2,), times=2)
    assert result["micro-F1@20%"] > 0.9 and result["macro-F1@20%"] > 0.9  # three planted communities
    out = os.path.join(tmp, "emb.bin")  # word2vec-style embedding file
    app.solver.save_embeddings(out)
    data = open(out, "rb").read()
    header, rest = data.split(b"\n", 1)
    assert header == b"300 32"
    name0 = app.graph.id2name[0].encode()
    assert rest.startswith(name0 + b" ")
    first = np.frombuffer(rest[len(name0) + 1:len(name0) + 1 + 32 * 4], np.float32)
    assert (first == app.solver.vertex_embeddings[0]).all()
    # `baseline` / `list` (cmd.py:193-260): a configuration directory laid out as the reference's, a dataset placeholder that
    # resolves to the file the reference's downloader would have left (dataset.py:183) — and to an error when it is not there
    configs, datasets = os.path.join(tmp, "config"), os.path.join(tmp, "dataset")
    os.makedirs(os.path.join(configs, "graph"))
    os.makedirs(os.path.join(configs, "template"))
    os.makedirs(os.path.join(datasets, "toy"))
    np.savetxt(os.path.join(datasets, "toy", "toy_train.txt"), edges, fmt="%d")
    baseline = dict(config, graph={"file_name": "<toy.train>", "as_undirected": True}, evaluate=[], save={"file_name": os.path.join(tmp, "b.pkl")})
    baseline["train"] = dict(config["train"], num_epoch=1000)
    for name in ("line_toy.yaml", "deepwalk_toy.yaml"):
        open(os.path.join(configs, "graph", name), "w").write(yaml.safe_dump(baseline))
    assert cmd.main(["baseline", "line", "toy", "--config-path", configs, "--dataset-path", datasets, "--epoch", "2", "--no-eval"]) == 0
    assert os.path.exists(os.path.join(tmp, "b.pkl"))
    assert cmd.list_main(argparse.Namespace(config_path=configs)) == 2
    for keywords, message in ((["toy"], "Ambiguous"), (["node2vec"], "Can't find")):
        try:
            cmd.find_baseline(keywords, configs)
        except ValueError as error:
            assert message in str(error)
            continue
        raise AssertionError("baseline lookup accepted %s" % keywords)
    bad = os.path.join(tmp, "bad.yaml")
    open(bad, "w").write("graph:\n  file_name: <blogcatalog.train>\n")
    try:
        cmd.load_config(bad, datasets)  # a dataset that is not there: nothing is downloaded
    except ValueError:
        return
    raise AssertionError("a dataset placeholder was accepted")


def word_graph_application(tmp):
    """WordGraphApplication (application.py:536-573): corpus -> co-occurrence graph -> the same GraphSolver path."""
    rng = np.random.default_rng(0)
    topics = [["cat", "dog", "pet", "vet", "fur"], ["gpu", "hbm", "wave", "lane", "simd"]]
    lines = [" ".join(rng.choice(topics[i % 2], 12)) for i in range(400)]
    path = os.path.join(tmp, "corpus.txt")
    open(path, "w").write("\n".join(lines) + "\n")
    app = gv.application.Application("word graph", dim=32)
    app.load(file_name=path, window=3, min_count=5)
    app.build(batch_size=200, episode_size=5)
    app.train(model="LINE", num_epoch=300, augmentation_step=1, log_frequency=1 << 30)
    assert isinstance(app.graph, gv.graph.WordGraph) and app.graph.num_vertex == 10
    v, c = app.solver.vertex_embeddings, app.solver.context_embeddings
    score = v @ c.T
    ids = [[app.graph.name2id[w] for w in topic] for topic in topics]
    inside = np.mean([score[np.ix_(t, t)].mean() for t in ids])
    across = np.mean([score[np.ix_(ids[0], ids[1])].mean(), score[np.ix_(ids[1], ids[0])].mean()])
    assert inside > across + 0.5  # words of a topic co-occur, words of different topics never do


# ---- one process per worker, over gloo (the engine's transport hook) ------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank(rank, world, port, out_dir, model, aug, partitions, order, device_sampling, graph_args, train_kw, dim):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gv.init_logging(logging.ERROR)
        kind, args, kw = graph_args
        edges = getattr(synthetic, kind)(*args, **kw)
        test = None
        if kind == "community_edges":
            edges, (valid, test) = synthetic.link_prediction_split(edges, (100, 3, 3))
        g = gv.graph.Graph()
        g.load(edges)
        log = Launches()
        log.clear()
        s = gv.solver.GraphSolver(dim, num_sampler_per_worker=2, seed=train_kw.pop("seed", 9), device_sampling=device_sampling,
                                  pair_order=gv.auto if order == "auto" else order, **train_kw.pop("solver", {}))
        s.build(g, num_partition=partitions, **train_kw.pop("build"))
        assert s.num_worker == world and s.num_local_worker == 1 and s.rank == rank
        assert s.num_partition == (partitions or world)
        part, local = hostlib.partition(g.vertex_weights, s.num_partition)[:2]
        inv = {(int(p), int(l)): v for v, (p, l) in enumerate(zip(part, local))}
        nbrs = [set() for _ in range(g.num_vertex)]
        for u, v in g.edges.tolist():
            nbrs[u].add(v)
        tails = set()
        if train_kw.pop("check_pairs", True):
            # every pair this rank trains on must be a real (walk) pair of the graph that lives in the block being trained
            session = s.session(model=model, augmentation_step=aug, **train_kw)
            watch = Batches()
            current = 0
            session.fill(current)
            while s.batch_id < s.num_batch:
                for step in range(session.steps):
                    hp, tp = session.block(step)
                    tails.add(tp)
                    first = len(watch.batches)
                    session.stage(step, current, step & 1)
                    session.train(step, current, step & 1)
                    session.exchange(step)
                    for _, rec in watch.batches[first:]:
                        if s.pair_order == "grouped":  # every (part of a) batch arrives in ascending head-row order
                            assert ascending_heads(s, rec)
                        for t_local, h_local in rec[::37].tolist():
                            h, t = inv[(hp, h_local)], inv[(tp, t_local)]        # KeyError = a pair routed to the wrong block
                            reach = nbrs[h] if aug == 1 else nbrs[h] | set().union(*[nbrs[x] for x in nbrs[h]])
                            assert t in reach, "pair (%d, %d) is not within %d steps" % (h, t, aug)
                current ^= 1
                if s.batch_id < s.num_batch:
                    session.fill(current)
            watch.close()
            session.close()
        else:
            s.train(model=model, augmentation_step=aug, **train_kw)
        assert s.transport == "caller-supplied transport"
        ids, lrs = log.get()
        out = dict(v=s.vertex_embeddings, c=s.context_embeddings, ids=ids, lrs=lrs, batch_id=s.batch_id, num_batch=s.num_batch,
                   tails=np.array(sorted(tails)))
        if test is not None:
            n2i = g.name2id
            keep = [(n2i[str(h)], n2i[str(t)], y) for h, t, y in zip(*test) if str(h) in n2i and str(t) in n2i]
            out["auc"] = link_prediction_auc(s.vertex_embeddings, s.context_embeddings, [k[0] for k in keep], [k[1] for k in keep],
                                             [k[2] for k in keep])
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    finally:
        dist.destroy_process_group()


def _spawn(world, out_dir, model, aug, partitions=0, order="sampled", device_sampling=False, graph_args=None, train_kw=None, dim=32):
    import torch.multiprocessing as mp
    graph_args = graph_args or ("power_law_edges", (240, 2400), dict(seed=6))
    train_kw = train_kw or dict(build=dict(batch_size=400, episode_size=3), num_epoch=4, random_walk_length=6,
                                random_walk_batch_size=4, p=0.25, q=0.25, log_frequency=100000)
    mp.spawn(_rank, args=(world, _free_port(), out_dir, model, aug, partitions, order, device_sampling, graph_args, train_kw, dim),
             nprocs=world, join=True)
    return [np.load(os.path.join(out_dir, "rank%d.npz" % i)) for i in range(world)]


def processes_over_gloo(tmp, world, model, aug, partitions, order, device_sampling):
    """One process per worker (gvx_solver_create_distributed), the engine's collectives carried by gloo through its
    transport hook: the same engine code path RCCL carries on GPUs."""
    r = _spawn(world, tmp, model, aug, partitions, order, device_sampling)
    P = partitions or world
    for other in r[1:]:  # after write-back every process holds the same, complete tables
        assert (r[0]["v"] == other["v"]).all() and (r[0]["c"] == other["c"]).all()
    assert np.abs(r[0]["c"]).max() > 0 and np.isfinite(r[0]["v"]).all()
    assert [x["tails"].tolist() for x in r] == [list(range(w, P, world)) for w in range(world)]  # context shards pinned per worker
    # the workers share one batch counter: ids interleave, every id exactly once, whole episodes
    ids = np.sort(np.concatenate([x["ids"] for x in r]))
    assert (ids == np.arange(len(ids))).all() and len(ids) % (P * P * 3) == 0
    for w, x in enumerate(r):
        assert (x["ids"] % world == w).all() and int(x["batch_id"]) == len(ids) >= int(x["num_batch"])
        np.testing.assert_allclose(x["lrs"], np.float32(0.025) * np.maximum(1 - x["ids"] / float(x["num_batch"]), 1e-4), rtol=1e-6)


def hub_rows_over_gloo(tmp):
    """Two processes, four partitions, hub rows requested: every rank builds the work lists of its own blocks; with the
    host build's sequential kernels the tables are those of the plain run, and both ranks end with the same tables."""
    kw = lambda **solver: dict(seed=9, solver=solver, check_pairs=False, build=dict(batch_size=400, episode_size=3), num_epoch=4,  # noqa: E731
                               log_frequency=100000)
    plain = _spawn(2, tmp, "LINE", 1, 4, "sampled", False, None, kw())
    for solver in (dict(hub_rows=20), dict(fidelity="reference")):
        hub = _spawn(2, tmp, "LINE", 1, 4, "sampled", False, None, kw(**solver))
        assert (hub[0]["v"] == hub[1]["v"]).all() and (hub[0]["c"] == hub[1]["c"]).all()
        assert (hub[0]["v"] == plain[0]["v"]).all() and (hub[0]["c"] == plain[0]["c"]).all()


def learning_quality_over_gloo(tmp):
    """Learning quality of the multi-process data path against the reference's OWN training loop on the same graph
    (tests/golden/reference_solver.npz `train_small_*`: 4000 nodes, LINE, 150 epochs; means over the seeds stored there):
    two processes / four partitions — context shards pinned per process, slot claims, asynchronous all-gather, head groups
    interleaved — within +-0.002 of the reference's loop at four partitions.  A stale or misplaced shard costs far more."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "reference_solver.npz"))
    n, e, communities, graph_seed, batch, episode, epochs = [int(x) for x in G["train_small_args"]]
    reference = np.asarray(G["train_small_w1_p4_aucs"], np.float64)
    aucs = []
    for seed in (3, 4, 5, 6, 7):
        r = _spawn(2, tmp, "LINE", 1, 4, "auto", False, ("community_edges", (n, e), dict(num_community=communities, seed=graph_seed)),
                   dict(seed=seed, check_pairs=False, build=dict(batch_size=batch, episode_size=episode), num_epoch=epochs,
                        log_frequency=1 << 30), dim=128)
        assert (r[0]["v"] == r[1]["v"]).all()
        aucs.append(float(r[0]["auc"]))
    print("2 processes / 4 partitions: AUC %s (mean %.6f) | reference loop, 4 partitions: %s (mean %.6f)"
          % (" ".join("%.6f" % a for a in aucs), np.mean(aucs), " ".join("%.6f" % a for a in reference), reference.mean()))
    assert abs(np.mean(aucs) - reference.mean()) <= 0.002


if __name__ == "__main__":
    assert os.environ.get("GVK_LIBRARY") and _lib.lib().gvh_is_host_build(), "run with GVK_LIBRARY = the host build"
    gv.init_logging(logging.ERROR)
    name = sys.argv[1]
    arguments = json.loads(sys.argv[2]) if len(sys.argv) > 2 else []
    globals()[name](*arguments)
    print("scenario %s%s: ok" % (name, tuple(arguments)))
