"""ctypes bindings of the CPU oracle (oracle/liboracle.so) and of the host build of the
reference's own arithmetic (oracle/_ref/libgvref.so).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by graphvite_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

SGD, MOMENTUM, ADAGRAD, RMSPROP, ADAM = range(5)

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build_oracle():
    """Compile oracle/ (and oracle/_ref when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle(object):
    def __init__(self):
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        self.lib = lib = C.CDLL(path)
        lib.gvo_sigmoid.restype = C.c_float
        lib.gvo_sigmoid.argtypes = [C.c_float]
        lib.gvo_lr.restype = C.c_float
        lib.gvo_lr.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int]
        lib.gvo_train.restype = C.c_int
        lib.gvo_train.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6 + [_u32p, _u32p, _f32p, C.c_int, C.c_int,
                                                                         C.c_float, C.c_float, C.c_float, _f32p]
        lib.gvo_hot_lists.restype = C.c_size_t
        lib.gvo_hot_lists.argtypes = [_u32p, _u32p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, _u32p, _u32p]
        lib.gvo_hot_unit_chains.restype = C.c_int
        lib.gvo_hot_unit_chains.argtypes = [C.c_int, _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, _u32p, _u32p,
                                            C.c_uint32, C.c_uint32, C.c_int]
        lib.gvo_train_pairs_hot.restype = C.c_int
        lib.gvo_train_pairs_hot.argtypes = [C.c_int, _f32p, _f32p, _u32p, _u32p, _f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                            C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.gvo_set_hub_snapshot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        lib.gvo_train_hot_moments.restype = C.c_int
        lib.gvo_train_hot_moments.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6 + [_u32p, _u32p, _f32p, C.c_int, C.c_int, C.c_float,
                                                                                     C.c_float, C.c_float, _f32p, C.c_uint32, C.c_uint32,
                                                                                     _u32p, _u32p]
        lib.gvo_train_hot.restype = C.c_int
        lib.gvo_train_hot.argtypes = [C.c_int, _f32p, _f32p, _u32p, _u32p, _f32p, C.c_int, C.c_int, C.c_float, C.c_float,
                                      C.c_float, C.c_uint32, C.c_uint32, _u32p, _u32p, C.c_uint32, C.c_uint32, C.c_int]
        lib.gvo_predict.restype = None
        lib.gvo_predict.argtypes = [C.c_int, _f32p, _f32p, _u32p, _f32p, C.c_int]
        lib.gvo_alias_build.restype = C.c_int
        lib.gvo_alias_build.argtypes = [_f32p, C.c_size_t, _f32p, C.c_void_p, C.c_int]
        lib.gvo_alias_sample.restype = C.c_uint64
        lib.gvo_alias_sample.argtypes = [_f32p, C.c_void_p, C.c_int, C.c_uint64, C.c_double, C.c_double]
        lib.gvo_alias_sample_gpu.restype = C.c_uint32
        lib.gvo_alias_sample_gpu.argtypes = [_f32p, _u32p, C.c_uint32, C.c_double, C.c_double]
        lib.gvo_philox4x32.restype = None
        lib.gvo_philox4x32.argtypes = [_u32p, _u32p, _u32p]
        lib.gvo_negative_draw.restype = C.c_uint32
        lib.gvo_negative_draw.argtypes = [_f32p, _u32p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.gvo_negative_draw_batch.restype = None
        lib.gvo_negative_draw_batch.argtypes = [_f32p, _u32p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.c_int, _u32p]
        lib.gvo_class_table_build.restype = C.c_uint32
        lib.gvo_class_table_build.argtypes = [_f32p, C.c_size_t, _u32p, _u32p, _f32p, _u32p]
        lib.gvo_negative_draw_class_batch.restype = None
        lib.gvo_negative_draw_class_batch.argtypes = [_u32p, _u32p, _f32p, _u32p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int,
                                                       C.c_int, _u32p]
        lib.gvo_sample_pairs.restype = None
        lib.gvo_sample_pairs.argtypes = [_f32p, _u32p, _u32p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_size_t, _u32p]
        lib.gvo_sample_walks_device.restype = C.c_int
        lib.gvo_sample_walks_device.argtypes = [_u64p, _u32p, _f32p, _u32p, C.c_uint32, _f32p, _u32p, C.c_void_p, _u32p,
                                                C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_uint64, _u32p,
                                                C.c_size_t, C.c_int, C.c_int, C.c_int]
        lib.gvo_host_uniforms.restype = None
        lib.gvo_host_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_size_t, _f64p]
        lib.gvo_partition.restype = C.c_int
        lib.gvo_partition.argtypes = [_f32p, C.c_uint32, C.c_int, _i32p, _u32p, _u32p]
        lib.gvo_schedule.restype = C.c_int
        lib.gvo_schedule.argtypes = [C.c_int, C.c_int, _i32p]
        lib.gvo_negative_weights.restype = None
        lib.gvo_negative_weights.argtypes = [_f32p, _u32p, C.c_uint32, C.c_float, _f32p]
        lib.gvo_sample_edges.restype = C.c_size_t
        lib.gvo_sample_edges.argtypes = [_u32p, _f32p, _u64p, C.c_uint64, _i32p, _u32p, C.c_int,
                                         C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, _f64p, C.c_size_t]
        lib.gvo_sample_walks.restype = C.c_size_t
        lib.gvo_sample_walks.argtypes = [C.c_int, _u32p, _f32p, _u64p, C.c_uint64, _u64p, _f32p, _u32p,
                                         C.c_void_p, _i32p, _u32p, C.c_int, C.POINTER(C.c_void_p), C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_float, C.c_float, _f64p, C.c_size_t]
        lib.gvo_sample_walks_reference_order.restype = C.c_size_t
        lib.gvo_sample_walks_reference_order.argtypes = [C.c_int, _u32p, _f32p, _u64p, C.c_uint64, _u64p, _f32p, _u32p,
                                                         C.c_void_p, _i32p, _u32p, C.c_int, C.POINTER(C.c_void_p), C.c_int,
                                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f64p,
                                                         C.c_size_t]
        lib.gvo_edge_edge_weights.restype = None
        lib.gvo_edge_edge_weights.argtypes = [_u32p, _f32p, _u64p, C.c_uint64, C.c_float, C.c_float, _f32p]

    # -- arithmetic ---------------------------------------------------------------------
    def sigmoid(self, x):
        return self.lib.gvo_sigmoid(x)

    def lr(self, init_lr, linear, batch_id, num_batch):
        return self.lib.gvo_lr(init_lr, int(linear), batch_id, num_batch)

    def train(self, vertex, context, batch, negatives, lr, wd, negative_weight, optimizer=SGD, moments=None,
              hp=(0, 0, 0)):
        """In place on vertex/context (and moments = [vm1, cm1, vm2, cm2]). Returns loss[B]."""
        B = batch.shape[0]
        k = negatives.size // B if B else 0
        loss = np.zeros(B, np.float32)
        m = list(moments or []) + [None] * 4
        hp = np.asarray(hp, np.float32)
        rc = self.lib.gvo_train(vertex.shape[1], optimizer, _opt(vertex), _opt(context), _opt(m[0]), _opt(m[1]),
                                _opt(m[2]), _opt(m[3]), batch.reshape(-1), negatives.reshape(-1), loss, B, k, lr,
                                wd, negative_weight, hp)
        assert rc == 0
        return loss

    def hot_lists(self, batch, negatives, hot_vertex, hot_context):
        """Work lists of the hub rows' chains in sample order: (chain_start [kv + kc + 1], entries [n])."""
        B = batch.shape[0]
        k = negatives.size // B if B else 0
        start = np.zeros(hot_vertex + hot_context + 1, np.uint32)
        entries = np.zeros(max(2 * (k + 1) * B, 1), np.uint32)
        n = self.lib.gvo_hot_lists(np.ascontiguousarray(batch.reshape(-1)), np.ascontiguousarray(negatives.reshape(-1)), B, k,
                                   hot_vertex, hot_context, start, entries)
        return start, entries[:n].copy()

    def train_hot(self, vertex, context, batch, negatives, lr, wd, negative_weight, hot_vertex, hot_context, chain_start,
                  entries, cap, max_tasks=0, lerp=False, long_task=0, round_steps=0):
        """One unit in the product's serialized hub-chain form (gvk_train_episode_hot(serialized=1)): the chains of both
        families from the unit's start state, then its pairs (`lerp`: hub rows read along the chains' way); in place.
        A chain of more than cap entries: tasks of cap entries side by side, at most max_tasks of them (the tasks one workgroup of
        the product trains: 256 / lanes per pair; 0 = no limit), in rounds of round_steps entries per task (the product with form
        GVK_HOT_ROUNDS: 4; 0 = one round) — or, long_task > 0 (experiments), tasks of long_task entries."""
        self.lib.gvo_set_long_task(int(long_task))
        self.lib.gvo_set_round_steps(int(round_steps))  # gvk.h GVK_HOT_ROUNDS: rounds of GVK_HOT_ROUND_STEPS = 4 entries per task; 0 = one round
        B = batch.shape[0]
        k = negatives.size // B if B else 0
        loss = np.zeros(max(B, 1), np.float32)
        entries = np.ascontiguousarray(entries, np.uint32)
        if entries.size == 0:
            entries = np.zeros(1, np.uint32)
        rc = self.lib.gvo_train_hot(vertex.shape[1], vertex, context, np.ascontiguousarray(batch.reshape(-1)),
                                    np.ascontiguousarray(negatives.reshape(-1)), loss, B, k, lr, wd, negative_weight, hot_vertex,
                                    hot_context, np.ascontiguousarray(chain_start, np.uint32), entries, cap, max_tasks, int(lerp))
        assert rc == 0
        return loss[:B]

    def train_hot_group(self, vertex, context, batch, negatives, lr, wd, negative_weight, hot_vertex, hot_context, chain_starts, entries,
                        cap, max_tasks=0, round_steps=0):
        """One GROUP of units in the serialized form of the grouped executor (gvk_train_episode_ahead with group > 1): the chains of the
        group's units one unit after the other — every chain from its own row as the unit before left it, its hub PARTNERS as the group
        found them —, then the pairs of every unit with the hub rows as that unit's chains left them.  batch / negatives: the group's
        samples, equal units; chain_starts / entries: per unit.  In place."""
        self.lib.gvo_set_long_task(0)
        self.lib.gvo_set_round_steps(int(round_steps))
        units = len(chain_starts)
        n = batch.shape[0] // units
        k = negatives.size // batch.shape[0]
        dim = vertex.shape[1]
        snap_v, snap_c = vertex[:hot_vertex].copy(), context[:hot_context].copy()
        self.lib.gvo_set_hub_snapshot(snap_v.ctypes.data, hot_vertex, snap_c.ctypes.data, hot_context)
        after = []
        try:
            for u in range(units):
                e = np.ascontiguousarray(entries[u], np.uint32)
                rc = self.lib.gvo_hot_unit_chains(dim, vertex, context, lr, wd, negative_weight, hot_vertex, hot_context,
                                                  np.ascontiguousarray(chain_starts[u], np.uint32), e if e.size else np.zeros(1, np.uint32), cap, max_tasks, k)
                assert rc == 0
                after.append((vertex[:hot_vertex].copy(), context[:hot_context].copy()))
        finally:
            self.lib.gvo_set_hub_snapshot(None, 0, None, 0)
        loss = np.zeros(batch.shape[0], np.float32)
        for u in range(units):  # the pairs of unit u read the hub rows as ITS chains left them
            vertex[:hot_vertex], context[:hot_context] = after[u]
            part, neg = np.ascontiguousarray(batch[u * n:(u + 1) * n].reshape(-1)), np.ascontiguousarray(negatives[u * n:(u + 1) * n].reshape(-1))
            out = np.zeros(n, np.float32)
            rc = self.lib.gvo_train_pairs_hot(dim, vertex, context, part, neg, out, n, k, lr, wd, negative_weight, hot_vertex, hot_context, None, None)
            assert rc == 0
            loss[u * n:(u + 1) * n] = out
        vertex[:hot_vertex], context[:hot_context] = after[-1]
        return loss

    def train_hot_moments(self, vertex, context, batch, negatives, lr, wd, negative_weight, optimizer, moments, hp, hot_vertex,
                          hot_context, chain_start, entries):
        """One unit in the serialized form of the moment optimizers' chains (gvo_train_hot_moments): every chain one sequential task on
        its row and moment rows, then the pairs (hub rows and their moment rows read, not written); in place on the tables and on
        moments = [vm1, cm1, vm2, cm2]."""
        B = batch.shape[0]
        k = negatives.size // B if B else 0
        loss = np.zeros(max(B, 1), np.float32)
        m = list(moments) + [None] * 4
        entries = np.ascontiguousarray(entries, np.uint32)
        if entries.size == 0:
            entries = np.zeros(1, np.uint32)
        rc = self.lib.gvo_train_hot_moments(vertex.shape[1], optimizer, _opt(vertex), _opt(context), _opt(m[0]), _opt(m[1]), _opt(m[2]),
                                            _opt(m[3]), np.ascontiguousarray(batch.reshape(-1)), np.ascontiguousarray(negatives.reshape(-1)),
                                            loss, B, k, lr, wd, negative_weight, np.asarray(hp, np.float32), hot_vertex, hot_context,
                                            np.ascontiguousarray(chain_start, np.uint32), entries)
        assert rc == 0
        return loss[:B]

    def predict(self, vertex, context, batch):
        out = np.zeros(batch.shape[0], np.float32)
        self.lib.gvo_predict(vertex.shape[1], vertex, context, batch.reshape(-1), out, batch.shape[0])
        return out

    # -- alias ---------------------------------------------------------------------------
    def alias_build(self, w, index_bytes=4):
        w = np.ascontiguousarray(w, np.float32)
        prob = np.zeros(w.size, np.float32)
        alias = np.zeros(w.size, np.uint64 if index_bytes == 8 else np.uint32)
        rc = self.lib.gvo_alias_build(w, w.size, prob, alias.ctypes.data_as(C.c_void_p), index_bytes)
        assert rc == 0
        return prob, alias

    def alias_sample(self, prob, alias, r1, r2):
        return self.lib.gvo_alias_sample(prob, alias.ctypes.data_as(C.c_void_p), alias.itemsize, prob.size, r1, r2)

    def alias_sample_gpu(self, prob, alias, r1, r2):
        return self.lib.gvo_alias_sample_gpu(prob, alias, prob.size, r1, r2)

    # -- rng contract ----------------------------------------------------------------------
    def philox(self, ctr, key):
        out = np.zeros(4, np.uint32)
        self.lib.gvo_philox4x32(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
        return out

    def negative_draw(self, prob, alias, seed, batch_id, sample_id, j):
        return self.lib.gvo_negative_draw(prob, alias, prob.size, seed, batch_id, sample_id, j)

    def negatives(self, prob, alias, seed, batch_id, batch_size, k):
        out = np.zeros((batch_size, k), np.uint32)
        self.lib.gvo_negative_draw_batch(prob, alias, prob.size, seed, batch_id, batch_size, k, out.reshape(-1))
        return out

    def class_table(self, weights):
        """(first, count, prob, alias) of the weight classes of a partition (gvk_class_table_build restated)."""
        w = np.ascontiguousarray(weights, np.float32)
        first, count = np.zeros(w.size, np.uint32), np.zeros(w.size, np.uint32)
        prob, alias = np.zeros(w.size, np.float32), np.zeros(w.size, np.uint32)
        n = self.lib.gvo_class_table_build(w, w.size, first, count, prob, alias)
        return first[:n].copy(), count[:n].copy(), prob[:n].copy(), alias[:n].copy()

    def negatives_by_class(self, classes, seed, batch_id, batch_size, k):
        first, count, prob, alias = (np.ascontiguousarray(a) for a in classes)
        out = np.zeros((batch_size, k), np.uint32)
        self.lib.gvo_negative_draw_class_batch(first, count, prob, alias, first.size, seed, batch_id, batch_size, k,
                                               out.reshape(-1))
        return out

    def sample_pairs(self, prob, alias, block_pairs, seed, first, n):
        out = np.zeros((n, 2), np.uint32)
        self.lib.gvo_sample_pairs(prob, alias, np.ascontiguousarray(block_pairs, np.uint32).reshape(-1), prob.size,
                                  seed, first, n, out.reshape(-1))
        return out

    def sample_walks_device(self, flat, edges_uv, edge_prob, edge_alias, nb_prob, nb_alias, sorted_nb, local, biased,
                            p, q, seed, first_walk, pool_pairs, L, aug, shuffle_base):
        pool = np.zeros((pool_pairs, 2), np.uint32)
        snb = None if sorted_nb is None else sorted_nb.ctypes.data_as(C.c_void_p)
        rc = self.lib.gvo_sample_walks_device(flat, np.ascontiguousarray(edges_uv).reshape(-1), edge_prob, edge_alias,
                                              edge_prob.size, nb_prob, nb_alias, snb, local, int(biased), p, q, seed,
                                              first_walk, pool.reshape(-1), pool_pairs, L, aug, shuffle_base)
        assert rc == 0
        return pool

    def host_uniforms(self, seed, stream, first, n):
        out = np.zeros(n, np.float64)
        self.lib.gvo_host_uniforms(seed, stream, first, n, out)
        return out

    # -- partition / schedule ----------------------------------------------------------------
    def partition(self, weights, P):
        weights = np.ascontiguousarray(weights, np.float32)
        part = np.zeros(weights.size, np.int32)
        local = np.zeros(weights.size, np.uint32)
        sizes = np.zeros(P, np.uint32)
        assert self.lib.gvo_partition(weights, weights.size, P, part, local, sizes) == 0
        return part, local, sizes

    def schedule(self, P, W):
        out = np.zeros(max(1, (P // max(W, 1)) ** 2 * W * W) * 2 + 2, np.int32)
        steps = self.lib.gvo_schedule(P, W, out)
        nw = 1 if P == 1 else W
        return out[:steps * nw * 2].reshape(steps, nw, 2)

    def negative_weights(self, vertex_weights, global_ids, exponent):
        out = np.zeros(global_ids.size, np.float32)
        self.lib.gvo_negative_weights(np.ascontiguousarray(vertex_weights, np.float32),
                                      np.ascontiguousarray(global_ids, np.uint32), global_ids.size, exponent, out)
        return out

    # -- samplers --------------------------------------------------------------------------
    @staticmethod
    def _pool_ptrs(pools):
        arr = (C.c_void_p * len(pools))()
        for i, p in enumerate(pools):
            arr[i] = p.ctypes.data
        return arr

    def sample_edges(self, edges_uv, edge_prob, edge_alias, part, local, P, pools, start, end, sample_batch_size,
                     rnd, tail_filter=-1):
        n = self.lib.gvo_sample_edges(np.ascontiguousarray(edges_uv).reshape(-1), edge_prob, edge_alias,
                                      edge_prob.size, part, local, P, self._pool_ptrs(pools), start, end,
                                      sample_batch_size, tail_filter, rnd, rnd.size)
        assert n != C.c_size_t(-1).value, "uniform stream too short"
        return n

    def sample_walks(self, biased, edges_uv, edge_prob, edge_alias, flat_offsets, nb_prob, nb_alias, ee_offsets,
                     part, local, P, pools, pool_size, start, end, walk_length, walk_batch, augmentation_step,
                     shuffle_base, rnd, tail_filter=-1, sorted_nb=None, p=1.0, q=1.0):
        eo = None if ee_offsets is None else ee_offsets.ctypes.data_as(C.c_void_p)
        n = self.lib.gvo_sample_walks(int(biased), edges_uv.reshape(-1), edge_prob, edge_alias, edge_prob.size,
                                      flat_offsets, nb_prob, nb_alias, eo, part, local, P,
                                      self._pool_ptrs(pools), pool_size, start, end, walk_length, walk_batch,
                                      augmentation_step, shuffle_base, tail_filter,
                                      None if sorted_nb is None else sorted_nb.ctypes.data_as(C.c_void_p), p, q, rnd,
                                      rnd.size)
        assert n < C.c_size_t(-2).value, "uniform stream too short / bad shuffle base"
        return n

    def sample_walks_reference_order(self, biased, edges_uv, edge_prob, edge_alias, flat_offsets, nb_prob, nb_alias,
                                     ee_offsets, part, local, P, pools, pool_size, start, end, walk_length, walk_batch,
                                     augmentation_step, shuffle_base, rnd):
        """The walk sampler consuming its uniforms walk by walk, as the reference's samplers do (graph.cuh:298-450)."""
        eo = None if ee_offsets is None else ee_offsets.ctypes.data_as(C.c_void_p)
        n = self.lib.gvo_sample_walks_reference_order(int(biased), edges_uv.reshape(-1), edge_prob, edge_alias,
                                                      edge_prob.size, flat_offsets, nb_prob, nb_alias, eo, part, local, P,
                                                      self._pool_ptrs(pools), pool_size, start, end, walk_length,
                                                      walk_batch, augmentation_step, shuffle_base, rnd, rnd.size)
        assert n < C.c_size_t(-2).value, "uniform stream too short / bad arguments"
        return n

    def edge_edge_weights(self, edges_uv, edge_weights, flat_offsets, e, p, q):
        v = int(edges_uv[e, 1])
        out = np.zeros(int(flat_offsets[v + 1] - flat_offsets[v]), np.float32)
        self.lib.gvo_edge_edge_weights(edges_uv.reshape(-1), edge_weights, flat_offsets, e, p, q, out)
        return out


class Reference(object):
    """Host build of the reference's own model / optimizer headers (oracle/ref_harness.cpp)."""

    def __init__(self, fast=False):
        name = "libgvref_fast.so" if fast else "libgvref.so"
        path = os.path.join(ORACLE_DIR, "_ref", name)
        if not os.path.exists(path):
            build_oracle()
        if not os.path.exists(path):
            raise FileNotFoundError("%s is not built and /root/reference is not present" % path)
        self.lib = lib = C.CDLL(path)
        lib.gvref_train.restype = C.c_int
        lib.gvref_train.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6 + [_u32p, _u32p, _f32p, C.c_int, C.c_int,
                                                                           C.c_float, C.c_float, C.c_float, _f32p]
        lib.gvref_train_mt.restype = C.c_int
        lib.gvref_train_mt.argtypes = [C.c_int, _f32p, _f32p, _u32p, _u32p, _f32p, C.c_int, C.c_int, C.c_float,
                                       C.c_float, C.c_float, C.c_int]
        lib.gvref_predict.restype = C.c_int
        lib.gvref_predict.argtypes = [C.c_int, _f32p, _f32p, _u32p, _f32p, C.c_int]
        lib.gvref_sigmoid.restype = C.c_float
        lib.gvref_sigmoid.argtypes = [C.c_float]
        lib.gvref_lr.restype = C.c_float
        lib.gvref_lr.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int]

        # the reference's own AliasTable (oracle/ref_alias_harness.cpp)
        alias_path = os.path.join(ORACLE_DIR, "_ref", "libgvref_alias.so")
        self.alias_lib = None
        if os.path.exists(alias_path):
            self.alias_lib = al = C.CDLL(alias_path)
            al.gvref_alias_build.restype = C.c_int
            al.gvref_alias_build.argtypes = [_f32p, C.c_uint32, _f32p, _u32p]
            al.gvref_alias_build64.restype = C.c_int
            al.gvref_alias_build64.argtypes = [_f32p, C.c_uint64, _f32p, C.c_void_p]
            al.gvref_alias_sample.restype = C.c_int
            al.gvref_alias_sample.argtypes = [_f32p, C.c_uint32, C.c_void_p, C.c_uint32, _u32p]

    @staticmethod
    def available():
        return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libgvref.so")) or os.path.exists(
            "/root/reference/include/instance/model/graph.h")

    def alias_build(self, weights, index_bytes=4):
        """AliasTable<float, uint32_t / size_t>::build of the reference (alias_table.cuh:84-128)."""
        w = np.ascontiguousarray(weights, np.float32)
        prob = np.zeros(w.size, np.float32)
        if index_bytes == 4:
            alias = np.zeros(w.size, np.uint32)
            assert self.alias_lib.gvref_alias_build(w, w.size, prob, alias) == 0
        else:
            alias = np.zeros(w.size, np.uint64)
            assert self.alias_lib.gvref_alias_build64(w, w.size, prob, alias.ctypes.data) == 0
        return prob, alias

    def alias_sample(self, weights, rand):
        """AliasTable::sample(rand1, rand2) (alias_table.cuh:148-152) for uniforms rand[m, 2] (float64)."""
        w = np.ascontiguousarray(weights, np.float32)
        r = np.ascontiguousarray(rand, np.float64).reshape(-1, 2)
        out = np.zeros(r.shape[0], np.uint32)
        assert self.alias_lib.gvref_alias_sample(w, w.size, r.ctypes.data, r.shape[0], out) == 0
        return out

    def train(self, vertex, context, batch, negatives, lr, wd, negative_weight, optimizer=SGD, moments=None,
              hp=(0, 0, 0)):
        B = batch.shape[0]
        k = negatives.size // B if B else 0
        loss = np.zeros(B, np.float32)
        m = list(moments or []) + [None] * 4
        hp = np.asarray(hp, np.float32)
        rc = self.lib.gvref_train(vertex.shape[1], optimizer, _opt(vertex), _opt(context), _opt(m[0]), _opt(m[1]),
                                  _opt(m[2]), _opt(m[3]), batch.reshape(-1), negatives.reshape(-1), loss, B, k,
                                  lr, wd, negative_weight, hp)
        assert rc == 0
        return loss

    def train_mt(self, vertex, context, batch, negatives, lr, wd, negative_weight, num_thread):
        B = batch.shape[0]
        loss = np.zeros(B, np.float32)
        rc = self.lib.gvref_train_mt(vertex.shape[1], vertex, context, batch.reshape(-1), negatives.reshape(-1),
                                     loss, B, negatives.size // B, lr, wd, negative_weight, num_thread)
        assert rc == 0
        return loss

    def predict(self, vertex, context, batch):
        out = np.zeros(batch.shape[0], np.float32)
        assert self.lib.gvref_predict(vertex.shape[1], vertex, context, batch.reshape(-1), out, batch.shape[0]) == 0
        return out

    def sigmoid(self, x):
        return self.lib.gvref_sigmoid(x)

    def lr(self, init_lr, linear, batch_id, num_batch):
        return self.lib.gvref_lr(init_lr, int(linear), batch_id, num_batch)


def link_prediction_auc(vertex, context, H, T, Y):
    """Rank AUC exactly as python/graphvite/application/application.py:433-449 (numpy restatement)."""
    score = np.einsum("ij,ij->i", vertex[H].astype(np.float32), context[T].astype(np.float32))
    order = np.argsort(-score, kind="stable")
    y = np.asarray(Y)[order]
    hit = np.cumsum(y)
    total = int((y == 0).sum()) * int((y == 1).sum())
    return float(hit[y == 0].sum()) / total


class ReferenceSolver(object):
    """The reference's own Graph / GraphSolver::build / get_schedule / get_sample_function / CPU samplers, compiled for
    the host as written (oracle/ref_solver_harness.cpp -> oracle/_ref/libgvref_solver.so).  The uniforms its samplers
    draw (cuRAND in the reference) are served from the oracle's per-thread Philox stream — the one the product's
    samplers consume — so that the edge sampler can be compared draw for draw.  One instance = one built solver."""

    PATH = os.path.join(ORACLE_DIR, "_ref", "libgvref_solver.so")
    PATHS = {128: PATH, 96: os.path.join(ORACLE_DIR, "_ref", "libgvref_solver_96.so")}  # one build per dim (oracle/Makefile)
    _libs = {}
    _callback = None

    @classmethod
    def available(cls, dim=128):
        return os.path.exists(cls.PATHS[dim])

    @classmethod
    def lib(cls, dim=128):
        if dim not in cls._libs:
            cls._libs[dim] = lib = C.CDLL(cls.PATHS[dim])
            assert lib.gvref_solver_dim() == dim
            lib.gvref_set_optimizer.argtypes = [C.c_char_p, C.c_float, C.c_float]
            lib.gvref_set_optimizer_momentum.argtypes = [C.c_float]
            lib.gvref_set_kernel_model.argtypes = [C.c_int, C.c_int, C.c_int]
            lib.gvref_solver_train.restype = C.c_int
            lib.gvref_solver_train.argtypes = [C.c_void_p, C.c_char_p] + [C.c_int] * 5 + [C.c_float] * 4 + [C.c_void_p] * 2
            lib.gvref_solver_create.restype = C.c_void_p
            lib.gvref_solver_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64] + [C.c_int] * 7
            lib.gvref_solver_destroy.argtypes = [C.c_void_p]
            lib.gvref_solver_info.argtypes = [C.c_void_p, C.c_void_p]
            lib.gvref_solver_partition.argtypes = [C.c_void_p] * 5
            lib.gvref_solver_edges.argtypes = [C.c_void_p] * 3
            lib.gvref_solver_schedule.restype = C.c_int
            lib.gvref_solver_schedule.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            lib.gvref_solver_sample.restype = C.c_int
            lib.gvref_solver_sample.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                                C.c_float, C.c_void_p]
            lib.gvref_solver_negative_table.restype = C.c_uint64
            lib.gvref_solver_negative_table.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                                        C.c_void_p, C.c_uint64]
            lib.gvref_solver_table.restype = C.c_uint64
            lib.gvref_solver_table.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
        return cls._libs[dim]

    def __init__(self, oracle, seed, edges, weights=None, as_undirected=True, num_worker=1, num_sampler_per_worker=1,
                 num_partition=0, num_negative=1, batch_size=100000, episode_size=0, dim=128, optimizer=None):
        """optimizer: None = the solver's default (SGD 0.025 / 5e-3 linear, graph.cuh:634-636) or (type, lr, weight_decay) with
        type one of core/optimizer.h's helper classes ("SGD", "Momentum", "AdaGrad", "RMSprop", "Adam"); a fourth element is
        Momentum's coefficient (the class's default: 0.999)."""
        self.dim = dim
        lib = self.lib(dim)
        lib.gvref_set_optimizer(*((b"", 0.0, 0.0) if optimizer is None else (optimizer[0].encode(), optimizer[1], optimizer[2])))
        if optimizer is not None and len(optimizer) > 3:
            lib.gvref_set_optimizer_momentum(optimizer[3])
        source_type = C.CFUNCTYPE(None, C.c_int, C.c_ulonglong, C.POINTER(C.c_double), C.c_size_t)

        def source(generator, position, out, n):  # generator index == sampler index == uniform stream
            u = oracle.host_uniforms(seed, generator, position, n)
            C.memmove(out, u.ctypes.data, n * 8)

        ReferenceSolver._callback = source_type(source)  # kept alive: the library holds the pointer
        lib.gvref_set_uniform_source(ReferenceSolver._callback)
        e = np.ascontiguousarray(edges, np.uint32)
        w = None if weights is None else np.ascontiguousarray(weights, np.float32)
        self.handle = lib.gvref_solver_create(e.ctypes.data, None if w is None else w.ctypes.data, len(e),
                                              int(as_undirected), num_worker, num_sampler_per_worker, num_partition,
                                              num_negative, batch_size, episode_size)
        info = np.zeros(8, np.int64)
        lib.gvref_solver_info(self.handle, info.ctypes.data)
        (self.num_vertex, self.num_edge, self.num_directed_edge, self.num_partition, self.episode_size,
         self.partition_size, self.num_sampler, self.num_worker) = [int(x) for x in info]
        self.batch_size = batch_size

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib(self.dim).gvref_solver_destroy(h)

    def partition(self):
        """(labels, part, local, vertex_weights) per vertex id."""
        N = self.num_vertex
        labels, part = np.zeros(N, np.uint32), np.zeros(N, np.int32)
        local, weights = np.zeros(N, np.uint32), np.zeros(N, np.float32)
        self.lib(self.dim).gvref_solver_partition(self.handle, labels.ctypes.data, part.ctypes.data, local.ctypes.data,
                                          weights.ctypes.data)
        return labels, part, local, weights

    def edges(self):
        uv, w = np.zeros((self.num_directed_edge, 2), np.uint32), np.zeros(self.num_directed_edge, np.float32)
        self.lib(self.dim).gvref_solver_edges(self.handle, uv.ctypes.data, w.ctypes.data)
        return uv, w

    def schedule(self):
        out = np.zeros(4096 * self.num_worker * 2, np.int32)
        steps = self.lib(self.dim).gvref_solver_schedule(self.handle, out.ctypes.data, 4096)
        return out[:steps * self.num_worker * 2].reshape(steps, self.num_worker, 2)

    def sample(self, model="LINE", augmentation_step=1, walk_length=40, walk_batch=100, shuffle_base=1, p=1.0, q=1.0):
        """Pools [P][P][episode_size * batch_size][2] = {tail, head} of the first episode's fill."""
        P, n = self.num_partition, self.episode_size * self.batch_size
        pools = np.zeros((P, P, n, 2), np.uint32)
        self.lib(self.dim).gvref_solver_sample(self.handle, model.encode(), augmentation_step, walk_length, walk_batch,
                                       shuffle_base, p, q, pools.ctypes.data)
        return pools

    def negative_table(self, head_partition, tail_partition, exponent=0.75, worker=0):
        """WorkerMixin::build_negative_sampler for the block: (prob, alias) over the tail partition's vertices."""
        prob, alias = np.zeros(self.partition_size, np.float32), np.zeros(self.partition_size, np.uint32)
        n = self.lib(self.dim).gvref_solver_negative_table(self.handle, worker, head_partition, tail_partition, exponent,
                                                   prob.ctypes.data, alias.ctypes.data, self.partition_size)
        return prob[:n], alias[:n]

    def table(self, which, index=0, capacity=1 << 16):
        """which: 0 the global edge table, 1 vertex_edge_tables[index], 2 edge_edge_tables[index] (after sample())."""
        prob, alias = np.zeros(capacity, np.float32), np.zeros(capacity, np.uint64)
        n = self.lib(self.dim).gvref_solver_table(self.handle, which, index, prob.ctypes.data, alias.ctypes.data, capacity)
        return prob[:n], alias[:n]


def reference_load(kind, path, a, b=0, normalization=False, delimiters=" \t\r\n", comment="#"):
    """The reference's own text loaders (oracle/ref_solver_harness.cpp): kind 0 = Graph::load_file(as_undirected=a),
    kind 1 = WordGraph::load_file_compact(window=a, min_count=b).  Returns (names, uv, edge_weights, vertex_weights,
    num_edge) of the flattened graph."""
    lib = ReferenceSolver.lib()
    lib.gvref_graph_load.restype = C.c_void_p
    lib.gvref_graph_load.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    lib.gvref_graph_destroy.argtypes = [C.c_void_p]
    lib.gvref_graph_info.argtypes = [C.c_void_p, C.c_void_p]
    lib.gvref_graph_data.argtypes = [C.c_void_p] * 5
    h = lib.gvref_graph_load(kind, path.encode(), int(a), int(b), int(normalization), delimiters.encode(), comment.encode())
    info = np.zeros(4, np.int64)
    lib.gvref_graph_info(h, info.ctypes.data)
    N, num_edge, D, name_bytes = [int(x) for x in info]
    names = C.create_string_buffer(max(name_bytes, 1))
    uv, ew, vw = np.zeros((D, 2), np.uint32), np.zeros(D, np.float32), np.zeros(N, np.float32)
    lib.gvref_graph_data(h, names, uv.ctypes.data, ew.ctypes.data, vw.ctypes.data)
    lib.gvref_graph_destroy(h)
    return names.raw[:name_bytes].decode().split("\n")[:-1] if name_bytes else [], uv, ew, vw, num_edge


def reference_train(rs, model="LINE", num_epoch=50, augmentation_step=1, walk_length=40, walk_batch=100, shuffle_base=1,
                    p=1.0, q=1.0, negative_sample_exponent=0.75, negative_weight=5.0, kernel_chunk=0, threads=1,
                    reads_at_start=False):
    """GraphSolver::train of the reference as written on a built ReferenceSolver, with the worker's kernel and negative
    draw emulated by host loops over its own model code (oracle/ref_solver_harness.cpp).  kernel_chunk 0: the samples
    of a batch one after the other; C > 0: chunk-synchronous model of the <<<8192, 512>>> launch with C warps resident
    (5120 on a V100) — lock step inside a chunk (reads_at_start: every row of the chunk read before any is written),
    last writer wins.  Returns (vertex_embeddings, context_embeddings,
    batch_id)."""
    lib = ReferenceSolver.lib(rs.dim)
    lib.gvref_set_kernel_model(int(kernel_chunk), int(bool(reads_at_start)), int(threads))
    vertex = np.zeros((rs.num_vertex, rs.dim), np.float32)
    context = np.zeros((rs.num_vertex, rs.dim), np.float32)
    batch_id = lib.gvref_solver_train(rs.handle, model.encode(), num_epoch, augmentation_step, walk_length, walk_batch,
                                      shuffle_base, p, q, negative_sample_exponent, negative_weight,
                                      vertex.ctypes.data, context.ctypes.data)
    return vertex, context, batch_id
