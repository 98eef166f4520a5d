"""Thin, typed Python face of the kernel ABI (include/gvk.h) over torch tensors.

torch supplies device memory and streams only; every computation is a libgvk.so kernel.
A `HipKernels` object is what `solver.GraphSolver` calls; tests for the host logic may inject an object
(the CPU tests run the engine over tests/hostdev instead: the same entry points served by the CPU oracle).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

OPTIMIZER_TYPES = {"SGD": _lib.SGD, "Momentum": _lib.MOMENTUM, "AdaGrad": _lib.ADAGRAD, "RMSprop": _lib.RMSPROP,
                   "Adam": _lib.ADAM}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need(t, dtype, name, device=None):
    if not isinstance(t, torch.Tensor) or t.dtype != dtype or not t.is_contiguous():
        raise ValueError("%s must be a contiguous torch tensor of dtype %s" % (name, dtype))
    if device is not None and t.device != device:
        raise ValueError("%s lives on %s, expected %s" % (name, t.device, device))
    if not t.is_cuda:
        raise ValueError("%s must live in GPU memory; graphvite_amd has no CPU training path" % name)
    return t


def alias_build(weights, index_bytes=4):
    """Host alias table (reference AliasTable::build). Returns (prob f32[n], alias u32|u64[n], packed)."""
    w = np.ascontiguousarray(weights, dtype=np.float32)
    if w.ndim != 1:
        raise ValueError("weights must be one-dimensional")
    prob = np.empty(w.size, np.float32)
    alias = np.empty(w.size, np.uint64 if index_bytes == 8 else np.uint32)
    packed = np.empty(w.size, dtype=np.dtype([("prob", np.float32), ("alias", np.uint32)])) if index_bytes == 4 \
        else None
    rc = _lib.lib().gvk_alias_build(w.ctypes.data, w.size, prob.ctypes.data, alias.ctypes.data, index_bytes,
                                    None if packed is None else packed.ctypes.data)
    _lib.check(rc, "gvk_alias_build")
    return prob, alias, packed


CLASS_ENTRY = np.dtype([("prob", np.float32), ("alias", np.uint32), ("first", np.uint32), ("count", np.uint32)])


def class_table_build(weights):
    """gvk_class_entry[num_class] over maximal runs of consecutive equal weights (gvk_class_table_build): the negative
    sampler's distribution in a table of a few thousand entries when the rows are sorted by degree."""
    w = np.ascontiguousarray(weights, dtype=np.float32)
    if w.ndim != 1 or w.size == 0:
        raise ValueError("weights must be a non-empty one-dimensional array")
    out = np.empty(w.size, CLASS_ENTRY)
    num = C.c_uint32(0)
    _lib.check(_lib.lib().gvk_class_table_build(w.ctypes.data, w.size, out.ctypes.data, C.byref(num)),
               "gvk_class_table_build")
    return out[:num.value].copy()


def classes_to_device(classes, device):
    """gvk_class_entry[n] -> int64 tensor [n, 2] on `device` (16-byte entries; the 2-D shape marks a class table)."""
    return torch.from_numpy(classes.view(np.int64).reshape(-1, 2).copy()).to(device)


def packed_to_device(packed, device):
    """gvk_alias_entry[n] -> int64 tensor [n] on `device` (8-byte entries, bit pattern preserved)."""
    return torch.from_numpy(packed.view(np.int64).copy()).to(device)


class OptimizerSpec(object):
    """Plain description of an optimizer for the kernels (type name + hyper-parameters)."""

    def __init__(self, type="SGD", lr=0.025, weight_decay=0.005, hp0=0.0, hp1=0.0, epsilon=0.0, schedule="linear"):
        if type not in OPTIMIZER_TYPES:
            raise ValueError("Unknown optimizer `%s`" % type)
        self.type, self.lr, self.weight_decay = type, float(lr), float(weight_decay)
        self.hp0, self.hp1, self.epsilon, self.schedule = float(hp0), float(hp1), float(epsilon), schedule

    @property
    def num_moment(self):
        return {"SGD": 0, "Adam": 2}.get(self.type, 1)

    def c_struct(self, lr=None):
        return _lib.Optimizer(OPTIMIZER_TYPES[self.type], self.lr if lr is None else lr, self.weight_decay, self.hp0,
                              self.hp1, self.epsilon)


class HipKernels(object):
    """Launches on torch's current stream of the tensors' device."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.lib()

    @staticmethod
    def _stream(t):
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    @staticmethod
    def _tables(vertex, context, moments):
        dev = vertex.device
        _need(vertex, torch.float32, "vertex")
        _need(context, torch.float32, "context", dev)
        if vertex.dim() != 2 or context.dim() != 2 or vertex.shape[1] != context.shape[1]:
            raise ValueError("vertex / context must be [rows, dim] with equal dim")
        m = list(moments or []) + [None] * 4
        for i, t in enumerate(m[:4]):
            if t is not None:
                _need(t, torch.float32, "moment", dev)
                if t.shape != (vertex if i % 2 == 0 else context).shape:
                    raise ValueError("moment table %d has the wrong shape" % i)
        return _lib.Tables(_ptr(vertex), _ptr(context), _ptr(m[0]), _ptr(m[1]), _ptr(m[2]), _ptr(m[3]),
                           vertex.shape[0], context.shape[0], 0, 0)

    @staticmethod
    def _negative(negatives, table, seed, dev):
        """`table`: int64 [rows] = gvk_alias_entry per row, or int64 [classes, 2] = gvk_class_entry per weight class."""
        if negatives is not None:
            _need(negatives, torch.int32, "negatives", dev)
        if table is not None:
            _need(table, torch.int64, "alias table", dev)
            if table.dim() == 2:
                return _lib.NegativeSource(_ptr(negatives), None, 0, seed, _ptr(table), table.shape[0])
        return _lib.NegativeSource(_ptr(negatives), _ptr(table), 0 if table is None else table.numel(), seed, None, 0)

    def train(self, vertex, context, pairs, loss, optimizer, num_negative, negative_weight, negatives=None,
              table=None, seed=0, batch_id=0, moments=None, lr=None):
        """One batch. pairs int32 [B, 2] = {tail, head}; negatives int32 [B, k] or None (draw from `table`)."""
        dev = vertex.device
        tables = self._tables(vertex, context, moments)
        _need(pairs, torch.int32, "pairs", dev)
        _need(loss, torch.float32, "loss", dev)
        B = pairs.shape[0]
        if pairs.dim() != 2 or pairs.shape[1] != 2 or loss.numel() < B:
            raise ValueError("pairs must be [B, 2] and loss must hold B floats")
        if negatives is not None and negatives.numel() != B * num_negative:
            raise ValueError("negatives must hold batch_size * num_negative ids")
        neg = self._negative(negatives, table, seed, dev)
        opt = optimizer.c_struct(lr)
        rc = self.lib.gvk_train(self._stream(vertex), vertex.shape[1], C.byref(opt), C.byref(tables), _ptr(pairs),
                                C.byref(neg), batch_id, _ptr(loss), B, num_negative, negative_weight)
        _lib.check(rc, "gvk_train")

    def train_episode(self, vertex, context, pool, loss, optimizer, num_negative, negative_weight, table, seed,
                      first_batch_id, total_batches, num_batches, batch_size, moments=None, batch_id_stride=1):
        """num_batches consecutive batches of a device-resident pool (int32 [>= num_batches*batch_size, 2])."""
        dev = vertex.device
        tables = self._tables(vertex, context, moments)
        _need(pool, torch.int32, "pool", dev)
        _need(loss, torch.float32, "loss", dev)
        if pool.numel() < num_batches * batch_size * 2 or loss.numel() < batch_size:
            raise ValueError("pool / loss too small for %d batches of %d" % (num_batches, batch_size))
        neg = self._negative(None, table, seed, dev)
        opt = optimizer.c_struct()
        rc = self.lib.gvk_train_episode(self._stream(vertex), vertex.shape[1], C.byref(opt),
                                        int(optimizer.schedule == "linear"), C.byref(tables), _ptr(pool),
                                        C.byref(neg), first_batch_id, batch_id_stride, total_batches, num_batches, _ptr(loss),
                                        batch_size, num_negative, negative_weight)
        _lib.check(rc, "gvk_train_episode")

    def hot_plan(self, dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts=1, chain_cap=0):
        """Bytes of device workspace the chains' work lists of num_batch batches (each trained as `parts` parts) and the
        mirrors of the hub rows need."""
        n = C.c_size_t(0)
        _lib.check(self.lib.gvk_hot_plan(dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts, chain_cap,
                                         C.byref(n)), "gvk_hot_plan")
        return n.value

    def hot_build(self, dim, workspace, pool, batch_size, num_batch, num_negative, table, seed, first_batch_id, hot_vertex,
                  hot_context, batch_id_stride=1, parts=1, chain_cap=0):
        """The work lists of the hub rows' chains for num_batch batches of a device pool (gvk_hot_build); workspace: uint8
        device tensor of hot_plan() bytes."""
        dev = pool.device
        _need(pool, torch.int32, "pool", dev)
        neg = self._negative(None, table, seed, dev)
        rc = self.lib.gvk_hot_build(self._stream(pool), dim, _ptr(workspace), workspace.numel(), _ptr(pool), batch_size, num_batch,
                                    num_negative, C.byref(neg), first_batch_id, batch_id_stride, hot_vertex, hot_context, parts,
                                    chain_cap)
        _lib.check(rc, "gvk_hot_build")

    HOT_SERIALIZED, HOT_LERP = 1, 2  # gvk.h GVK_HOT_SERIALIZED, GVK_HOT_LERP

    def train_episode_hot(self, vertex, context, pool, loss, optimizer, num_negative, negative_weight, table, seed,
                          first_batch_id, total_batches, num_batches, batch_size, workspace, hot_vertex, hot_context,
                          workspace_batches=None, batch_id_stride=1, serialized=False, parts=1, chain_cap=0, lerp=False, moments=None):
        """gvk_train_episode_hot: batches whose hub rows are trained by chains (work lists from hot_build in `workspace`);
        moments = [vm1, cm1(, vm2, cm2)] for a moment optimizer (its chains: one sequential task per hub row)."""
        dev = vertex.device
        tables = self._tables(vertex, context, moments)
        _need(pool, torch.int32, "pool", dev)
        _need(loss, torch.float32, "loss", dev)
        neg = self._negative(None, table, seed, dev)
        opt = optimizer.c_struct()
        rc = self.lib.gvk_train_episode_hot(self._stream(vertex), vertex.shape[1], C.byref(opt),
                                            int(optimizer.schedule == "linear"), C.byref(tables), _ptr(pool), C.byref(neg),
                                            first_batch_id, batch_id_stride, total_batches, num_batches, _ptr(loss), batch_size,
                                            num_negative, negative_weight, _ptr(workspace), workspace.numel(), hot_vertex,
                                            hot_context, num_batches if workspace_batches is None else workspace_batches,
                                            parts, chain_cap, (self.HOT_SERIALIZED if serialized else 0) | (self.HOT_LERP if lerp else 0))
        _lib.check(rc, "gvk_train_episode_hot")

    HOT_ROUNDS = 4  # gvk.h GVK_HOT_ROUNDS

    def ahead_plan(self, dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts=1, chain_cap=0):
        """gvk_ahead_plan: bytes of workspace of the chain-stream executor (work lists, versions, slot words, the ring)."""
        n = C.c_size_t(0)
        _lib.check(self.lib.gvk_ahead_plan(dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts, chain_cap,
                                           C.byref(n)), "gvk_ahead_plan")
        return n.value

    def ahead_build(self, dim, workspace, pool, batch_size, num_batch, num_negative, table, seed, first_batch_id, hot_vertex,
                    hot_context, batch_id_stride=1, parts=1, chain_cap=0, group=1):
        """gvk_ahead_build: the work lists of gvk_hot_build + the hub rows' versions and the slots that name them."""
        dev = pool.device
        _need(pool, torch.int32, "pool", dev)
        neg = self._negative(None, table, seed, dev)
        rc = self.lib.gvk_ahead_build(self._stream(pool), dim, _ptr(workspace), workspace.numel(), _ptr(pool), batch_size, num_batch,
                                      num_negative, C.byref(neg), first_batch_id, batch_id_stride, hot_vertex, hot_context, parts,
                                      chain_cap, group)
        _lib.check(rc, "gvk_ahead_build")

    def train_episode_ahead(self, vertex, context, pool, loss, optimizer, num_negative, negative_weight, table, seed,
                            first_batch_id, total_batches, num_batches, batch_size, workspace, hot_vertex, hot_context,
                            workspace_batches=None, batch_id_stride=1, serialized=False, parts=1, chain_cap=0, pair_launches=0,
                            rounds=False, chain_stream=None, group=1):
        """gvk_train_episode_ahead: the chains on `chain_stream` (a torch.cuda.Stream; made once per device when not given), a batch
        ahead of the pairs on the current stream."""
        dev = vertex.device
        tables = self._tables(vertex, context, None)
        _need(pool, torch.int32, "pool", dev)
        _need(loss, torch.float32, "loss", dev)
        neg = self._negative(None, table, seed, dev)
        opt = optimizer.c_struct()
        side = 0
        if not serialized and group == 1:
            if chain_stream is None:
                streams = self.__dict__.setdefault("_chain_streams", {})
                chain_stream = streams.get(dev)
                if chain_stream is None:
                    chain_stream = streams[dev] = torch.cuda.Stream(device=dev)
            side = chain_stream.cuda_stream
        rc = self.lib.gvk_train_episode_ahead(self._stream(vertex), side, vertex.shape[1], C.byref(opt),
                                              int(optimizer.schedule == "linear"), C.byref(tables), _ptr(pool), C.byref(neg),
                                              first_batch_id, batch_id_stride, total_batches, num_batches, _ptr(loss), batch_size,
                                              num_negative, negative_weight, _ptr(workspace), workspace.numel(), hot_vertex,
                                              hot_context, num_batches if workspace_batches is None else workspace_batches,
                                              parts, chain_cap, pair_launches, group,
                                              (self.HOT_SERIALIZED if serialized else 0) | (self.HOT_ROUNDS if rounds else 0))
        _lib.check(rc, "gvk_train_episode_ahead")

    def predict(self, vertex, context, pairs, logits):
        dev = vertex.device
        _need(vertex, torch.float32, "vertex")
        _need(context, torch.float32, "context", dev)
        _need(pairs, torch.int32, "pairs", dev)
        _need(logits, torch.float32, "logits", dev)
        if logits.numel() < pairs.shape[0]:
            raise ValueError("logits too small")
        rc = self.lib.gvk_predict(self._stream(vertex), vertex.shape[1], _ptr(vertex), _ptr(context), _ptr(pairs),
                                  _ptr(logits), pairs.shape[0])
        _lib.check(rc, "gvk_predict")

    def probe_row_traffic(self, vertex, context, pairs, negatives, bump=0.0):
        """Measurement aid (gvk_probe_row_traffic): read and write back the head, tail and given negative row of every
        pair — the memory traffic of a training batch (SGD, one negative) without its arithmetic."""
        dev = vertex.device
        _need(vertex, torch.float32, "vertex")
        _need(context, torch.float32, "context", dev)
        _need(pairs, torch.int32, "pairs", dev)
        _need(negatives, torch.int32, "negatives", dev)
        n = pairs.numel() // 2
        if negatives.numel() < n:
            raise ValueError("one negative per pair")
        rc = self.lib.gvk_probe_row_traffic(self._stream(vertex), vertex.shape[1], _ptr(vertex), _ptr(context), _ptr(pairs),
                                            _ptr(negatives), bump, n)
        _lib.check(rc, "gvk_probe_row_traffic")

    def alias_sample(self, table, rand, result):
        _need(table, torch.int64, "alias table")
        _need(rand, torch.float64, "rand", table.device)
        _need(result, torch.int32, "result", table.device)
        rc = self.lib.gvk_alias_sample(self._stream(table), _ptr(table), table.numel(), _ptr(rand), _ptr(result),
                                       result.numel())
        _lib.check(rc, "gvk_alias_sample")

    def negative_draw(self, table, seed, batch_id, out, batch_size, num_negative):
        """The negatives of a batch as the training kernels draw them, from a row table [rows] or a class table [n, 2]."""
        _need(table, torch.int64, "alias table")
        _need(out, torch.int32, "negatives", table.device)
        if table.dim() == 2:
            rc = self.lib.gvk_negative_draw_classes(self._stream(table), _ptr(table), table.shape[0], seed, batch_id,
                                                    _ptr(out), batch_size, num_negative)
            return _lib.check(rc, "gvk_negative_draw_classes")
        rc = self.lib.gvk_negative_draw(self._stream(table), _ptr(table), table.numel(), seed, batch_id, _ptr(out),
                                        batch_size, num_negative)
        _lib.check(rc, "gvk_negative_draw")

    def sample_pairs(self, table, block_pairs, seed, first_index, pool, n):
        """pool[:n] = positive pairs of one block drawn on the device (gvk_sample_pairs)."""
        _need(table, torch.int64, "block alias table")
        _need(block_pairs, torch.int32, "block pairs", table.device)
        _need(pool, torch.int32, "pool", table.device)
        if block_pairs.numel() != 2 * table.numel() or pool.numel() < 2 * n:
            raise ValueError("block_pairs must hold one {tail, head} record per table entry and pool 2 * n values")
        rc = self.lib.gvk_sample_pairs(self._stream(table), _ptr(table), _ptr(block_pairs), table.numel(), seed,
                                       first_index, _ptr(pool), n)
        _lib.check(rc, "gvk_sample_pairs")

    @staticmethod
    def pack_edge_table(table, block_pairs):
        """gvk_edge_entry[count] (int64 [count, 2]) from a block's alias table and its {tail, head} records."""
        return torch.cat([table.view(torch.int32).view(-1, 2), block_pairs.view(-1, 2)], 1).contiguous().view(torch.int64)

    def sample_edges(self, edge_table, seed, first_index, pool, n):
        """pool[:n] = positive pairs of one block from its packed table (gvk_sample_edges): the draws of sample_pairs."""
        _need(edge_table, torch.int64, "packed block table")
        _need(pool, torch.int32, "pool", edge_table.device)
        if edge_table.dim() != 2 or edge_table.shape[1] != 2 or pool.numel() < 2 * n:
            raise ValueError("edge_table must be [count, 2] int64 (16-byte entries) and pool must hold 2 * n values")
        rc = self.lib.gvk_sample_edges(self._stream(edge_table), _ptr(edge_table), edge_table.shape[0], seed, first_index,
                                       _ptr(pool), n)
        _lib.check(rc, "gvk_sample_edges")

    def group_pairs(self, pool_in, pool_out, batch_size, num_batch, num_row):
        """pool_out = pool_in with the pairs of every batch that share a head row made adjacent (gvk_group_pairs).
        Runs on the current stream; the scratch buffer comes from torch's caching allocator."""
        dev = pool_in.device
        _need(pool_in, torch.int32, "pool_in", dev)
        _need(pool_out, torch.int32, "pool_out", dev)
        n = 2 * batch_size * num_batch
        if pool_in.numel() < n or pool_out.numel() < n:
            raise ValueError("pools hold fewer than %d batches of %d pairs" % (num_batch, batch_size))
        row_bits = max(int(num_row - 1).bit_length(), 1)
        need = C.c_size_t(0)
        _lib.check(self.lib.gvk_group_pairs(None, None, None, None, C.byref(need), batch_size, num_batch, row_bits),
                   "gvk_group_pairs")
        work = torch.empty(need.value, dtype=torch.uint8, device=dev)
        rc = self.lib.gvk_group_pairs(self._stream(pool_in), _ptr(pool_in), _ptr(pool_out), _ptr(work), C.byref(need),
                                      batch_size, num_batch, row_bits)
        _lib.check(rc, "gvk_group_pairs")

    def spread_pairs(self, pool_in, pool_out, num_pair, units):
        """pool_out = pool_in with record i at place (i % units) * (num_pair / units) + i / units (gvk_spread_pairs)."""
        dev = pool_in.device
        _need(pool_in, torch.int32, "pool_in", dev)
        _need(pool_out, torch.int32, "pool_out", dev)
        if pool_in.numel() < 2 * num_pair or pool_out.numel() < 2 * num_pair:
            raise ValueError("pools hold fewer than %d pairs" % num_pair)
        _lib.check(self.lib.gvk_spread_pairs(self._stream(pool_in), _ptr(pool_in), _ptr(pool_out), num_pair, units), "gvk_spread_pairs")

    def sample_walks(self, walk_graph, seed, first_walk, pool, pool_pairs, walk_length, augmentation_step,
                     shuffle_base):
        """pool[:pool_pairs] = random-walk positive pairs drawn on the device (gvk_sample_walks).
        walk_graph: dict of device tensors flat_offsets (int64), edges_uv (int32), edge_table / neighbor_table (int64
        packed alias entries), local (int32), optional sorted_neighbors (int32), plus biased / p / q."""
        g = walk_graph
        dev = g["edges_uv"].device
        desc = self._walk_graph(g, dev)
        _need(pool, torch.int32, "pool", dev)
        if pool.numel() < 2 * pool_pairs:
            raise ValueError("pool too small")
        rc = self.lib.gvk_sample_walks(self._stream(pool), C.byref(desc), seed, first_walk, _ptr(pool), pool_pairs,
                                       walk_length, augmentation_step, shuffle_base)
        _lib.check(rc, "gvk_sample_walks")

    @staticmethod
    def _walk_graph(g, dev):
        _need(g["flat_offsets"], torch.int64, "flat_offsets", dev)
        _need(g["edges_uv"], torch.int32, "edges_uv", dev)
        _need(g["edge_table"], torch.int64, "edge_table", dev)
        _need(g["neighbor_table"], torch.int64, "neighbor_table", dev)
        _need(g["local"], torch.int32, "local", dev)
        biased = bool(g.get("biased", False))
        snb = g.get("sorted_neighbors")
        if biased:
            _need(snb, torch.int32, "sorted_neighbors", dev)
        D = g["edge_table"].numel()
        if g["edges_uv"].numel() != 2 * D or g["neighbor_table"].numel() != D:
            raise ValueError("edge arrays / tables disagree in size")
        return _lib.WalkGraph(_ptr(g["flat_offsets"]), _ptr(g["edges_uv"]), _ptr(g["edge_table"]),
                              _ptr(g["neighbor_table"]), _ptr(snb), _ptr(g["local"]), g["local"].numel(), D, int(biased),
                              float(g.get("p", 1.0)), float(g.get("q", 1.0)))

    def sample_walks_blocks(self, walk_graph, part, num_partition, seed, first_walk, pools, offsets, capacity, walk_length,
                            augmentation_step, shuffle_base, max_rounds=64):
        """Random-walk positives for SEVERAL partitions, drawn on the device (gvk_sample_walks_blocks): fills the pool
        of every block b with offsets[b] >= 0 — `capacity` {tail, head} records at pools[2 * offsets[b]:] — repeating
        the call until all of them are full.  part: int32 [num_vertex]; offsets: int64 [P * P] on the device (-1 = block
        not collected).  The first round assumes equal block shares; later rounds are sized from the shares the counters
        show.  Returns the number of walks consumed (the caller advances first_walk by it)."""
        g = walk_graph
        dev = g["edges_uv"].device
        desc = self._walk_graph(g, dev)
        _need(part, torch.int32, "part", dev)
        _need(pools, torch.int32, "pools", dev)
        _need(offsets, torch.int64, "offsets", dev)
        P = int(num_partition)
        if offsets.numel() != P * P or part.numel() != g["local"].numel():
            raise ValueError("offsets must hold P * P entries and part one entry per vertex")
        wanted = (offsets >= 0).cpu().numpy()
        if not wanted.any():
            return 0
        aug, L = int(augmentation_step), int(walk_length)
        per_walk = aug * L - aug * (aug - 1) // 2
        # stripes: slot counters per pool (one per wavefront-id class) — the largest divisor of the capacity up to 256
        stripes = max(d for d in range(1, 257) if capacity % d == 0)
        stripe_capacity = capacity // stripes
        counters = torch.zeros(P * P * stripes, dtype=torch.int32, device=dev)
        every = 64 * stripes  # walks per launch: every stripe gets the same number of wavefronts
        walks = (-(-capacity * P * P // per_walk) // every + 1) * every
        used = 0
        self.walk_rounds = []  # walks launched per round of the last call (diagnostics)
        for _ in range(max_rounds):
            self.walk_rounds.append(walks)
            rc = self.lib.gvk_sample_walks_blocks(self._stream(pools), C.byref(desc), _ptr(part), P, seed, first_walk + used,
                                                  walks, _ptr(pools), _ptr(offsets), _ptr(counters), capacity, stripes, L,
                                                  aug, shuffle_base)
            _lib.check(rc, "gvk_sample_walks_blocks")
            used += walks
            # fences the stream: a handful of round trips per episode, while the GPU trains the episode before
            count = counters.cpu().numpy().astype(np.int64).reshape(P * P, stripes)[wanted]
            deficit = np.maximum(stripe_capacity - count, 0)
            if not deficit.any():
                return used
            share = count / float(used * per_walk)  # fraction of all pairs that fell into each wanted stripe
            if (share[deficit > 0] == 0).any() and used * per_walk > 64 * capacity * P * P:
                raise ValueError("a block of the partition grid receives no random-walk pairs; use fewer partitions")
            floor = 1.0 / (64 * P * P * stripes)
            need = deficit[deficit > 0] / np.maximum(share[deficit > 0], floor) / per_walk
            walks = (int(need.max() * 1.1) // every + 1) * every
        raise RuntimeError("gvk_sample_walks_blocks: pools not full after %d rounds" % max_rounds)

    def set_lanes_per_pair(self, lanes):
        _lib.check(self.lib.gvk_set_tuning(_lib.TUNE_LANES_PER_PAIR, lanes), "gvk_set_tuning")

    def set_variant(self, variant):
        _lib.check(self.lib.gvk_set_tuning(_lib.TUNE_VARIANT, variant), "gvk_set_tuning")

    def set_run_cap(self, run_cap):
        """Longest run of adjacent same-head pairs one lane group trains in sequence (0 = from the batch size)."""
        _lib.check(self.lib.gvk_set_tuning(_lib.TUNE_RUN_CAP, run_cap), "gvk_set_tuning")

    def set_segment_steps(self, steps):
        """train_segment_kernel: pairs per lane group and wavefront (0 = per-dim default, 1, 2 or 4)."""
        _lib.check(self.lib.gvk_set_tuning(_lib.TUNE_SEGMENT_STEPS, steps), "gvk_set_tuning")

    def set_tuning(self, key, value):
        """Any GVK_TUNE_* knob of include/gvk.h by number (experiments)."""
        _lib.check(self.lib.gvk_set_tuning(int(key), int(value)), "gvk_set_tuning")

    def train_launches(self, batch_size, num_row):
        """Q: a batch on a head table of num_row rows is trained as Q launches of batch_size / Q samples (gvk_train_launches);
        pools are regrouped per part."""
        return int(self.lib.gvk_train_launches(int(batch_size), int(num_row)))

    def set_split_hits(self, hits):
        """GVK_TUNE_SPLIT_HITS: samples per table row one launch may hold (default 4; 0 = one launch per batch)."""
        _lib.check(self.lib.gvk_set_tuning(_lib.TUNE_SPLIT_HITS, hits), "gvk_set_tuning")

    @property
    def has_ab_builds(self):
        """True when the loaded library is the A/B library (GVK_LIBRARY=.../libgvk_ab.so)."""
        return bool(self.lib.gvk_has_ab_builds())

    def set_generation(self, samples):
        """Parity experiment: train every batch as consecutive launches of at most `samples` samples (0 = off)."""
        _lib.check(self.lib.gvk_set_tuning(_lib.TUNE_GENERATION, samples), "gvk_set_tuning")

    def describe_train(self, dim, optimizer_type="SGD", num_negative=1, explicit_negatives=False, batch_size=100000,
                       num_row=1 << 20):
        """Name of the kernel gvk_train launches for this configuration (num_row rows in the head table) under the
        current tuning."""
        name = C.create_string_buffer(160)
        _lib.check(self.lib.gvk_describe_train(dim, OPTIMIZER_TYPES[optimizer_type], num_negative,
                                               int(explicit_negatives), batch_size, num_row, name, len(name)),
                   "gvk_describe_train")
        return name.value.decode()
