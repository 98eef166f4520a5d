"""graphvite_amd.solver — `GraphSolver`, the drop-in for graphvite.solver.GraphSolver (pyGraphSolver, include/bind.h:383-513;
GraphSolver / SolverMixin / WorkerMixin, include/instance/graph.cuh:586-813, include/core/solver.h:87-888,1170-1623).

A BINDING, not an orchestrator: everything a training run does — partitions, schedule, samplers, uploads, the slot claim and
the all-gather of head shards over RCCL, the routing of random-walk pools, write-back — lives in the native engine
(include/gvx.h, graphvite_amd/csrc/gvx_engine.cpp), the same code the pybind11 module `libgraphvite` binds.  This file
turns keyword arguments into gvx_* calls and hands back numpy views.  Two ways to use several GPUs:

  * one process (the reference's way): GraphSolver(dim, device_ids=[0, 1, 2, 3]);
  * one process per GPU: launch with `python -m torch.distributed.run --nproc-per-node N ...`, call
    torch.distributed.init_process_group first; every process constructs the same solver, which then is worker `rank` of
    the job (the RCCL unique id of the engine travels through one torch.distributed broadcast; after that torch is not
    involved).

There is no CPU training path: without a GPU, GraphSolver raises.  (The CPU tests load a host build of the same engine
through GVK_LIBRARY — tests/hostdev — and, for several processes, hand it a gloo transport.)
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .base import auto, dtype
from .graph import Graph
from .optimizer import SGD, Optimizer

_OPTIMIZER_TYPES = {"Default": -1, "SGD": _lib.SGD, "Momentum": _lib.MOMENTUM, "AdaGrad": _lib.ADAGRAD,
                    "RMSprop": _lib.RMSPROP, "Adam": _lib.ADAM}
_MODES = {_lib.MODE_EDGE: "edge", _lib.MODE_WALK: "walk", _lib.MODE_BIASED_WALK: "biased_walk",
          _lib.MODE_BIASED_REJECT: "biased_reject"}
_PAIR_ORDERS = {auto: 0, "sampled": 1, "grouped": 2}
_NEGATIVE_TABLES = {"auto": 0, "rows": 1, "classes": 2}
_TRAIN_DEFAULTS = dict(model="LINE", num_epoch=2000, resume=False, augmentation_step=auto, random_walk_length=40,
                       random_walk_batch_size=100, shuffle_base=auto, p=1, q=1, positive_reuse=1,
                       negative_sample_exponent=0.75, negative_weight=5, log_frequency=1000)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def _train_config(kwargs):
    unknown = set(kwargs) - set(_TRAIN_DEFAULTS)
    if unknown:
        raise TypeError("unexpected training argument(s): %s" % ", ".join(sorted(unknown)))
    kw = dict(_TRAIN_DEFAULTS, **kwargs)
    c = _lib.TrainConfig()
    c.model = str(kw["model"]).encode()
    c.num_epoch, c.resume = int(kw["num_epoch"]), int(bool(kw["resume"]))
    c.augmentation_step = int(kw["augmentation_step"])
    c.random_walk_length, c.random_walk_batch_size = int(kw["random_walk_length"]), int(kw["random_walk_batch_size"])
    c.shuffle_base = int(kw["shuffle_base"])
    c.p, c.q = float(kw["p"]), float(kw["q"])
    c.positive_reuse = int(kw["positive_reuse"])
    c.negative_sample_exponent, c.negative_weight = float(kw["negative_sample_exponent"]), float(kw["negative_weight"])
    c.log_frequency = int(kw["log_frequency"])
    return c


class _GlooTransport(object):
    """gvx_transport over torch.distributed on HOST memory — how the CPU tests run the engine's multi-process path (host
    build of the engine, gloo).  The engine's pointers are wrapped as tensors; both calls are synchronous."""

    def __init__(self, dist):
        import torch
        self.dist, self.torch = dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.error = None
        self._gather = _lib.TRANSPORT_ALL_GATHER(self.all_gather)
        self._exchange = _lib.TRANSPORT_ALL_TO_ALL(self.all_to_all)
        self.struct = _lib.Transport(self._gather, self._exchange, None)

    def _tensor(self, pointer, nbytes):
        buffer = (C.c_uint8 * nbytes).from_address(pointer)
        return self.torch.frombuffer(buffer, dtype=self.torch.uint8)

    def all_gather(self, user, slab, nbytes, stream):
        try:
            whole = self._tensor(slab, nbytes * self.world)
            mine = whole[self.rank * nbytes:(self.rank + 1) * nbytes].clone()
            self.dist.all_gather_into_tensor(whole, mine)
            return _lib.GVK_OK
        except BaseException as e:  # surfaces as the engine's error; the exception itself is re-raised by the caller
            self.error = e
            return _lib.GVK_EHIP

    def all_to_all(self, user, send, recv, nbytes, stream):
        try:
            self.dist.all_to_all_single(self._tensor(recv, nbytes * self.world), self._tensor(send, nbytes * self.world).clone())
            return _lib.GVK_OK
        except BaseException as e:
            self.error = e
            return _lib.GVK_EHIP


class TrainingSession(object):
    """A training run step by step (include/gvx.h gvx_session_*; benchmarks, custom loops).  Every call acts on all LOCAL
    workers at once (all of them in one process, one per process otherwise).

        session = solver.session(model="LINE", num_epoch=10, augmentation_step=1)
        session.fill(0)                                   # the pools of an episode, pool set 0
        for step in range(session.steps):
            session.stage(step, 0, step & 1)              # H2D copy / regrouping pass into device buffer step & 1
            session.train(step, 0, step & 1)              # the block's batches
            session.exchange(step)                        # all-gather of the head shards (no-op with one worker)
        session.close()                                   # device tables -> solver.vertex_embeddings / context_embeddings
    """

    def __init__(self, solver, resident_pools=False, **train_kwargs):
        self.solver = solver
        self._lib = solver._lib
        self._handle = solver._handle
        config = _train_config(train_kwargs)
        solver._apply_options()
        solver._check(self._lib.gvx_session_open(self._handle, C.byref(config), int(bool(resident_pools))), "session")
        self.open = True
        solver._refresh()
        self.steps = self._lib.gvx_session_steps(self._handle)

    def block(self, step, worker=0):
        """(head partition, tail partition) local worker `worker` trains at `step`."""
        hp, tp = C.c_int(), C.c_int()
        self.solver._check(self._lib.gvx_session_block(self._handle, step, worker, C.byref(hp), C.byref(tp)), "session.block")
        return hp.value, tp.value

    def fill(self, pool_set=0):
        self.solver._check(self._lib.gvx_session_fill(self._handle, pool_set), "session.fill")

    def stage(self, step, pool_set=0, buffer=0):
        self.solver._check(self._lib.gvx_session_stage(self._handle, step, pool_set, buffer), "session.stage")

    def train(self, step, pool_set=0, buffer=0, first=0, count=None):
        """Batches [first, first + count) of the pool staged for `step` (default: all of the episode)."""
        if count is None:
            count = self.solver.episode_size - first
        self.solver._check(self._lib.gvx_session_train(self._handle, step, pool_set, buffer, first, count), "session.train")
        self.solver.batch_id += count * self.solver.num_worker

    def exchange(self, step):
        self.solver._check(self._lib.gvx_session_exchange(self._handle, step), "session.exchange")

    def wait(self):
        """Order the compute streams behind every pending exchange (a fence for timed regions)."""
        self.solver._check(self._lib.gvx_session_wait(self._handle), "session.wait")

    def synchronize(self):
        self.solver._check(self._lib.gvx_session_synchronize(self._handle), "session.synchronize")

    def stream(self, worker=0):
        """The compute stream (hipStream_t, as an integer) of a local worker: where events around train() belong."""
        return self._lib.gvx_session_stream(self._handle, worker)

    def loss(self, worker=0):
        """Mean per-sample loss of the worker's most recent batch."""
        out = C.c_float()
        self.solver._check(self._lib.gvx_session_loss(self._handle, worker, C.byref(out)), "session.loss")
        return out.value

    def probe(self, step=0, pool_set=0, buffer=0, launches=200):
        """gvk_probe_row_traffic on the block of `step`: milliseconds per launch (the ceiling of the access pattern)."""
        out = C.c_float()
        self.solver._check(self._lib.gvx_session_probe(self._handle, step, pool_set, buffer, launches, C.byref(out)),
                           "session.probe")
        return out.value

    def close(self):
        if self.open:
            self.open = False
            self.solver.exchange_stats = self.solver._exchange_stats()
            self.solver._check(self._lib.gvx_session_close(self._handle), "session.close")
            self.solver._refresh()

    finish = close


class GraphSolver(object):
    """
    GraphSolver(dim, float_type=dtype.float32, index_type=dtype.uint32, device_ids=[], num_sampler_per_worker=auto,
                gpu_memory_limit=auto)
    Graph embedding solver.

    Parameters:
        dim (int): dimension of embeddings (32, 64, 96, 128, 256 or 512)
        float_type (dtype): type of parameters (float32)
        index_type (dtype): type of node indexes (uint32)
        device_ids (list of int, optional): GPU ids, [] for auto.  In a torch.distributed job (one process per GPU) the list
            names the GPUs of the whole job and this process drives device_ids[rank] (default: LOCAL_RANK); otherwise this
            process drives all of them, one worker per entry.
        num_sampler_per_worker (int, optional): number of sampler threads per GPU
        gpu_memory_limit (int, optional): memory limit for each GPU in bytes

    Beyond the reference's arguments: `seed`, `device_sampling` (draw the positive samples on the GPU) and `pair_order` —
    "sampled": a batch is trained in the order the samplers produced it; "grouped": the pairs of a batch that share a head
    row are made adjacent on the device first (gvk_group_pairs; same samples, same batches; a row shared by k samples is
    fetched from HBM once and the k samples are trained as runs, one after the other on one copy of the row when the table
    is small); auto (default): by table size.
    """

    available_dims = (32, 64, 96, 128, 256, 512)  # src/graphvite.cu:52-59
    available_models = ("DeepWalk", "LINE", "node2vec")

    def __init__(self, dim, float_type=dtype.float32, index_type=dtype.uint32, device_ids=(), num_sampler_per_worker=auto,
                 gpu_memory_limit=auto, seed=0, device_sampling=False, pair_order=auto, hub_rows=None,
                 fidelity=auto):
        if dim not in self.available_dims or float_type != dtype.float32 or index_type != dtype.uint32:
            raise AttributeError("Can't find an instantiation of GraphSolver with dim=%s, float_type=%s, "
                                 "index_type=%s" % (dim, float_type, index_type))
        if pair_order not in _PAIR_ORDERS:
            raise ValueError("pair_order must be auto, 'sampled' or 'grouped', not %r" % (pair_order,))
        self.dim = dim
        self._lib = lib = _lib.lib()
        self._handle = None
        self._transport = None
        self._schedule_callback = None
        self._schedule_error = None
        dist = _dist()
        device_ids = [int(d) for d in device_ids]
        world = dist.get_world_size() if dist else 1
        if world > 1:
            rank = dist.get_rank()
            if device_ids and len(device_ids) != world:
                raise ValueError("%d GPUs listed but this job has %d processes (one per GPU)" % (len(device_ids), world))
            device = device_ids[rank] if device_ids else int(os.environ.get("LOCAL_RANK", rank))
            unique_id, transport = b"", None
            if dist.get_backend() == "gloo":  # host build of the engine (CPU tests): collectives over gloo
                self._transport = _GlooTransport(dist)
                transport = C.byref(self._transport.struct)
                device = 0
            else:
                unique_id = self._broadcast_unique_id(dist, rank, device)
            handle = lib.gvx_solver_create_distributed(dim, rank, world, device, unique_id, len(unique_id), transport,
                                                       int(num_sampler_per_worker), int(gpu_memory_limit))
        else:
            ids = (C.c_int * max(len(device_ids), 1))(*device_ids)
            handle = lib.gvx_solver_create(dim, ids, len(device_ids), int(num_sampler_per_worker), int(gpu_memory_limit))
        if not handle:
            message = lib.gvk_last_error().decode("utf-8", "replace")
            if "No GPU" in message:
                raise RuntimeError(message + " (graphvite_amd has no CPU training path)")
            raise ValueError(message)
        self._handle = handle
        self.seed = int(seed)
        self.device_sampling = bool(device_sampling)
        self._pair_order_request = pair_order
        # GVX_HUB_ROWS (gvx.h): None = the default rule, "auto" = by expected hits per batch, 0 = off, N = the first N rows
        self.hub_rows_request = -2 if hub_rows is None else (-1 if hub_rows == "auto" else int(hub_rows))
        self.hub_parts = 0  # GVX_HUB_PARTS (gvx.h): 0 = the rule
        self.hub_lerp = None  # GVX_HUB_LERP (gvx.h): None = the rule, False / True
        self.hub_rounds = None  # GVX_HUB_ROUNDS (gvx.h): None = the rule, False / True: long chains in one round / in rounds
        self.hub_chain_cap = 0  # GVX_HUB_CHAIN_CAP (gvx.h): 0 = the kernels' default
        # GVX_HUB_EXECUTOR (gvx.h): None = the rule (the chains as a stream of their own, a batch ahead of the pairs), "fused" = one
        # launch per part of a batch carries its pairs and the next part's chains, "ahead" = the chain stream, "grouped" = one launch for the
        # chains of hub_group parts and the pairs of the parts before them
        self.hub_executor = os.environ.get("GVX_HUB_EXECUTOR") or None
        self.hub_pair_launches = int(os.environ.get("GVX_HUB_PAIR_LAUNCHES", "0"))  # GVX_HUB_PAIR_LAUNCHES: 0 = one launch per part
        self.hub_group = int(os.environ.get("GVX_HUB_GROUP", "0"))  # GVX_HUB_GROUP: parts whose chains share a launch under "grouped" (0 = the rule's)
        if fidelity is not auto and fidelity not in ("auto", "throughput", "reference"):
            raise ValueError("fidelity must be auto, 'throughput' or 'reference', not %r" % (fidelity,))
        self.fidelity = "auto" if fidelity is auto else fidelity  # GVX_FIDELITY (gvx.h)
        self.negative_table = "auto"          # "rows": one alias slot per row (the reference's); "classes": by weight class
        self.node2vec_table_limit = 1 << 30   # per-edge table entries before node2vec samples by rejection
        self.graph = None
        self.optimizer = None
        self.exchange_stats = {"exchanges": 0, "bytes_sent_per_gpu": 0}
        self.timing = None
        from .kernels import HipKernels
        self.kernels = HipKernels()  # tuning knobs and gvk_describe_train
        self._refresh()

    def _broadcast_unique_id(self, dist, rank, device):
        import torch
        buffer = (C.c_char * _lib.GVX_UNIQUE_ID_BYTES)()
        if rank == 0:
            _lib.check(self._lib.gvx_unique_id(buffer, len(buffer)), "gvx_unique_id")
        t = torch.frombuffer(bytearray(buffer.raw), dtype=torch.uint8).clone()
        if dist.get_backend() == "nccl":
            t = t.to(torch.device("cuda", device))
        dist.broadcast(t, 0)
        return t.cpu().numpy().tobytes()

    def __del__(self):
        handle, self._handle = getattr(self, "_handle", None), None
        if handle:
            self._lib.gvx_solver_destroy(handle)

    def _check(self, rc, what):
        if rc != _lib.GVK_OK and self._transport is not None and self._transport.error is not None:
            error, self._transport.error = self._transport.error, None
            raise error
        _lib.check(rc, what)

    def _apply_options(self):
        for option, value in ((_lib.GVX_SEED, self.seed), (_lib.GVX_DEVICE_SAMPLING, int(self.device_sampling)),
                              (_lib.GVX_PAIR_ORDER, _PAIR_ORDERS[self._pair_order_request]),
                              (_lib.GVX_NEGATIVE_TABLE, _NEGATIVE_TABLES[self.negative_table]),
                              (_lib.GVX_NODE2VEC_TABLE_LIMIT, int(self.node2vec_table_limit)),
                              (_lib.GVX_HUB_ROWS, int(self.hub_rows_request)), (_lib.GVX_HUB_PARTS, int(self.hub_parts)),
                              (_lib.GVX_HUB_LERP, -1 if self.hub_lerp is None else int(bool(self.hub_lerp))),
                              (_lib.GVX_HUB_ROUNDS, -1 if self.hub_rounds is None else int(bool(self.hub_rounds))),
                              (_lib.GVX_HUB_CHAIN_CAP, int(self.hub_chain_cap)),
                              (_lib.GVX_HUB_EXECUTOR, {None: -1, "fused": 0, "ahead": 1, "grouped": 2}[self.hub_executor]),
                              (_lib.GVX_HUB_GROUP, int(self.hub_group)),
                              (_lib.GVX_HUB_PAIR_LAUNCHES, int(self.hub_pair_launches)),
                              (_lib.GVX_FIDELITY, {"auto": -1, "throughput": 0, "reference": 1}[self.fidelity])):
            self._check(self._lib.gvx_solver_set(self._handle, option, value), "GraphSolver")

    def _exchange_stats(self):
        sent, count = C.c_uint64(), C.c_uint64()
        self._lib.gvx_session_exchange_stats(self._handle, C.byref(sent), C.byref(count))
        return {"bytes_sent_per_gpu": sent.value, "exchanges": count.value}

    def _refresh(self):
        """The read-only members the reference exposes (bind.h:408-436), from the engine."""
        m = _lib.SolverMembers()
        self._check(self._lib.gvx_solver_get(self._handle, C.byref(m)), "GraphSolver")
        for name in ("num_partition", "num_negative", "num_epoch", "episode_size", "batch_size", "augmentation_step",
                     "random_walk_length", "random_walk_batch_size", "shuffle_base", "positive_reuse", "log_frequency",
                     "num_worker", "num_sampler", "negative_sample_exponent", "negative_weight", "p", "q", "gpu_memory_limit",
                     "gpu_memory_cost", "batch_id", "num_batch", "rank"):
            setattr(self, name, getattr(m, name))
        self.resume = bool(m.resume)
        self.model = (m.model or b"").decode()
        self.num_local_worker = m.num_local_worker
        self.num_sampler_per_worker = m.num_sampler // max(m.num_worker, 1)
        self.pair_order = {2: "grouped", 3: "spread"}.get(m.pair_order, "sampled")
        self.partition_rows = m.partition_rows
        self.hub_rows = m.hub_rows
        self.hub_parts_used, self.hub_lerp_used, self.hub_rounds_used = m.hub_parts, bool(m.hub_lerp), bool(m.hub_rounds)
        self.lists_prefetched = m.lists_prefetched
        self.transport = (m.transport or b"").decode()
        self.train_seconds = m.train_seconds
        self._mode = _MODES.get(m.sampler_mode, "edge")

    # names kept from the round-2 solver for callers that looked inside it
    _part_size = property(lambda self: self.partition_rows)
    _sampler = property(lambda self: None if self.device_sampling else self)

    # ------------------------------------------------------------------------------------------------ build
    def build(self, graph, optimizer=auto, num_partition=auto, num_negative=1, batch_size=100000, episode_size=auto):
        """
        build(graph, optimizer=auto, num_partition=auto, num_negative=1, batch_size=100000, episode_size=auto)
        Determine and allocate all resources for the solver.
        """
        if not isinstance(graph, Graph):
            raise TypeError("graph must be a graphvite_amd.graph.Graph")
        optimizer = Optimizer(optimizer)
        o = _lib.SolverOptimizer()
        o.type = _OPTIMIZER_TYPES[optimizer.type]
        o.lr, o.weight_decay, o.epsilon = optimizer.init_lr, optimizer.weight_decay, optimizer.epsilon
        o.hp0 = {"Momentum": optimizer.momentum, "RMSprop": optimizer.alpha, "Adam": optimizer.beta1}.get(optimizer.type, 0.0)
        o.hp1 = optimizer.beta2
        o.schedule = {"constant": 0, "linear": 1}.get(optimizer.schedule.type, 2)
        if o.schedule == 2:
            function = optimizer.schedule.schedule_function

            def call(batch_id, num_batch, user):  # on the engine's thread, once per batch (optimizer.h:132-134)
                try:
                    return float(function(batch_id, num_batch))
                except BaseException as e:  # re-raised by train(); the run continues at the floor rate meanwhile
                    self._schedule_error = self._schedule_error or e
                    return 1e-4
            self._schedule_callback = _lib.SCHEDULE_FUNCTION(call)
            o.schedule_function = self._schedule_callback
        self._apply_options()
        self._check(self._lib.gvx_solver_build(self._handle, graph._handle, C.byref(o), int(num_partition), int(num_negative),
                                               int(batch_size), int(episode_size)), "GraphSolver.build")
        if optimizer.type == "Default":  # what build() resolved `auto` to (solver.h:290-296; graph.cuh:634-636)
            lr = optimizer.init_lr if optimizer.init_lr > 0 else 0.025
            optimizer = SGD(lr, 5e-3, "linear")
        self.graph, self.optimizer = graph, optimizer
        self.num_vertex, self.num_edge = graph.num_vertex, graph.num_edge
        self.num_moment = optimizer.num_moment
        self._refresh()

    def _embeddings(self, which):
        rows = C.c_uint64()
        data = self._lib.gvx_solver_embeddings(self._handle, which, C.byref(rows))
        if not data or self.graph is None:
            return None
        # the array's base chain ends at this ctypes buffer, which keeps the solver — the owner of the memory — alive for as long
        # as any view of it exists (the reference's binding ties the array's lifetime to the solver the same way, bind.h:90-106).
        # A later build() re-allocates the tables: views taken before it must not be used afterwards.
        buffer = (C.c_float * (rows.value * self.dim)).from_address(data)
        buffer._owner = self
        return np.frombuffer(buffer, dtype=np.float32).reshape(rows.value, self.dim)

    @property
    def vertex_embeddings(self):
        """Writable numpy view of the engine's stable host table (bind.h:90-106)."""
        return self._embeddings(0)

    @property
    def context_embeddings(self):
        return self._embeddings(1)

    # ------------------------------------------------------------------------------------------------ info
    def info(self):
        n = self._lib.gvx_solver_info(self._handle, None, 0)
        text = C.create_string_buffer(n + 1)
        self._lib.gvx_solver_info(self._handle, text, len(text))
        return text.value.decode()

    __repr__ = info

    # ------------------------------------------------------------------------------------------------ train
    def train(self, model="LINE", num_epoch=2000, resume=False, augmentation_step=auto, random_walk_length=40,
              random_walk_batch_size=100, shuffle_base=auto, p=1, q=1, positive_reuse=1,
              negative_sample_exponent=0.75, negative_weight=5, log_frequency=1000):
        """
        train(model='LINE', num_epoch=2000, resume=False, augmentation_step=auto, random_walk_length=40,
              random_walk_batch_size=100, shuffle_base=auto, p=1, q=1, positive_reuse=1,
              negative_sample_exponent=0.75, negative_weight=5, log_frequency=1000)
        Train node embeddings.
        """
        config = _train_config(dict(model=model, num_epoch=num_epoch, resume=resume, augmentation_step=augmentation_step,
                                    random_walk_length=random_walk_length, random_walk_batch_size=random_walk_batch_size,
                                    shuffle_base=shuffle_base, p=p, q=q, positive_reuse=positive_reuse,
                                    negative_sample_exponent=negative_sample_exponent, negative_weight=negative_weight,
                                    log_frequency=log_frequency))
        if self.graph is None:
            raise RuntimeError("The model must be built on a graph first")
        self._apply_options()
        first = self.batch_id if resume else 0
        self._schedule_error = None
        rc = self._lib.gvx_solver_train(self._handle, C.byref(config))
        if self._schedule_error is not None:
            error, self._schedule_error = self._schedule_error, None
            raise error
        self._check(rc, "GraphSolver.train")
        self._refresh()
        self.exchange_stats = self._exchange_stats()
        self.timing = {"batches": self.batch_id - first, "episodes": self.train_seconds}

    def session(self, resident_pools=False, **train_kwargs):
        """Configure a training run (same keyword arguments as train()), move the tables to HBM and return the
        TrainingSession that drives it step by step."""
        if self.graph is None:
            raise RuntimeError("The model must be built on a graph first")
        return TrainingSession(self, resident_pools=resident_pools, **train_kwargs)

    # ------------------------------------------------------------------------------------------------ predict
    def predict(self, samples):
        """
        predict(samples)
        Predict logits for samples.

        Parameters:
            samples (ndarray): pairs with shape (?, 2), each pair is ordered as (v, c)
        """
        samples = np.asarray(samples)
        if samples.ndim != 2 or samples.shape[1] != 2:
            raise ValueError("Expect an array with shape (?, 2), but shape %s is found" % (samples.shape,))
        if self.graph is None:
            raise RuntimeError("The model must be built on a graph first")
        pairs = np.ascontiguousarray(samples, np.int64)
        logits = np.zeros(len(pairs), np.float32)
        self._check(self._lib.gvx_solver_predict(self._handle, pairs.ctypes.data, len(pairs), logits.ctypes.data),
                    "GraphSolver.predict")
        return logits

    def save_embeddings(self, file_name):
        """Save vertex embeddings in word2vec binary format: "N dim\\n", then per node its name, a space, dim raw
        float32 values and a newline (GraphSolver::save_embeddings, graph.cuh:796-805; unbound in the reference)."""
        if self.graph is None:
            raise RuntimeError("The model must be built on a graph first")
        self._check(self._lib.gvx_solver_save_embeddings(self._handle, str(file_name).encode()), "save_embeddings")

    def clear(self):
        """Free CPU and GPU memory, except the embeddings on CPU."""
        self._check(self._lib.gvx_solver_clear(self._handle), "GraphSolver.clear")
