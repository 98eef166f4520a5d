"""graphvite_amd.solver — `GraphSolver`, the drop-in for graphvite.solver.GraphSolver
(pyGraphSolver, include/bind.h:383-513; GraphSolver / SolverMixin / WorkerMixin, include/instance/graph.cuh:586-813,
include/core/solver.h:87-888,1170-1623), redesigned for MI355X:

  * one process per GPU.  A process IS one of the reference's workers; `torch.distributed` (RCCL over xGMI)
    replaces the host-memory hub the reference moves partitions through (solver.h:1349-1428).
  * 288 GB of HBM per GPU: every GPU keeps the WHOLE vertex table ([P][S][dim], partition-major) and the
    context shard(s) of the tail partition(s) it owns for good, together with their negative alias tables.
    Nothing is evicted, reloaded or rebuilt between schedule steps.
  * a schedule step = every GPU trains its (head partition, tail partition) block from a sample pool that was
    uploaded in one piece (no per-batch H2D), negatives drawn inside the kernel, lr applied per batch; then ONE
    collective: all-gather of the head shards just trained.  The reference's per-step D2H + CPU scatter +
    CPU gather + H2D is gone.
  * CPU samplers (native threads, include/gvs.h) fill the next episode's pools while the GPU trains this one.

Only torch tensors (device memory, streams) and torch.distributed are used from torch; all arithmetic is in
libgvk.so.  There is no CPU training path: without a GPU, GraphSolver raises.
"""
import logging
import math
import os
import threading

import numpy as np
import torch

from . import _lib, hostlib
from ._lib import profiler_range
from .base import MiB, auto, cpu_budget, dtype, io, logger
from .graph import Graph
from .optimizer import SGD, Optimizer

kMaxPartition = 16          # solver.h:51-57
kMinBatchSize = int(1e4)
kMaxNegativeWeight = 10
kSamplePerVertex = 175
kMinEpisodeSample = int(2e7)
kExpectedDegree = 1600      # graph.cuh:55


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


class TrainingSession(object):
    """The pieces `GraphSolver.train()` is made of, exposed step by step (benchmarks, custom loops).

        session = solver.session(model="LINE", num_epoch=10, augmentation_step=1)
        pools = session.new_host_pools()          # pinned host pools, one per block this GPU trains
        session.fill(pools)                       # native CPU samplers
        resident = session.upload(pools)          # -> HBM
        for step, (hp, tp) in enumerate(session.blocks):
            session.train_block(hp, tp, resident[(hp, tp)])
            session.exchange(step)                # all-gather of the head shards (no-op on one GPU)
        session.finish()                          # device tables -> solver.vertex_embeddings / context_embeddings
    """

    def __init__(self, solver, **train_kwargs):
        defaults = dict(model="LINE", num_epoch=2000, resume=False, augmentation_step=auto, random_walk_length=40,
                        random_walk_batch_size=100, shuffle_base=auto, p=1, q=1, positive_reuse=1,
                        negative_sample_exponent=0.75, negative_weight=5, log_frequency=1000)
        unknown = set(train_kwargs) - set(defaults)
        if unknown:
            raise TypeError("unexpected training argument(s): %s" % ", ".join(sorted(unknown)))
        defaults.update(train_kwargs)
        self.solver = solver
        solver._configure_training(**defaults)
        self.state = solver._upload_state()
        #: (head partition, tail partition) this GPU trains at each schedule step of an episode
        self.blocks = [(int(step[solver.rank][0]), int(step[solver.rank][1])) for step in solver._schedule]

    def new_host_pools(self, sets=1):
        pools = self.solver._host_pools(sets)
        return pools[0] if sets == 1 else pools

    def fill(self, pools):
        self.solver._fill(pools)

    def upload(self, pools, group=True):
        """Host pools -> device pools ready for train_block() (regrouped when the solver's pair_order says so;
        group=False leaves that to stage())."""
        out = {}
        for block, pool in pools.items():
            landed = pool.to(self.solver.device)
            if group and self.solver.pair_order == "grouped":
                out[block] = torch.empty_like(landed)
                self.solver._group_pairs(landed, out[block])
            else:
                out[block] = landed
        return out

    def stage(self, landed, out, num_batches=None):
        """What the episode loop does to a pool after its H2D copy: with pair_order "grouped", regroup (the first
        `num_batches` batches of) `landed` into `out` on the current stream and return `out`; otherwise return
        `landed` untouched."""
        if self.solver.pair_order != "grouped":
            return landed
        self.solver._group_pairs(landed, out, num_batches)
        return out

    def train_block(self, hp, tp, pool, num_batches=None):
        """Train `num_batches` (default: episode_size) batches of block (hp, tp) from a device-resident pool."""
        solver = self.solver
        saved = solver.episode_size
        if num_batches is not None:
            solver.episode_size = int(num_batches)
        try:
            solver._train_block(self.state, hp, tp, pool)
        finally:
            solver.episode_size = saved

    def exchange(self, step_index):
        """Start the (asynchronous) all-gather of the head shards trained at this schedule step."""
        if self.solver.num_worker > 1:
            self.solver._exchange(self.state, step_index % len(self.blocks))

    def wait_exchange(self, hp=None):
        """Fence the compute stream behind the pending all-gather of head partition hp's group (all groups if None);
        train_block() does this itself — call it first only to keep the wait out of a timed region."""
        if self.solver.num_worker > 1:
            self.solver._wait_exchange(self.state, None if hp is None else hp // self.solver.num_worker)

    @property
    def loss(self):
        """Per-sample loss of the most recent batch (device tensor)."""
        return self.state["loss"]

    def negative_table(self, tail_partition):
        return self.state["negative_tables"][tail_partition]

    def finish(self):
        self.solver._write_back(self.state)
        self.state = None


class GraphSolver(object):
    """
    GraphSolver(dim, float_type=dtype.float32, index_type=dtype.uint32, device_ids=[], num_sampler_per_worker=auto,
                gpu_memory_limit=auto)
    Graph embedding solver.

    Parameters:
        dim (int): dimension of embeddings (32, 64, 96, 128, 256 or 512)
        float_type (dtype): type of parameters (float32)
        index_type (dtype): type of node indexes (uint32)
        device_ids (list of int, optional): GPU ids, [] for auto.  One process drives ONE GPU; for several GPUs
            launch one process per GPU (`python -m torch.distributed.run --nproc-per-node N ...`) — every
            process constructs the same solver and `device_ids` then lists the GPUs of the whole job.
        num_sampler_per_worker (int, optional): number of sampler threads per GPU
        gpu_memory_limit (int, optional): memory limit for each GPU in bytes

    Beyond the reference's arguments: `kernels` (test seam), `seed`, `device_sampling` (draw the positive samples on
    the GPU) and `pair_order` — "sampled": a batch is trained in the order the samplers produced it; "grouped": the
    pairs of a batch that share a head row are made adjacent on the device first (gvk_group_pairs; same samples, same
    batches; a row shared by k samples is fetched from HBM once and the k samples are trained as runs, one after the
    other on one copy of the row when the table is small); auto (default), at dim >= 64: grouped when a partition's
    table is cache-resident (< 16 MiB: every batch hits the hub rows hundreds of times and the runs keep training
    close to sequential), grouped for independent edge draws up to 256 MiB (shards of multi-GPU runs: a shared head row
    becomes one fetch), the sampler's order otherwise.
    """

    available_dims = (32, 64, 96, 128, 256, 512)  # src/graphvite.cu:52-59
    available_models = ("DeepWalk", "LINE", "node2vec")

    def __init__(self, dim, float_type=dtype.float32, index_type=dtype.uint32, device_ids=(),
                 num_sampler_per_worker=auto, gpu_memory_limit=auto, kernels=None, seed=0, device_sampling=False,
                 pair_order=auto):
        if dim not in self.available_dims or float_type != dtype.float32 or index_type != dtype.uint32:
            raise AttributeError("Can't find an instantiation of GraphSolver with dim=%s, float_type=%s, "
                                 "index_type=%s" % (dim, float_type, index_type))
        self.dim = dim
        dist = _dist()
        self.num_worker = dist.get_world_size() if dist else 1
        self.rank = dist.get_rank() if dist else 0
        device_ids = list(device_ids)
        if kernels is None:
            if not torch.cuda.is_available():
                raise RuntimeError("No GPU devices found (graphvite_amd has no CPU training path)")
            from .kernels import HipKernels
            kernels = HipKernels()
            if device_ids and len(device_ids) != self.num_worker:
                raise ValueError("%d GPUs requested but this job has %d process(es): graphvite_amd runs one process "
                                 "per GPU — launch with `python -m torch.distributed.run --nproc-per-node %d`"
                                 % (len(device_ids), self.num_worker, len(device_ids)))
            local = int(os.environ.get("LOCAL_RANK", self.rank if dist else 0))
            index = device_ids[self.rank] if device_ids else (local if dist else torch.cuda.current_device())
            self.device = torch.device("cuda", index)
        else:
            self.device = torch.device(getattr(kernels, "device", "cpu"))
        self.kernels = kernels
        if num_sampler_per_worker == auto:
            # the reference takes hardware_concurrency / #GPU - 1 (solver.h:193-194); the usable CPUs are what counts
            num_sampler_per_worker = max(cpu_budget() // max(self._local_world(), 1) - 1, 1)
        self.num_sampler_per_worker = int(num_sampler_per_worker)
        self.num_sampler = self.num_sampler_per_worker * self.num_worker
        self._gpu_memory_request = gpu_memory_limit
        self.gpu_memory_limit = gpu_memory_limit
        self.gpu_memory_cost = 0
        self._upload_chunk_bytes = 256 << 20  # host <-> device table traffic goes through chunks of this size
        self.seed = seed
        self.node2vec_table_limit = 1 << 30  # entries (8 B each) of per-edge alias tables before switching to rejection
        # the negative sampler's table: "rows" = one alias slot per row of the tail partition (the reference's), "classes"
        # = an alias table over the classes of equal-weight rows (same distribution, cache-resident), "auto" = classes
        # when they are at least 8 times fewer than the rows (unweighted graphs: always)
        self.negative_table = "auto"
        # extension: draw LINE's positive edge samples on the GPU instead of CPU threads
        self.device_sampling = bool(device_sampling)
        if pair_order not in (auto, "sampled", "grouped"):
            raise ValueError("pair_order must be auto, 'sampled' or 'grouped', not %r" % (pair_order,))
        self._pair_order_request = pair_order
        self.pair_order = "sampled" if pair_order == auto else pair_order
        self.graph = None
        self.batch_id = 0
        self._sampler = None
        self._device_state = None
        self._predict_cache = None
        self.vertex_embeddings = None
        self.context_embeddings = None
        # attributes the reference exposes read-only (bind.h:415-436)
        self.num_partition = self.num_negative = self.episode_size = self.batch_size = 0
        self.optimizer = None
        self.negative_sample_exponent = self.negative_weight = 0.0
        self.model = ""
        self.num_epoch = 0
        self.resume = False
        self.augmentation_step = self.random_walk_length = self.random_walk_batch_size = self.shuffle_base = 0
        self.p = self.q = 1.0
        self.positive_reuse = 1
        self.log_frequency = 1000

    @staticmethod
    def _local_world():
        return int(os.environ.get("LOCAL_WORLD_SIZE", "1"))

    # ------------------------------------------------------------------------------------------------ build
    def build(self, graph, optimizer=auto, num_partition=auto, num_negative=1, batch_size=100000, episode_size=auto):
        """
        build(graph, optimizer=auto, num_partition=auto, num_negative=1, batch_size=100000, episode_size=auto)
        Determine and allocate all resources for the solver.
        """
        if not isinstance(graph, Graph):
            raise TypeError("graph must be a graphvite_amd.graph.Graph")
        if graph.num_vertex == 0 or graph.num_directed_edge == 0:
            raise ValueError("The graph is empty")
        self.clear()
        self.graph = graph
        optimizer = Optimizer(optimizer)
        if optimizer.type == "Default":  # solver.h:290-296
            default = SGD(0.025, 5e-3)   # GraphSolver::get_default_optimizer, graph.cuh:634-636
            if optimizer.init_lr > 0:
                default.init_lr = default.lr = optimizer.init_lr
            optimizer = default
        self.optimizer = optimizer
        self.num_vertex, self.num_edge = graph.num_vertex, graph.num_edge
        self.num_moment = optimizer.num_moment
        self.num_negative, self.batch_size = int(num_negative), int(batch_size)
        if self.batch_size < 1 or self.num_negative < 0:
            raise ValueError("batch_size must be positive and num_negative non-negative")
        if self.batch_size < kMinBatchSize:
            logger.warning("It is recommended to a minimum batch size of %d, but %d is specified",
                           kMinBatchSize, self.batch_size)
        self.batch_id = 0
        W = self.num_worker
        min_partition = W  # get_min_partition, non-tied (solver.h:269-276)
        limit = self._gpu_memory_request  # what the user asked for; `auto` is resolved anew at every build
        if limit == auto:
            limit = torch.cuda.mem_get_info(self.device)[0] if self.device.type == "cuda" else 1 << 62
        self.episode_size = 0  # nothing of a previous build enters the estimates below
        if num_partition == auto:
            num_partition = min_partition
            while num_partition < kMaxPartition and self._memory_demand(num_partition, episode_size) >= limit:
                num_partition += min_partition
        else:
            if num_partition < min_partition:
                raise ValueError("#partition should be no less than %d" % min_partition)
            if num_partition % W:
                raise ValueError("#partition (%d) must be a multiple of #worker (%d)" % (num_partition, W))
            if num_partition > kMaxPartition:
                logger.warning("It is recommended to use a maximum #partition of %d, but %d partitions are specified",
                               kMaxPartition, num_partition)
        self.num_partition = P = int(num_partition)
        self.gpu_memory_limit = limit
        self.gpu_memory_cost = self._memory_demand(P, episode_size)
        if self.gpu_memory_cost >= limit:
            raise MemoryError("Can't satisfy the specified GPU memory limit")

        # partitions (heads and tails are the same partition, solver.h:389-390)
        self._part, self._local, self._part_sizes = hostlib.partition(graph.vertex_weights, P)
        self._part_size = int(self._part_sizes.max())
        order = np.argsort(self._part.astype(np.int64) * (1 << 32) + self._local, kind="stable")
        starts = np.concatenate([[0], np.cumsum(self._part_sizes.astype(np.int64))]).astype(np.int64)
        self._part_ids = [order[starts[p]:starts[p + 1]] for p in range(P)]  # global ids in local order
        # device tables are partition-major [P][S][dim]: slot p * S + local(v) holds vertex v
        S = self._part_size
        self._row_of_vertex = self._part.astype(np.int64) * S + self._local.astype(np.int64)
        self._vertex_of_row = np.zeros(P * S, np.int64)  # padding slots point at vertex 0 and are never trained
        self._vertex_of_row[self._row_of_vertex] = np.arange(self.num_vertex, dtype=np.int64)
        self._schedule = self._overlap_order(hostlib.schedule(P, W), P, W)
        self._my_tails = sorted({int(step[self.rank][1]) for step in self._schedule})
        self._step_heads = None

        if episode_size == auto:  # solver.h:426-436
            expected = int(float(self.num_vertex) * kSamplePerVertex / P / self.batch_size)
            expected = max(expected, 1)
            if P == 1:
                expected = max(expected, kMinEpisodeSample // self.batch_size)
            episode_size = expected
        self.episode_size = int(episode_size)
        if self.episode_size < 1:
            raise ValueError("episode_size must be positive")

        # host embeddings: stable buffers, exposed as writable numpy views (bind.h:90-106)
        self.vertex_embeddings = np.zeros((self.num_vertex, self.dim), np.float32)
        self.context_embeddings = np.zeros((self.num_vertex, self.dim), np.float32)
        self._moments_host = None
        self._sampler = None  # the CPU sampler (edge alias table over all edges) is built at the first train()
        self._positive_index = 0  # ... and the device sampler's stream starts over
        self._sampler_mode = None

    @staticmethod
    def _overlap_order(schedule, P, W):
        """The reference walks the block groups x-major (solver.h:562-574): all steps that use head partitions
        x .. x + W - 1 come back to back, and each needs the exchange of the one before.  An episode may visit its P^2
        blocks in any order, so with P = m * W (m > 1) the steps are interleaved across the m head groups: the
        all-gather of group x's shards then runs on the collective stream while the next m - 1 steps train on the
        other groups.  With P == W there is a single group and the order is the reference's."""
        m = P // W if P > 1 else 1
        if m <= 1:
            return schedule
        order = [(xi * m + yi) * W + o for yi in range(m) for o in range(W) for xi in range(m)]
        return schedule[order]

    def _memory_demand(self, P, episode_size=auto):
        """Bytes of HBM this design keeps resident per GPU with P partitions (episode_size: the requested one)."""
        S = (self.num_vertex + P - 1) // P
        tails = max(P // self.num_worker, 1)
        rows = P * S + tails * S
        demand = rows * self.dim * 4 * (1 + self.num_moment)
        demand += tails * S * 8                                   # negative alias tables
        demand += self.batch_size * 4                             # loss
        if episode_size == auto:
            episode_size = max(int(float(self.num_vertex) * kSamplePerVertex / P / self.batch_size), 1)
            if P == 1:
                episode_size = max(episode_size, kMinEpisodeSample // self.batch_size)
        demand += 3 * int(episode_size) * self.batch_size * 8     # two device pool buffers + the regrouping landing buffer
        demand += 2 * self._upload_chunk_bytes                    # upload / write-back transient: one chunk + its row ids
        return demand

    # ------------------------------------------------------------------------------------------------ info
    def info(self):
        lines = ["GraphSolver<%d, float32, uint32>" % self.dim, io.header("Resource"),
                 "#worker: %d, #sampler: %d, #partition: %d" % (self.num_worker, self.num_sampler, self.num_partition),
                 "tied weights: no, episode size: %d" % self.episode_size,
                 "gpu memory limit: %s" % io.size_string(self.gpu_memory_limit if self.gpu_memory_limit else 0),
                 "gpu memory cost: %s" % io.size_string(self.gpu_memory_cost), io.header("Sampling")]
        if self.model == "LINE":
            lines.append("augmentation step: %d, shuffle base: %d" % (self.augmentation_step, self.shuffle_base))
        if self.model == "DeepWalk":
            lines.append("augmentation step: %d" % self.augmentation_step)
        if self.model == "node2vec":
            lines.append("augmentation step: %d, p: %g, q: %g" % (self.augmentation_step, self.p, self.q))
        lines += ["random walk length: %d" % self.random_walk_length,
                  "random walk batch size: %d" % self.random_walk_batch_size,
                  "#negative: %d, negative sample exponent: %g" % (self.num_negative, self.negative_sample_exponent),
                  io.header("Training"), "model: %s" % self.model,
                  self.optimizer.info() if self.optimizer else "optimizer: -",
                  "#epoch: %d, batch size: %d" % (self.num_epoch, self.batch_size),
                  "resume: %s" % io.yes_no(self.resume),
                  "positive reuse: %d, negative weight: %g" % (self.positive_reuse, self.negative_weight)]
        return "\n".join(lines)

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
        import time
        t0 = time.time()
        self._configure_training(model, num_epoch, resume, augmentation_step, random_walk_length,
                                 random_walk_batch_size, shuffle_base, p, q, positive_reuse,
                                 negative_sample_exponent, negative_weight, log_frequency)
        t1 = time.time()
        state = self._upload_state()
        if self.device_sampling:
            if self._mode == "edge":
                self._upload_block_tables(state)
            else:
                self._upload_walk_graph(state)
        t2 = time.time()
        first_batch = self.batch_id

        def report(failed=False):
            """Write-back and timing.  On the error path (an exception is propagating on this rank while its peers may be
            inside training collectives) nothing collective is issued: local shards only, then the error is re-raised."""
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            t3 = time.time()
            self._write_back(state, collective=not failed)
            t4 = time.time()
            self.timing = {"configure": t1 - t0, "upload": t2 - t1, "episodes": t3 - t2, "write_back": t4 - t3,
                           "batches": self.batch_id - first_batch, "loop": getattr(self, "_loop_timing", None)}
            logger.info("[time] configure %.2f s, upload %.2f s, %d batches in %.2f s (%.1f M edge-samples/s), "
                        "write back %.2f s", t1 - t0, t2 - t1, self.batch_id - first_batch, t3 - t2,
                        (self.batch_id - first_batch) * self.batch_size / max(t3 - t2, 1e-9) / 1e6, t4 - t3)

        def guarded(loop):
            try:
                loop()
            except BaseException:
                try:
                    report(failed=True)
                except Exception:  # the original error is the one to surface
                    pass
                raise
            report()

        walks_over_blocks = self._mode != "edge" and self.num_partition > 1
        if self.device_sampling and not walks_over_blocks:
            def loop():
                while self.batch_id < self.num_batch:
                    self._train_episode_device_sampling(state)
            return guarded(loop)
        if walks_over_blocks and (self.num_worker > 1 or self.device_sampling):
            return guarded(lambda: self._train_routed(state))
        per_episode = len(self._schedule) * self.episode_size * self.positive_reuse * self.num_worker
        pools = self._host_pools()
        uploads = [[], []]  # per pool set: events of the async H2D copies still reading its pinned buffers

        def loop():
            self._fill(pools[0])
            current = 0
            self._loop_timing = {"wait_upload": 0.0, "enqueue": 0.0, "wait_fill": 0.0, "fill": 0.0}
            while self.batch_id < self.num_batch:  # solver.h:629-649 — one iteration = one episode
                # the samplers may only overwrite a pool set once the GPU has finished copying it out; this also
                # keeps the host at most one episode ahead of the device (producer / consumer, double buffer)
                ta = time.time()
                for event in uploads[current ^ 1]:
                    event.synchronize()
                uploads[current ^ 1] = []
                tb = time.time()
                self._fill_error = None
                filler = None
                if self.batch_id + per_episode < self.num_batch:  # no pools for an episode that will not run
                    filler = threading.Thread(target=self._fill_guarded, args=(pools[current ^ 1],))
                    filler.start()
                try:
                    uploads[current] = self._train_episode(state, pools[current])
                finally:
                    tc = time.time()
                    if filler is not None:
                        with profiler_range("Wait for sample threads"):  # solver.h:645
                            filler.join()
                td = time.time()
                self._loop_timing["wait_upload"] += tb - ta
                self._loop_timing["enqueue"] += tc - tb
                self._loop_timing["wait_fill"] += td - tc
                if self._fill_error is not None:
                    raise self._fill_error
                current ^= 1
        guarded(loop)

    def session(self, **train_kwargs):
        """Configure a training run (same keyword arguments as train()), move the tables to HBM and return the
        TrainingSession that drives it block by block."""
        return TrainingSession(self, **train_kwargs)

    def _configure_training(self, model, num_epoch, resume, augmentation_step, random_walk_length,
                            random_walk_batch_size, shuffle_base, p, q, positive_reuse, negative_sample_exponent,
                            negative_weight, log_frequency):
        """Argument checks, auto hyper-parameters, embedding init and sampler tables: everything GraphSolver::train
        and SolverMixin::train do before the episode loop (graph.cuh:770-793, solver.h:588-628)."""
        if self.graph is None:
            raise RuntimeError("The model must be built on a graph first")
        if model not in self.available_models:
            raise ValueError("Invalid model `%s`" % model)
        if augmentation_step == auto:  # graph.cuh:781-784
            density = math.log(float(self.num_edge) / self.num_vertex)
            # as many edges as vertices: the reference divides by log(1) = 0 and fails the checks below on the result
            augmentation_step = int(math.log(kExpectedDegree) / density) if density else random_walk_length + 1
        if shuffle_base == auto:
            shuffle_base = augmentation_step
        if model in ("DeepWalk", "node2vec"):
            shuffle_base = 1
        if augmentation_step < 1:
            raise ValueError("`augmentation_step` should be a positive integer")
        if augmentation_step > random_walk_length:
            raise ValueError("`random_walk_length` should be no less than `augmentation_step`")
        if positive_reuse < 1 or log_frequency < 1 or num_epoch < 0:
            raise ValueError("positive_reuse / log_frequency must be positive and num_epoch non-negative")
        if negative_weight > kMaxNegativeWeight:
            logger.warning("It is recommended to a maximum negative weight of %d, but %g is specified",
                           kMaxNegativeWeight, negative_weight)
        self.model, self.num_epoch, self.resume = model, int(num_epoch), bool(resume)
        self.augmentation_step, self.shuffle_base = int(augmentation_step), int(shuffle_base)
        self.random_walk_length, self.random_walk_batch_size = int(random_walk_length), int(random_walk_batch_size)
        self.p, self.q = float(p), float(q)
        self.positive_reuse = int(positive_reuse)
        self.negative_sample_exponent, self.negative_weight = float(negative_sample_exponent), float(negative_weight)
        self.log_frequency = int(log_frequency)
        self.sample_batch_size = self.random_walk_length * self.random_walk_batch_size  # graph.cuh:791
        pool_size = self.episode_size * self.batch_size
        if self.augmentation_step > 1 and pool_size % self.shuffle_base:
            raise ValueError("Can't perform pseudo shuffle on %d elements by a shuffle base of %d. Try setting the "
                             "episode size to a multiple of the shuffle base" % (pool_size, self.shuffle_base))

        logger.warning(io.block(self.info()))
        if not self.resume:
            self._init_embeddings()
            self.batch_id = 0
        self.num_batch = self.batch_id + self.num_epoch * self.num_edge // self.batch_size  # solver.h:611
        self._predict_cache = None

        mode = "edge" if self.augmentation_step == 1 else ("biased_walk" if model == "node2vec" else "walk")
        if mode == "biased_walk":
            # the reference's per-edge alias tables need sum over edges of deg(head) entries (graph.cuh:656-677) and
            # run it out of memory on hub-heavy graphs (doc/source/benchmark.rst:53-54); past the limit the same
            # transition distribution is sampled by rejection over the per-vertex tables (gvs.h GVS_MODE_BIASED_REJECT)
            degree = np.diff(self.graph.flat_offsets.astype(np.int64))
            entries = int(degree[self.graph.edges[:, 1]].sum())
            if entries > self.node2vec_table_limit:
                logger.warning("node2vec: %d per-edge table entries exceed the limit of %d; sampling by rejection",
                               entries, self.node2vec_table_limit)
                mode = "biased_reject"
        self._mode = mode
        if self._pair_order_request == auto:
            # Regroup (gvk_group_pairs) by the size of a partition's table, at dim >= 64:
            #   < 16 MiB   (a BlogCatalog-sized graph) every batch hits every hub row hundreds of times; with same-head
            #              samples adjacent the kernel trains them as runs of up to 20 consecutive updates on one copy of the row,
            #              which keeps link-prediction AUC within 0.002 of sequential training; any sampler;
            #   < 256 MiB  (the shards of multi-GPU runs) tables live in L2 / Infinity Cache: adjacent same-head samples
            #              make a head row one fetch (2.3 -> 2.9 G edge-samples/s per GPU on 32 MB shards), AUC unchanged;
            #              independent edge draws only — random-walk pools come in the reference's pseudo-shuffled walk
            #              order, which already has locality (DeepWalk end to end -16 % when regrouped);
            #   larger     the sampler's order, as the reference: the gain is within run-to-run noise and the pass is
            #              not free (-12 % on a Friendster shard).
            # Not at dim 32 (a batch trains in 16 us there; the pass costs the same at every dim).
            table = self._part_size * self.dim * 4
            regroup = table < MiB(16) or (table < MiB(256) and mode == "edge")
            self.pair_order = "grouped" if regroup and self.dim >= 64 and self.device.type == "cuda" else "sampled"
        if self.device_sampling:
            return  # positives are drawn on the device: no CPU sampler needed
        if self._sampler is None:
            self._sampler = hostlib.Sampler(self.graph, self._part, self._local, self.num_partition,
                                            (self.seed + 0x9E3779B97F4A7C15 * (self.rank + 1)) & (2 ** 64 - 1))
        key = (mode, self.p, self.q)
        if self._sampler_mode != key:  # get_sample_function, graph.cuh:680-721
            self._sampler.prepare(mode, self.p, self.q, self.num_sampler_per_worker + 1)
            if mode == "edge" and self.num_partition > 1:
                for tp in self._my_tails:
                    self._sampler.prepare_column(tp, self.num_sampler_per_worker + 1)
            self._sampler_mode = key
        self._mode = mode


    # ---- embeddings -------------------------------------------------------------------------------------
    def _init_embeddings(self):
        """vertex ~ U(-0.5/dim, 0.5/dim), context = 0 (GraphSolver::init_embeddings, graph.cuh:724-731).
        The generator is seeded identically on every process so all ranks start from the same table."""
        rng = np.random.default_rng(self.seed + 5489)
        rng.random(out=self.vertex_embeddings, dtype=np.float32)     # U[0, 1) straight into the stable buffer
        self.vertex_embeddings -= np.float32(0.5)
        self.vertex_embeddings *= np.float32(1.0 / self.dim)
        self.context_embeddings[:] = 0
        self._moments_host = None

    def _to_device(self, array):
        t = torch.from_numpy(np.ascontiguousarray(array))
        return t.to(self.device, non_blocking=False)

    def _upload_state(self):
        """Partition-major device tables.  state["head"]: [P slots][1 + m][S][dim] — a slot holds one head partition's
        vertex rows followed by its m moment tables, so that the W shards a schedule step trains form ONE contiguous
        slab per head group and the exchange is a single in-place all-gather (`_exchange`); state["slot_of"][hp] says
        where partition hp lives (identity on one GPU).  state["context"] (+ moments): [S][dim] per owned tail."""
        P, S, dim = self.num_partition, self._part_size, self.dim
        nm = self.num_moment
        chunk_rows = max(self._upload_chunk_bytes // (dim * 4), 1)

        def scatter_in(dest, host, parts, tables=1, table=0):
            """host [N][dim] (global ids) -> dest [len(parts) slots][tables][S][dim], table `table`, for the vertices whose
            partition is in `parts` (slot = position in `parts`).  The host table is streamed once, in contiguous chunks
            of at most 256 MiB; the permutation runs on the device, so the transient is one chunk, not a second table."""
            slot = np.full(P, -1, np.int64)
            slot[list(parts)] = np.arange(len(parts))
            flat = dest.view(-1, dim)
            for start in range(0, self.num_vertex, chunk_rows):
                stop = min(start + chunk_rows, self.num_vertex)
                where = slot[self._part[start:stop]]
                keep = where >= 0
                if not keep.any():
                    continue
                rows = (where[keep] * tables + table) * S + self._local[start:stop][keep].astype(np.int64)
                block = host[start:stop] if keep.all() else host[start:stop][keep]
                flat[self._to_device(rows)] = self._to_device(block)

        head = torch.zeros((P, 1 + nm, S, dim), dtype=torch.float32, device=self.device)
        context = torch.zeros((len(self._my_tails), S, dim), dtype=torch.float32, device=self.device)
        scatter_in(head, self.vertex_embeddings, range(P), 1 + nm, 0)
        scatter_in(context, self.context_embeddings, self._my_tails)
        state = {"head": head, "context": context, "slot_of": list(range(P)), "part_at": list(range(P))}
        mh = self._moments_host if self.resume and self._moments_host is not None else None
        for j in range(nm):
            state["context_m%d" % j] = torch.zeros((len(self._my_tails), S, dim), dtype=torch.float32, device=self.device)
            if mh:
                scatter_in(head, mh["vertex"][j], range(P), 1 + nm, 1 + j)
                scatter_in(state["context_m%d" % j], mh["context"][j], self._my_tails)
        # negative sampler per owned tail partition: deg^exponent in local order (solver.h:1264-1278)
        from .kernels import alias_build, class_table_build, classes_to_device, packed_to_device
        weights = self.graph.vertex_weights
        state["negative_tables"] = {}
        for tp in self._my_tails:
            w = hostlib.negative_weights(weights, self._part_ids[tp], self.negative_sample_exponent)
            # rows of equal weight (= equal degree: the partition is sorted by it) form a class; a few thousand classes
            # instead of one slot per row keep the sampler's table in the caches (gvk.h)
            classes = class_table_build(w) if self.negative_table != "rows" else None
            if classes is not None and (self.negative_table == "classes" or classes.size * 8 <= w.size):
                state["negative_tables"][tp] = classes_to_device(classes, self.device)
                continue
            _, _, packed = alias_build(w)
            state["negative_tables"][tp] = packed_to_device(packed, self.device)
        state["loss"] = torch.zeros(self.batch_size, dtype=torch.float32, device=self.device)
        # The pools are the elastic part, as in the reference: when they do not fit, the episode is halved
        # (solver.h:437-455) — the device pair of buffers here, the pinned host sets in _host_pools.
        while True:
            try:
                pool_elems = self.episode_size * self.batch_size * 2
                state["pool_dev"] = [torch.empty(pool_elems, dtype=torch.int32, device=self.device) for _ in range(2)]
                if self.pair_order == "grouped":  # uploads land here and are regrouped into pool_dev
                    state["pool_stage"] = torch.empty(pool_elems, dtype=torch.int32, device=self.device)
                break
            except (RuntimeError, MemoryError):
                state.pop("pool_dev", None)
                state.pop("pool_stage", None)
                self._halve_episode("GPU")
        if self.device.type == "cuda":
            state["copy_stream"] = torch.cuda.Stream(self.device)
        self._device_state = state
        return state

    def _halve_episode(self, where):
        if self.episode_size <= 1:
            raise MemoryError("Out of %s memory. Try to reduce the size of your graph or the dimension of your "
                              "embeddings." % where)
        base = max(self.shuffle_base, 1) if self.augmentation_step > 1 else 1
        half = self.episode_size // 2
        while half > 1 and (half * self.batch_size) % base:
            half -= 1
        logger.warning("Fail to allocate %s memory for episode size of %d. Use %d instead.", where,
                       self.episode_size, half)
        self.episode_size = max(half, 1)

    def _host_pools(self, sets=2):
        """`sets` (double buffer) sets of pinned host pools, one pool per block this worker trains in an episode."""
        blocks = sorted({(int(step[self.rank][0]), int(step[self.rank][1])) for step in self._schedule})
        pin = self.device.type == "cuda"
        while True:
            pool_elems = self.episode_size * self.batch_size * 2
            try:
                return [{b: torch.empty(pool_elems, dtype=torch.int32, pin_memory=pin) for b in blocks}
                        for _ in range(sets)]
            except (RuntimeError, MemoryError):
                self._halve_episode("host")

    # ---- sampling -----------------------------------------------------------------------------------------
    def _fill(self, pools):
        with profiler_range("Sample threads"):  # solver.h:622
            self._fill_pools(pools)

    def _fill_pools(self, pools):
        P = self.num_partition
        tails = {tp for (_, tp) in pools}
        pool_size = self.episode_size * self.batch_size
        # this worker's blocks form whole columns (hp ranges over all partitions for each owned tail)
        for tp in sorted(tails):
            column = {(hp, tp): pools[(hp, tp)] for hp in range(P)}
            # 4 slices per OS thread: a thread that gets descheduled delays a quarter-size slice, not the whole fill
            self._sampler.fill(column, pool_size, self._mode, 4 * self.num_sampler_per_worker,
                               sample_batch_size=self.sample_batch_size, walk_length=self.random_walk_length,
                               walk_batch=self.random_walk_batch_size, augmentation_step=self.augmentation_step,
                               shuffle_base=self.shuffle_base, tail_partition=tp if P > 1 else -1,
                               os_threads=self.num_sampler_per_worker)

    _fill_error = None

    def _fill_guarded(self, pools):
        import time
        self._fill_error = None
        t0 = time.time()
        try:
            self._fill(pools)
        except BaseException as e:  # surfaced on the training thread
            self._fill_error = e
        if getattr(self, "_loop_timing", None) is not None:
            self._loop_timing["fill"] += time.time() - t0

    # ---- several GPUs, random-walk models: every rank samples a slice of EVERY block, pairs are routed ------------
    def _train_routed(self, state):
        """A walk yields pairs for all P^2 blocks, so with W ranks each rank samples the 1/W-th slice of every block pool
        (no pair is thrown away for belonging to another GPU, unlike a per-column filter) and one all-to-all hands every
        block's W slices to the GPU that trains it.  The slices come from this rank's CPU samplers — pipeline per episode:
        CPU fill (e + 1) || copy stream + RCCL: H2D, all_to_all, un-interleave (e) || compute stream: train (e - 1) — or,
        with device_sampling, from gvk_sample_walks_blocks on the side stream (walks binned per block on the GPU, no host
        threads, no PCIe): sample + all_to_all (e + 1) || train (e).  Also the single-GPU path of device-sampled walks
        over several partitions (W = 1: nothing to route)."""
        import time
        import torch.distributed as dist
        W, r, P, B = self.num_worker, self.rank, self.num_partition, self.batch_size
        n = self.episode_size * B
        if n % W:
            raise ValueError("episode_size * batch_size (%d) must be a multiple of the number of GPUs (%d) for the "
                             "random-walk models" % (n, W))
        n_slice = n // W
        if self.augmentation_step > 1 and n_slice % self.shuffle_base:
            raise ValueError("Can't perform pseudo shuffle on %d elements by a shuffle base of %d" %
                             (n_slice, self.shuffle_base))
        cuda = self.device.type == "cuda"
        tails_of = [sorted({int(step[w][1]) for step in self._schedule}) for w in range(W)]
        bpr = P * len(tails_of[0])                      # blocks each rank trains
        order = [[(hp, tp) for tp in tails_of[w] for hp in range(P)] for w in range(W)]  # canonical per-owner order
        mine = {block: i for i, block in enumerate(order[r])}
        elems = n_slice * 2
        on_device = self.device_sampling
        host = None if on_device else [torch.empty((W, bpr, elems), dtype=torch.int32, pin_memory=cuda) for _ in range(2)]
        send = [torch.empty((W, bpr, elems), dtype=torch.int32, device=self.device) for _ in range(2)]
        recv = torch.empty((W, bpr, elems), dtype=torch.int32, device=self.device) if W > 1 else None
        pools = [torch.empty((bpr, W, elems), dtype=torch.int32, device=self.device) for _ in range(2)]
        landing = torch.empty_like(pools[0]) if self.pair_order == "grouped" else None  # regrouped into pools[s]
        views = None if on_device else [{order[w][i]: host[s][w, i] for w in range(W) for i in range(bpr)} for s in range(2)]
        if on_device:  # block (hp, tp) of owner w at index i starts (w * bpr + i) * n_slice pairs into send[s]
            where = np.full(P * P, -1, np.int64)
            for w in range(W):
                for i, (hp, tp) in enumerate(order[w]):
                    where[hp * P + tp] = (w * bpr + i) * n_slice
            offsets = self._to_device(where)
            walk_seed = (self.seed * 0x9E3779B1 + 0x77616c6b + r) & (2 ** 64 - 1)

        def sample(s):
            """This rank's slice of every block pool, drawn and binned on the device into send[s] (current stream)."""
            with profiler_range("Sample walks (device)"):
                used = self.kernels.sample_walks_blocks(state["walk_graph"], state["walk_graph"]["part"], P, walk_seed,
                                                        state["positive_index"], send[s].view(-1), offsets, n_slice,
                                                        self.random_walk_length, self.augmentation_step, self.shuffle_base)
            state["positive_index"] += used
        route_stream = torch.cuda.Stream(self.device) if cuda else None
        copied, routed, trained = [None, None], [None, None], [None, None]

        def fill(s):
            self._sampler.fill(views[s], n_slice, self._mode, 4 * self.num_sampler_per_worker,
                               sample_batch_size=self.sample_batch_size, walk_length=self.random_walk_length,
                               walk_batch=self.random_walk_batch_size, augmentation_step=self.augmentation_step,
                               shuffle_base=self.shuffle_base, tail_partition=-1,
                               os_threads=self.num_sampler_per_worker)

        def exchange_slices(s):
            """send[s] [owner][block][slice] -> (all_to_all) [sampler rank][block][slice]; one rank: nothing to route."""
            if W == 1:
                return send[s]
            dist.all_to_all_single(recv.view(-1), send[s].view(-1))
            return recv

        def route(s):
            """host[s] -> send[s] (or sampled straight into it) -> (all_to_all) recv -> pools[s], on the side stream."""
            if not cuda:
                sample(s) if on_device else send[s].copy_(host[s])
                (pools[s] if landing is None else landing).copy_(exchange_slices(s).permute(1, 0, 2))
                if landing is not None:
                    self._group_pairs(landing.view(-1), pools[s].view(-1))
                return
            with torch.cuda.stream(route_stream):
                if trained[s] is not None:
                    route_stream.wait_event(trained[s])  # pools[s] was last read by the episode two back
                if on_device:
                    sample(s)
                else:
                    send[s].copy_(host[s], non_blocking=True)
                    copied[s] = torch.cuda.Event()
                    copied[s].record()
                (pools[s] if landing is None else landing).copy_(exchange_slices(s).permute(1, 0, 2))
                if landing is not None:
                    self._group_pairs(landing.view(-1), pools[s].view(-1))
                routed[s] = torch.cuda.Event()
                routed[s].record()

        def train(s):
            compute = torch.cuda.current_stream(self.device) if cuda else None
            if cuda:
                compute.wait_event(routed[s])
            for i, step in enumerate(self._schedule):
                hp, tp = int(step[r][0]), int(step[r][1])
                self._train_block(state, hp, tp, pools[s][mine[(hp, tp)]].view(-1))
                if W > 1:
                    self._exchange(state, i)
            if cuda:
                trained[s] = torch.cuda.Event()
                trained[s].record(compute)

        self._loop_timing = {"wait_upload": 0.0, "enqueue": 0.0, "wait_fill": 0.0, "fill": 0.0}
        per_episode = len(self._schedule) * self.episode_size * self.positive_reuse * W
        if on_device:
            # sample + route episode e + 1 while episode e trains: train(e) is enqueued first, so the host round trips of
            # the sampling rounds (reading the block counters) happen while the GPU is busy with the kernels of e
            route(0)
            current = 0
            while self.batch_id < self.num_batch:
                more = self.batch_id + per_episode < self.num_batch
                train(current)
                if more:
                    route(current ^ 1)
                current ^= 1
            return
        fill(0)
        current = 0
        while self.batch_id < self.num_batch:
            route(current)
            ta = time.time()
            if cuda and copied[current ^ 1] is not None:
                copied[current ^ 1].synchronize()  # the other host set may be refilled once its H2D has landed
            tb = time.time()
            self._fill_error = None
            filler = None
            if self.batch_id + per_episode < self.num_batch:  # no pools for an episode that will not run
                filler = threading.Thread(target=self._fill_guarded_call, args=(fill, current ^ 1))
                filler.start()
            try:
                train(current)
            finally:
                tc = time.time()
                if filler is not None:
                    filler.join()
            self._loop_timing["wait_upload"] += tb - ta
            self._loop_timing["enqueue"] += tc - tb
            self._loop_timing["wait_fill"] += time.time() - tc
            if self._fill_error is not None:
                raise self._fill_error
            current ^= 1

    def _fill_guarded_call(self, fn, arg):
        import time
        self._fill_error = None
        t0 = time.time()
        try:
            fn(arg)
        except BaseException as e:  # surfaced on the training thread
            self._fill_error = e
        if getattr(self, "_loop_timing", None) is not None:
            self._loop_timing["fill"] += time.time() - t0

    # ---- device-side positive sampling (edge mode) ---------------------------------------------------------
    def _upload_block_tables(self, state):
        """Per block this worker trains: the block's directed edges as {tail, head} local-id records and an alias
        table over their weights, packed 16 bytes per edge — what gvk_sample_edges draws from."""
        from .kernels import alias_build, packed_to_device
        edges, weights = self.graph.edges, self.graph.edge_weights
        hp_of, tp_of = self._part[edges[:, 0]], self._part[edges[:, 1]]
        state["block_tables"] = {}
        for hp, tp in sorted({(int(s[self.rank][0]), int(s[self.rank][1])) for s in self._schedule}):
            ids = np.nonzero((hp_of == hp) & (tp_of == tp))[0]
            if ids.size == 0:
                raise ValueError("block (%d, %d) has no edges; use fewer partitions for this graph" % (hp, tp))
            pairs = np.stack([self._local[edges[ids, 1]], self._local[edges[ids, 0]]], 1).astype(np.uint32)
            _, _, packed = alias_build(weights[ids])
            state["block_tables"][(hp, tp)] = self.kernels.pack_edge_table(
                packed_to_device(packed, self.device), self._to_device(pairs.view(np.int32).reshape(-1)))
        state["positive_index"] = getattr(self, "_positive_index", 0)

    def _upload_walk_graph(self, state):
        """CSR, per-vertex alias tables and the global edge table in HBM — what gvk_sample_walks walks on."""
        from .kernels import alias_build, packed_to_device
        g = self.graph
        edges, weights, flat = g.edges, g.edge_weights, g.flat_offsets
        D = g.num_directed_edge
        if D >= 2 ** 32:
            raise ValueError("device_sampling supports graphs with fewer than 2^32 directed edges")
        _, _, edge_packed = alias_build(weights)
        entry = np.dtype([("prob", np.float32), ("alias", np.uint32)])
        nb = np.zeros(D, entry)
        _lib.check(_lib.lib().gvs_graph_neighbor_tables(g._handle, self.num_sampler_per_worker + 1, nb.ctypes.data),
                   "gvs_graph_neighbor_tables")
        walk = {"flat_offsets": self._to_device(flat.astype(np.int64)),
                "edges_uv": self._to_device(edges.astype(np.uint32).view(np.int32).reshape(-1)),
                "edge_table": packed_to_device(edge_packed, self.device),
                "neighbor_table": packed_to_device(nb, self.device),
                "local": self._to_device(self._local.view(np.int32)), "part": self._to_device(self._part.astype(np.int32)),
                "biased": self._mode in ("biased_walk", "biased_reject"), "p": self.p, "q": self.q}
        if walk["biased"]:
            # ascending neighbour ids inside each vertex's CSR segment: one device sort of (u << 32 | v) keys
            uv = walk["edges_uv"].view(-1, 2).to(torch.int64) & 0xFFFFFFFF
            keys = (uv[:, 0] << 32) | uv[:, 1]
            walk["sorted_neighbors"] = (torch.sort(keys).values & 0xFFFFFFFF).to(torch.int32).contiguous()
        state["walk_graph"] = walk
        state["positive_index"] = getattr(self, "_positive_index", 0)

    def _train_episode_device_sampling(self, state):
        """One episode with the positive samples drawn on the device.  Same pipeline as `_train_episode`, with a
        sampling kernel where that one has an H2D copy: block g's pool is drawn (and regrouped, pair_order "grouped")
        into device buffer g & 1 on the copy stream while block g - 1 trains on the compute stream."""
        n = self.episode_size * self.batch_size
        seed = (self.seed * 0x9E3779B1 + 0x706f73 + self.rank) & (2 ** 64 - 1)
        cuda = self.device.type == "cuda"
        walks = self._mode != "edge"  # single partition: one block, walks drawn on the device
        steps = [(0, 0)] if walks else [(int(s[self.rank][0]), int(s[self.rank][1])) for s in self._schedule]
        ready = state.setdefault("uploaded", [None, None])      # per device buffer: its pool has been drawn
        released = state.setdefault("released", [None, None])   # per device buffer: its last reader has finished
        base = state.get("global_step", 0)

        def draw(g, block):
            """Pool of `block` for global step g into device buffer g & 1 (on the current stream)."""
            buf = state["pool_dev"][g & 1]
            landing = state.get("pool_stage", buf)
            if walks:
                L, aug = self.random_walk_length, self.augmentation_step
                self.kernels.sample_walks(state["walk_graph"], seed, state["positive_index"], landing, n, L, aug,
                                          self.shuffle_base)
                per_walk = aug * L - aug * (aug - 1) // 2
                state["positive_index"] += (n + per_walk - 1) // per_walk
            else:
                self.kernels.sample_edges(state["block_tables"][block], seed, state["positive_index"], landing, n)
                state["positive_index"] += n
            self._group_pairs(landing, buf)

        def produce(g, block):
            if not cuda:
                return draw(g, block)
            with torch.cuda.stream(state["copy_stream"]):
                if released[g & 1] is not None:
                    state["copy_stream"].wait_event(released[g & 1])
                draw(g, block)
                ready[g & 1] = torch.cuda.Event()
                ready[g & 1].record()

        per_episode = len(steps) * self.episode_size * self.positive_reuse * self.num_worker
        if state.get("produced_for") != base:  # the first pool of a training run; later ones are drawn one block ahead
            produce(base, steps[0])
        for i, (hp, tp) in enumerate(steps):
            g = base + i
            compute = torch.cuda.current_stream(self.device) if cuda else None
            if cuda:
                compute.wait_event(ready[g & 1])
            if i + 1 < len(steps):
                produce(g + 1, steps[i + 1])
            elif self.batch_id + per_episode < self.num_batch:  # next episode's first block, while this one's last trains
                produce(g + 1, steps[0])
                state["produced_for"] = g + 1
            self._train_block(state, hp, tp, state["pool_dev"][g & 1])
            if cuda:
                released[g & 1] = torch.cuda.Event()
                released[g & 1].record(compute)
            if self.num_worker > 1:
                self._exchange(state, i)
        state["global_step"] = base + len(steps)

    # ---- one episode ----------------------------------------------------------------------------------------
    def _train_episode(self, state, pools):
        """One episode from host pools: block g's pool is copied into device buffer g & 1 on the copy stream while
        block g - 1 trains; a buffer is overwritten only after the kernels that read it (two blocks earlier) are
        done.  The counter g runs across episodes, so the first upload of an episode overlaps the last block of
        the previous one.  Returns the upload events (the host pools may be refilled once they have fired)."""
        W, r = self.num_worker, self.rank
        cuda = self.device.type == "cuda"
        steps = [(int(s[r][0]), int(s[r][1])) for s in self._schedule]
        uploaded = state.setdefault("uploaded", [None, None])   # per device buffer: its H2D copy has landed
        released = state.setdefault("released", [None, None])   # per device buffer: its last reader has finished
        base = state.get("global_step", 0)
        issued = []

        def upload(i):
            b = (base + i) & 1
            buf = state["pool_dev"][b]
            if cuda:
                with torch.cuda.stream(state["copy_stream"]):
                    if released[b] is not None:
                        state["copy_stream"].wait_event(released[b])
                    landing = state.get("pool_stage", buf)
                    landing.copy_(pools[steps[i]], non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record()
                    self._group_pairs(landing, buf)
                    ev = torch.cuda.Event()
                    ev.record()
                uploaded[b] = ev
                issued.append(copied)
            else:
                landing = state.get("pool_stage", buf)
                landing.copy_(pools[steps[i]])
                self._group_pairs(landing, buf)

        upload(0)
        for i, (hp, tp) in enumerate(steps):
            b = (base + i) & 1
            compute = torch.cuda.current_stream(self.device) if cuda else None
            if cuda:
                compute.wait_event(uploaded[b])
            if i + 1 < len(steps):
                upload(i + 1)
            self._train_block(state, hp, tp, state["pool_dev"][b])
            if cuda:
                ev = torch.cuda.Event()
                ev.record(compute)
                released[b] = ev
            if W > 1:
                self._exchange(state, i)
        state["global_step"] = base + len(steps)
        return issued

    def _group_pairs(self, pool, out, num_batches=None):
        """pair_order="grouped": inside every batch of a device-resident pool, bring the pairs that share a head row
        next to each other (gvk_group_pairs, on the current stream; `out` is `pool` itself when nothing is to be done).
        The order of the samples inside a batch carries no meaning — they are i.i.d. draws that the kernel processes
        concurrently — but adjacent pairs run in the same workgroup at the same time, so a head row that k pairs of a
        batch share is fetched from HBM once instead of k times (30 % of the head rows of a 100k batch on a
        power-law graph are repeats)."""
        if self.pair_order != "grouped":
            return
        if num_batches is None:
            num_batches = pool.numel() // 2 // self.batch_size
        # a batch on a small partition is trained as Q launches of batch_size / Q samples (gvk_train_launches, DESIGN.md
        # §7.8): what is made of runs is what runs concurrently, i.e. a part
        launches = getattr(self.kernels, "train_launches", None)
        parts = launches(self.batch_size, self._part_size) if launches else 1
        with profiler_range("Regroup"):
            self.kernels.group_pairs(pool, out, self.batch_size // parts, num_batches * parts, self._part_size)

    def _tables(self, state, hp, tp):
        ti = self._my_tails.index(tp)
        slot = state["head"][state["slot_of"][hp]]
        moments = None
        if self.num_moment:
            moments = [None] * 4
            for j in range(self.num_moment):
                moments[2 * j] = slot[1 + j]
                moments[2 * j + 1] = state["context_m%d" % j][ti]
        return slot[0], state["context"][ti], moments

    def _claim_slot(self, state, hp):
        """Several GPUs: bring head partition hp to THIS rank's slot of its head group (slot x * W + rank) before it is
        trained, so that the group's slab is [what rank 0 trained][what rank 1 trained]... and the exchange is one
        in-place all-gather.  One device-local copy of a shard, at most; whatever partition sat in the slot is trained by
        another rank in this very step and arrives with the gather.  All ranks apply the same bookkeeping."""
        W, r = self.num_worker, self.rank
        group = hp // W
        step = state.setdefault("claimed", {}).get(group)
        if step is not None and step[r] == hp:
            return
        heads = self._heads_of_step(hp)
        target = group * W + r
        source = state["slot_of"][hp]
        if source != target:
            state["head"][target].copy_(state["head"][source])
        for q, part in enumerate(heads):  # after the gather of this step, slot x * W + q holds what rank q trained
            state["slot_of"][part] = group * W + q
            state["part_at"][group * W + q] = part
        state["claimed"][group] = heads

    def _heads_of_step(self, hp):
        """The head partitions the W ranks train (rank order) in the schedule steps in which THIS rank trains hp."""
        if getattr(self, "_step_heads", None) is None:
            self._step_heads = {int(step[self.rank][0]): [int(a[0]) for a in step] for step in self._schedule}
        return self._step_heads[hp]

    def _train_block(self, state, hp, tp, pool):
        """WorkerMixin::train (solver.h:1511-1522): positive_reuse x episode_size batches of one block."""
        with profiler_range("Train Batch"):  # solver.h:1526; one range per block: its batches are back-to-back launches
            self._train_block_batches(state, hp, tp, pool)

    def _train_block_batches(self, state, hp, tp, pool):
        if self.num_worker > 1:
            self._wait_exchange(state, hp // self.num_worker)  # this block reads head group hp // W
            self._claim_slot(state, hp)
        vertex, context, moments = self._tables(state, hp, tp)
        table = state["negative_tables"][tp]
        spec = self.optimizer.spec()
        W, r, B = self.num_worker, self.rank, self.batch_size
        native_schedule = self.optimizer.schedule.type in ("linear", "constant")
        seed = (self.seed * 0x100000001B3 + r) & (2 ** 64 - 1)
        for reuse in range(self.positive_reuse):
            done = 0
            while done < self.episode_size:
                # this worker's batches carry ids first, first + W, ... (one shared counter, solver.h:1520)
                first = self.batch_id + (reuse * self.episode_size + done) * W + r
                if first % self.log_frequency == 0:  # solver.h:1527-1549 (the loss is the previous batch's)
                    logger.info("Batch id: %d / %d", first, self.num_batch)
                    logger.info("loss = %g", float(state["loss"].mean().item()))
                # run up to, not including, this worker's next logging batch
                n = 1
                while n < self.episode_size - done and (first + n * W) % self.log_frequency:
                    n += 1
                if native_schedule:
                    self.kernels.train_episode(vertex, context, pool[done * B * 2:], state["loss"], spec,
                                               self.num_negative, self.negative_weight, table, seed, first,
                                               self.num_batch, n, B, moments=moments, batch_id_stride=W)
                else:  # custom Python schedule: lr computed on the host per batch (optimizer.h:132-134)
                    for b in range(n):
                        bid = first + b * W
                        lr = self.optimizer.init_lr * self.optimizer.schedule(bid, self.num_batch)
                        self.kernels.train(vertex, context, pool[(done + b) * B * 2:(done + b + 1) * B * 2].view(B, 2),
                                           state["loss"], spec, self.num_negative, self.negative_weight, table=table,
                                           seed=seed, batch_id=bid, moments=moments, lr=lr)
                done += n
        self.batch_id += self.episode_size * self.positive_reuse * W

    def _exchange(self, state, step_index):
        """After a schedule step every worker has trained a different head partition of one head group, each in its own
        slot of the group's slab (`_claim_slot`): ONE in-place all-gather of the slab — vertex rows and moment tables
        together — gives every GPU the whole, current group again (RCCL over xGMI; gloo in the CPU tests).  The collective
        is asynchronous; `_wait_exchange` fences the next block that reads that group."""
        import torch.distributed as dist
        W, r = self.num_worker, self.rank
        heads = [int(a[0]) for a in self._schedule[step_index]]
        group = min(heads) // W
        self._wait_exchange(state, group)
        self._claim_slot(state, heads[r])
        slab = state["head"][group * W:(group + 1) * W]
        with profiler_range("Exchange"):
            work = dist.all_gather_into_tensor(slab.view(-1), slab[r].view(-1), async_op=True)
        state.setdefault("pending_exchange", {})[group] = [work]
        state["exchanged_bytes"] = state.get("exchanged_bytes", 0) + slab[r].numel() * 4 * (W - 1)
        state["exchanges"] = state.get("exchanges", 0) + 1

    def _wait_exchange(self, state, group=None):
        pending = state.get("pending_exchange")
        if not pending:
            return
        for g in ([group] if group is not None else list(pending)):
            for work in pending.pop(g, []):
                work.wait()

    def _write_back(self, state, collective=True):
        """Device -> the stable host arrays behind the numpy views (WorkerMixin::write_back, solver.h:1498-1504).
        Context shards (and context moments) of the other workers arrive by all-gather; with collective=False (an error
        is propagating on this rank: its peers may be anywhere in their own collectives) only what this rank holds is
        written back and nothing is exchanged."""
        if state is None:
            return
        if "positive_index" in state:  # the device sampler's stream goes on where it stopped in the next train()
            self._positive_index = state["positive_index"]
        import torch.distributed as dist
        W, P, S = self.num_worker, self.num_partition, self._part_size
        if collective:
            self._wait_exchange(state)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

        def scatter(host, dev, slot_of_part, tables=1, table=0):
            """device [slots][tables][S][dim], table `table` -> host [N][dim] for the partitions that have a slot:
            un-permute on the device in chunks of at most 256 MiB, one D2H per chunk."""
            slot = np.asarray(slot_of_part, np.int64)
            owned = slot[self._part] >= 0
            rows = (slot[self._part[owned]] * tables + table) * S + self._local[owned].astype(np.int64)
            flat = dev.view(-1, self.dim)
            ids = np.flatnonzero(owned) if not owned.all() else None
            step = max(self._upload_chunk_bytes // (self.dim * 4), 1)
            for start in range(0, len(rows), step):
                values = flat[self._to_device(rows[start:start + step])].cpu().numpy()
                if ids is None:
                    host[start:start + len(values)] = values
                else:
                    host[ids[start:start + len(values)]] = values

        def context_slots(name):
            """(device tensor, slot of every partition or -1) of a context-side table, all workers' shards included."""
            mine = state[name]
            slot = np.full(P, -1, np.int64)
            if W == 1 or not collective:
                slot[self._my_tails] = np.arange(len(self._my_tails))
                return mine, slot
            gathered = torch.empty((W,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(gathered.view(-1), mine.view(-1))
            tails = []
            for rank in range(W):
                tails += sorted({int(step[rank][1]) for step in self._schedule})
            slot[tails] = np.arange(len(tails))
            return gathered, slot

        head_slot, nt = np.asarray(state["slot_of"], np.int64), 1 + self.num_moment
        scatter(self.vertex_embeddings, state["head"], head_slot, nt, 0)
        ctx, slot = context_slots("context")
        scatter(self.context_embeddings, ctx, slot)
        if self.num_moment:
            shape = (self.num_vertex, self.dim)
            self._moments_host = {"vertex": [np.zeros(shape, np.float32) for _ in range(self.num_moment)],
                                  "context": [np.zeros(shape, np.float32) for _ in range(self.num_moment)]}
            for j in range(self.num_moment):
                scatter(self._moments_host["vertex"][j], state["head"], head_slot, nt, 1 + j)
                cm, slot = context_slots("context_m%d" % j)
                scatter(self._moments_host["context"][j], cm, slot)
        self.exchange_stats = {"exchanges": state.get("exchanges", 0), "bytes_sent_per_gpu": state.get("exchanged_bytes", 0)}
        self._device_state = None

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
        if self.vertex_embeddings is None:
            raise RuntimeError("The model must be built on a graph first")
        if samples.size and (samples.min() < 0 or samples.max() >= self.num_vertex):
            raise ValueError("node index out of range")
        if self._predict_cache is None:
            self._predict_cache = (self._to_device(self.vertex_embeddings), self._to_device(self.context_embeddings))
        vertex, context = self._predict_cache
        n, B = samples.shape[0], max(self.batch_size, 1)
        # records are {tail, head}: numpy columns (v, c) are reversed on the way in (solver.h:1127-1132)
        pairs = self._to_device(np.ascontiguousarray(samples[:, ::-1].astype(np.uint32)).view(np.int32))
        logits = torch.empty(n, dtype=torch.float32, device=self.device)
        for start in range(0, n, B):
            self.kernels.predict(vertex, context, pairs[start:start + B], logits[start:start + B])
        return logits.cpu().numpy()

    def save_embeddings(self, file_name):
        """Save vertex embeddings in word2vec binary format: "N dim\n", then per node its name, a space, dim raw
        float32 values and a newline (GraphSolver::save_embeddings, graph.cuh:796-805; unbound in the reference)."""
        if self.vertex_embeddings is None:
            raise RuntimeError("The model must be built on a graph first")
        names = self.graph.id2name
        with open(file_name, "wb") as fout:
            fout.write(b"%d %d\n" % (self.num_vertex, self.dim))
            for i in range(self.num_vertex):
                fout.write(names[i].encode() + b" ")
                fout.write(self.vertex_embeddings[i].tobytes())
                fout.write(b"\n")

    def clear(self):
        """Free CPU and GPU memory, except the embeddings on CPU."""
        self._device_state = None
        self._predict_cache = None
        self._sampler = None
        self._sampler_mode = None
        self._moments_host = None
        if self.device.type == "cuda":
            torch.cuda.empty_cache()
