#!/usr/bin/env python
"""bench.py — million edge-samples/sec of the LINE dim-128 SGD hot path on N MI355X GPUs.

A "step" is one batch (100 000 edge-samples, num_negative 1) of negative-sampling SGD PER GPU on the synthetic
power-law graph BASELINE.json's metric is quoted on (configs[1]: 1M nodes / 10M edges, dim 128, fp32, LINE,
augmentation_step 1), driven through the product path: Graph -> GraphSolver.build (degree partition; one
partition at N = 1, 2N at N > 1) -> the native CPU edge sampler fills the block pools -> pools uploaded to HBM ->
per block visit, as in the episode loop: [regrouping pass gvk_group_pairs on the copy stream while the previous block
trains] -> gvk_train_episode (negatives drawn in-kernel, lr schedule per batch) for `--block-batches` batches ->
with N > 1 all GPUs all-gather the head shards they just trained (RCCL over xGMI, asynchronous).

The timed region is WHOLE block visits: it starts on a block boundary, so every block visit whose batches are timed
has its regrouping pass, its staging and its exchange inside the region too (`regroup` / `exchange` report what ran
inside it).  N = 1: `--block-batches` defaults to at most `--steps` and exactly K steps are timed.  N > 1: a block visit
is as long as the reference's episode rule makes it for this graph and partition count (num_vertex * 175 / P /
batch_size batches, solver.h:426-436; at most 250) — an exchange every 20 batches, 5 to 20 times as often as training
ever does it, would make a short run measure the fabric instead of the path — and the K requested steps are rounded
UP to whole visits, at least `--min-visits` of them: `steps` in the line is what was timed, `steps_requested` is K.
The region starts with no exchange in flight and closes with a fence, so of its n exchanges n - 1 overlap the next
visit (as in the episode loop) and the last one is fully exposed: the value is a lower bound of the steady state.

    python bench.py [--steps K] [--warmup W]                                                     (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                                                    (N > 1)

Rank 0 prints ONE JSON line.  The sample pools, alias tables and embedding tables are resident in HBM when
the timed region starts (sampling is a CPU producer that runs concurrently in real training; its rate is
reported separately as `sampler`, and `end_to_end` times GraphSolver.train() itself — the reference's figure of merit
`[time] GraphApplication.train`, python/graphvite/util.py:158-166 — with the CPU samplers and with device-side
sampling).  `roofline` is for the training kernel (HBM-bound): achieved = algorithmic bytes per launch (3088 B per
edge-sample at dim 128, k = 1) / the average launch duration measured with HIP events on the launch
stream over the timed region; `roofline.kernel` is what the library says it launched (gvk_describe_train).
`cpu_baseline` (N = 1, rank 0) times the reference's own host-compiled arithmetic (oracle/_ref, Hogwild over all host
cores) on a bounded sample of the same batches — on this workload (configs[1]) and on the BlogCatalog-sized
quick-start shape (configs[0]).
"""
import argparse
import json
import logging
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(dim, k):
    """Every touched row read once and written once, plus 16 B of indices / loss."""
    return 8 * dim * (k + 2) + 16


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=400)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--vertices", type=int, default=1000000)
    p.add_argument("--edges", type=int, default=10000000)
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--batch", type=int, default=100000)
    p.add_argument("--negatives", type=int, default=1)
    p.add_argument("--block-batches", type=int, default=0,
                   help="batches per (head, tail) block pool = batches between two exchanges; 0 = the solver's "
                        "auto episode size for this graph (solver.h:426-436), capped at 250 and at --steps (so that "
                        "the timed region is made of whole block visits)")
    p.add_argument("--min-visits", type=int, default=4,
                   help="N > 1 without --block-batches: the timed region holds at least this many whole block visits")
    p.add_argument("--no-end-to-end", action="store_true", help="skip the GraphSolver.train() runs (`end_to_end`)")
    p.add_argument("--end-to-end", action="store_true",
                   help="run `end_to_end` with several GPUs too (default: one GPU only — an error on one rank inside a "
                        "training run would leave the others waiting in a collective, and the headline line with them)")
    p.add_argument("--end-to-end-batches", type=int, default=36000,
                   help="batches per GPU of each end-to-end run: about twenty episodes (the auto episode size is 1750 batches "
                        "here), so that the first pool fill — the one nothing can overlap — is a twentieth of the run, not a "
                        "seventh; real trainings run hundreds of episodes")
    p.add_argument("--lanes", type=int, default=0, help="A/B knob: lanes per pair (0 = per-dim default)")
    p.add_argument("--variant", type=int, default=0, help="A/B knob: kernel build variant (gvk.h GVK_TUNE_VARIANT)")
    p.add_argument("--run-cap", type=int, default=0, help="A/B knob: longest same-head run per lane group (gvk.h GVK_TUNE_RUN_CAP)")
    p.add_argument("--segment-steps", type=int, default=0, help="A/B knob: pairs per lane group and wavefront (GVK_TUNE_SEGMENT_STEPS)")
    p.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                   help="A/B knob: any GVK_TUNE_* key of include/gvk.h by number, e.g. --tune 6=1")
    p.add_argument("--xcd-bucket", choices=["head", "tail"], default=None,
                   help="experiment: reorder every batch so that block b (16 pairs, XCD b % 8) holds pairs whose "
                        "head / tail row id is congruent to b mod 8")
    p.add_argument("--partitions", type=int, default=0,
                   help="experiment: vertex partitions (default: 1 on one GPU, 2 x #GPU otherwise); on one GPU this shows "
                        "the kernel at the shard size of a multi-GPU run")
    p.add_argument("--host-order", choices=["head", "head+tail", "tail"], default=None,
                   help="experiment: batches pre-ordered on the host (no device pass): by head row; by head row with the "
                        "pairs whose head is unique in the batch ordered by tail row instead; by tail row")
    p.add_argument("--xcd-sorted", action="store_true",
                   help="experiment: with --xcd-bucket, also sort each bucket by row (same-row pairs adjacent in time)")
    p.add_argument("--pair-order", choices=["auto", "sampled", "grouped"], default="auto",
                   help="GraphSolver(pair_order=...): auto (the product default) regroups the pairs of a batch by head "
                        "row on the device when a partition's table reaches 16 MiB")
    p.add_argument("--optimizer", choices=["SGD", "Momentum", "AdaGrad", "RMSprop", "Adam"], default="SGD",
                   help="experiment: moment optimizers move (1 + m) x the row bytes (m = 1, Adam 2)")
    p.add_argument("--graph", choices=["power-law", "community"], default="power-law",
                   help="experiment: 'community' swaps in a hub-free planted-partition graph of the same size")
    p.add_argument("--sampler-threads", type=int, default=0, help="0 = host cores / GPUs")
    p.add_argument("--negative-table", choices=["auto", "rows", "classes"], default="auto",
                   help="the negative sampler's table: one alias slot per row (the reference's), or an alias table over the "
                        "classes of equal-degree rows (same distribution, cache-resident); auto = classes when 8x fewer")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-access-pattern", action="store_true", help="skip the row-traffic probe (`roofline.access_pattern`)")
    p.add_argument("--cpu-seconds", type=float, default=3.0, help="wall seconds given to the CPU baseline")
    p.add_argument("--seed", type=int, default=1024)
    return p.parse_args(argv)


def time_reference(ref, cores, v, pool, prob, alias, B, k, seconds, seed):
    """Hogwild over `cores` threads with the reference's own arithmetic on batches of `pool` ({tail, head} records,
    partition-local ids), negatives from the given alias table.  Returns (edge-samples/s, batches, seconds)."""
    rng = np.random.default_rng(seed)
    n = len(v)
    c = np.zeros_like(v)
    nb = pool.shape[0] // B
    negs = []
    for _ in range(4):  # negatives from the same alias table (numpy, vectorised)
        idx = rng.integers(0, n, (B, k))
        u = rng.random((B, k)).astype(np.float32)
        negs.append(np.where(u < prob[idx], idx, alias[idx]).astype(np.uint32))
    done, t0 = 0, time.perf_counter()
    while True:
        b = done % nb
        ref.train_mt(v, c, pool[b * B:(b + 1) * B], negs[done % len(negs)], 0.025, 0.005, 5.0, cores)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds and done >= 2:
            return done * B / el, done, el


def cpu_baseline(args, solver, pool, table_packed):
    """Reference arithmetic (oracle/_ref/libgvref_fast.so), Hogwild over all host cores, on the first batches of
    rank 0's first block pool, starting from the same initial tables; then the same on the quick-start shape
    (configs[0]: a BlogCatalog-sized graph, whose 5 MB tables live in the CPU caches).  Bench infrastructure only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Reference
    try:
        ref = Reference(fast=True)
    except (FileNotFoundError, OSError):
        return None
    from graphvite_amd.base import cpu_budget
    cores = cpu_budget()  # CPUs the container may really use (cgroup quota), not the 256 hardware threads it sees
    B, k = args.batch, args.negatives
    prob, alias = np.ascontiguousarray(table_packed["prob"]), np.ascontiguousarray(table_packed["alias"])
    v = solver.vertex_embeddings[solver._part_ids[0]].copy()   # partition-local tables, like the GPU's
    rate, done, el = time_reference(ref, cores, v, pool, prob, alias, B, k, args.cpu_seconds, args.seed + 7)
    how = ("%.1f s wall, %d threads = the container's CPU quota on a %d-thread host, Hogwild; -Ofast x86-64-v3 host "
           "build of the reference's own LINE::forward/backward + sgd_update" % (el, cores, os.cpu_count() or 1))
    out = {"value": rate / 1e6, "unit": "million edge-samples/sec", "cores": cores, "kind": "reference",
           "sample": "%d batches of %d edge-samples of rank 0's first block pool (%s)" % (done, B, how)}
    # configs[0]: the reference's CPU-runnable case — quick-start hyper-parameters on a BlogCatalog-sized graph
    import graphvite_amd as gv
    from graphvite_amd import hostlib, synthetic
    from graphvite_amd.kernels import alias_build
    graph = gv.graph.Graph()
    graph.load(synthetic.hub_community_edges(10312, 333983, gamma=2.8, num_community=39, seed=args.seed))
    part, local, sizes = hostlib.partition(graph.vertex_weights, 1)
    sampler = hostlib.Sampler(graph, part, local, 1, args.seed)
    sampler.prepare("walk", 1.0, 1.0, cores)
    pool1 = np.zeros((20 * B, 2), np.uint32)
    import torch
    sampler.fill({(0, 0): torch.from_numpy(pool1.view(np.int32).reshape(-1))}, 20 * B, "walk", 4 * cores,
                 sample_batch_size=4000, walk_length=40, walk_batch=100, augmentation_step=2, shuffle_base=2,
                 tail_partition=-1, os_threads=cores)
    order = np.argsort(local)
    w = hostlib.negative_weights(graph.vertex_weights, order, 0.75)
    prob1, alias1, _ = alias_build(w)
    rng = np.random.default_rng(args.seed)
    v1 = ((rng.random((graph.num_vertex, args.dim), dtype=np.float32) - np.float32(0.5)) / np.float32(args.dim))
    rate1, done1, el1 = time_reference(ref, cores, v1, pool1, prob1, alias1, B, k, args.cpu_seconds, args.seed + 8)
    out["c1"] = {"value": rate1 / 1e6, "unit": "million edge-samples/sec", "cores": cores, "kind": "reference",
                 "sample": "configs[0] shape: %d batches of %d edge-samples (LINE, augmentation_step 2 walk sampler) on a "
                           "synthetic BlogCatalog-sized graph (10 312 nodes / 333 983 edges), %.1f s wall, same build and "
                           "threads" % (done1, B, el1)}
    return out


def access_pattern(solver, session, landed, blocks, B, dim, launches=200):
    """The ceiling of the memory system for what a training batch touches: gvk_probe_row_traffic reads and writes back
    the head row, the tail row and one negative row per edge-sample of the SAME pools on the SAME tables (fresh negatives
    per launch, drawn as the training kernel draws them), no arithmetic, no dependent draw.  HIP events around
    back-to-back launches on the launch stream."""
    import torch
    hp, tp = blocks[0]
    vertex, context, _ = solver._tables(session.state, hp, tp)
    pool = landed[blocks[0]]
    batches = min(pool.numel() // 2 // B, launches)
    table = session.negative_table(tp)
    negatives = torch.empty((batches, B), dtype=torch.int32, device=vertex.device)
    for b in range(batches):
        solver.kernels.negative_draw(table, 0x51ed, b, negatives[b], B, 1)

    def sweep():
        for i in range(launches):
            b = i % batches
            solver.kernels.probe_row_traffic(vertex, context, pool[b * B * 2:(b + 1) * B * 2], negatives[b])
    sweep()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sweep()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    achieved = algorithmic_bytes(dim, 1) * B / (ms * 1e-3)
    return {"kernel": "probe_rows_kernel: the rows of a batch read and written back, nothing else", "kernel_ms": ms,
            "achieved": achieved / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "launches": launches}


def end_to_end(args, gv, graph, world, threads, partitions):
    """GraphSolver.train() as a user calls it, timed by this process: the reference's own figure of merit is the wall
    time of GraphApplication.train (python/graphvite/util.py:158-166).  Two runs: CPU sampler threads feeding the GPU
    (the north_star pipeline) and positives drawn on the device.  `value` counts the episode loop (sampling, uploads,
    regrouping, kernels, exchanges — everything between the first and the last batch); `train_seconds` is the whole
    call including embedding init, table upload and write-back."""
    import torch
    out = {}
    B = args.batch
    epochs = max(args.end_to_end_batches * world * B // graph.num_edge, 1)
    for name, device_sampling in (("cpu_samplers", False), ("device_sampling", True)):
        solver = gv.solver.GraphSolver(args.dim, num_sampler_per_worker=threads, seed=args.seed,
                                       device_sampling=device_sampling,
                                       pair_order=gv.auto if args.pair_order == "auto" else args.pair_order)
        solver.build(graph, optimizer=gv.optimizer.SGD(0.025, 0.005, "linear"), num_partition=partitions,
                     num_negative=args.negatives, batch_size=B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.train(model="LINE", num_epoch=epochs, augmentation_step=1, log_frequency=1 << 30)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        timing = solver.timing
        out[name] = {"value": timing["batches"] * B / timing["episodes"] / 1e6, "unit": "million edge-samples/sec",
                     "batches": timing["batches"], "episode_seconds": timing["episodes"], "train_seconds": wall,
                     "sampler_threads_per_gpu": 0 if device_sampling else threads, "pair_order": solver.pair_order,
                     "episode_size": solver.episode_size,
                     "exchange_bytes_sent_per_gpu_per_step": (solver.exchange_stats["bytes_sent_per_gpu"] //
                                                              max(solver.exchange_stats["exchanges"], 1))}
        solver.clear()
        del solver
        torch.cuda.empty_cache()
    return out


def main(argv=None, stand_in_kernels=None):
    """stand_in_kernels is the test seam of tests/bench_dry_run.py: with a kernel stand-in injected, the same loop runs on
    the CPU over gloo (a logic test of the multi-rank walk; the JSON line says so).  bench.py itself never sets it."""
    args = parse(argv)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with `python -m torch.distributed.run "
                         "--nproc-per-node %d ... bench.py --gpus %d`" % (args.gpus, world, args.gpus, args.gpus))
    cuda = stand_in_kernels is None
    if not cuda:
        args.no_cpu_baseline = True
    if cuda:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    if world > 1:
        if cuda:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import graphvite_amd as gv
    from graphvite_amd import synthetic
    gv.init_logging(logging.ERROR)

    N, E, B, k, dim = args.vertices, args.edges, args.batch, args.negatives, args.dim
    # two head groups per GPU (P = 2 * #GPU) let the all-gather of one group overlap the training on the other
    partitions = args.partitions or (world if world == 1 else 2 * world)
    steps_requested = args.steps
    if not args.block_batches:
        auto = max(int(float(N) * 175 / partitions / B), 1)  # the reference's episode size, solver.h:426-436
        if world == 1:
            auto = max(auto, int(2e7) // B)
            args.block_batches = max(min(auto, 250, args.steps), 1)
        else:  # whole visits of the real length, see the docstring
            args.block_batches = max(min(auto, 250), 1)
            args.steps = max(-(-args.steps // args.block_batches), args.min_visits) * args.block_batches
    from graphvite_amd.base import cpu_budget
    threads = args.sampler_threads or max(cpu_budget() // world, 1)

    # ---- product path up to the resident state ----
    graph = gv.graph.Graph()
    if args.graph == "community":
        graph.load(synthetic.community_edges(N, E, num_community=max(N // 1000, 1), seed=args.seed))
    else:
        graph.load(synthetic.power_law_edges(N, E, seed=args.seed))
    if args.xcd_bucket or args.xcd_sorted or args.host_order:
        args.pair_order = "sampled"  # the placement experiments lay the batches out themselves
    solver = gv.solver.GraphSolver(dim, num_sampler_per_worker=threads, seed=args.seed, kernels=stand_in_kernels, pair_order=gv.auto if args.pair_order == "auto" else args.pair_order)
    solver.negative_table = args.negative_table
    if args.lanes:
        solver.kernels.set_lanes_per_pair(args.lanes)
    if args.variant:
        solver.kernels.set_variant(args.variant)
    if args.run_cap:
        solver.kernels.set_run_cap(args.run_cap)
    if args.segment_steps:
        solver.kernels.set_segment_steps(args.segment_steps)
    for item in args.tune:
        key, value = item.split("=")
        solver.kernels.set_tuning(int(key), int(value))
    optimizer = gv.optimizer.SGD(0.025, 0.005, "linear") if args.optimizer == "SGD" else \
        gv.optimizer.Optimizer(args.optimizer, 1e-3, 0.005)
    solver.build(graph, optimizer=optimizer, num_partition=partitions, num_negative=k,
                 batch_size=B, episode_size=args.block_batches)
    residency = 2 * (partitions * partitions // world)  # two batches of every block before the warm-up steps
    total_batches = (residency + args.warmup + args.steps) * world
    epochs = total_batches * B // graph.num_edge + 1
    session = solver.session(model="LINE", num_epoch=epochs, augmentation_step=1, log_frequency=1 << 30)
    solver.num_batch = total_batches  # the lr schedule spans exactly the batches this run trains
    pools = session.new_host_pools()
    t0 = time.perf_counter()
    session.fill(pools)
    fill_s = time.perf_counter() - t0
    blocks = session.blocks
    if args.host_order:
        for pool in pools.values():
            rec = pool.numpy().view(np.uint32).reshape(-1, B, 2)
            for i in range(rec.shape[0]):
                r = rec[i]
                if args.host_order == "tail":
                    rec[i] = r[np.argsort(r[:, 0], kind="stable")]
                    continue
                r = r[np.argsort(r[:, 1], kind="stable")]
                if args.host_order == "head+tail":
                    h = r[:, 1]
                    single = np.ones(B, bool)
                    single[1:] &= h[1:] != h[:-1]
                    single[:-1] &= h[:-1] != h[1:]
                    lone = r[single]
                    r = np.concatenate([r[~single], lone[np.argsort(lone[:, 0], kind="stable")]])
                rec[i] = r
    if args.xcd_sorted and not args.xcd_bucket:  # experiment: whole batch sorted by head row, no XCD placement
        for pool in pools.values():
            rec = pool.numpy().view(np.uint32).reshape(-1, B, 2)
            for i in range(rec.shape[0]):
                rec[i] = rec[i][np.argsort(rec[i, :, 1], kind="stable")]
    if args.xcd_bucket:
        column = 1 if args.xcd_bucket == "head" else 0
        for pool in pools.values():
            rec = pool.numpy().view(np.uint32).reshape(-1, B, 2)
            for i in range(rec.shape[0]):
                cls = rec[i, :, column] % 8
                order = np.lexsort((rec[i, :, column], cls)) if args.xcd_sorted else np.argsort(cls, kind="stable")
                counts = np.bincount(cls, minlength=8)
                nmin = int(counts.min()) // 16
                starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
                within = np.arange(B) - starts[cls[order]]          # rank of the pair inside its class
                placed = within < nmin * 16
                dest = np.empty(B, np.int64)
                dest[placed] = (8 * (within[placed] // 16) + cls[order][placed]) * 16 + within[placed] % 16
                dest[~placed] = nmin * 16 * 8 + np.arange(int((~placed).sum()))
                out = np.empty_like(rec[i])
                out[dest] = rec[i][order]
                rec[i] = out
    landed = session.upload(pools, group=False)  # every block pool of this GPU's column, resident in HBM
    sampled = len(blocks) * args.block_batches * B
    # With pair_order "grouped" the episode loop regroups a pool on the copy stream after its H2D copy, while the
    # previous block trains (GraphSolver._train_episode).  The timed loop below does the same for every block visit —
    # everything except the PCIe copy — so the cost of the regrouping pass is inside the measurement.
    grouped = solver.pair_order == "grouped"
    work = [torch.empty_like(next(iter(landed.values()))) for _ in range(2)] if grouped else [None, None]
    copy_stream = torch.cuda.Stream(dev) if cuda else None
    ready, released = [None, None], [None, None]
    regroup_events = []  # (start, end) of every regrouping pass staged while `timing["on"]`
    timing = {"on": False}

    def stage(step):
        """Stage the pool of block visit `step`: the regrouping pass, on the copy stream."""
        pool = landed[blocks[step % len(blocks)]]
        if not grouped:
            return pool
        b = step & 1
        if not cuda:
            return session.stage(pool, work[b])
        with torch.cuda.stream(copy_stream):
            if released[b] is not None:
                copy_stream.wait_event(released[b])
            if timing["on"]:
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
            session.stage(pool, work[b])
            if timing["on"]:
                g1.record()
                regroup_events.append((g0, g1))
            ready[b] = torch.cuda.Event()
            ready[b].record()
        return work[b]

    first_table = session.negative_table(blocks[0][1])
    negative_table = ("%d weight classes (alias table over the classes of equal-degree rows, then a uniform row of the class)"
                      % first_table.shape[0]) if first_table.dim() == 2 else "%d row slots" % first_table.numel()
    kernel_events = []
    # The walk over the schedule continues across the residency pass, the warm-up and the timed steps, block by block
    # as the episode loop walks it: block_batches batches of a block, the exchange, the next block — whose pool was
    # staged while this one trained.  A run of K steps simply trains the next K batches of that walk.
    walk = {"step": 0, "offset": 0, "pool": None, "next": None}

    def run(num_batches, timed, leave_block=False):
        compute = torch.cuda.current_stream(dev) if cuda else None
        done = 0
        while done < num_batches:
            step = walk["step"]
            hp, tp = blocks[step % len(blocks)]
            if walk["offset"] == 0:  # entering a block
                if walk["pool"] is None:
                    walk["pool"] = stage(step)
                if grouped and cuda:
                    compute.wait_event(ready[step & 1])
                walk["next"] = stage(step + 1)
                session.wait_exchange(hp)  # fence here, so that the events below bracket kernels only
            n = min(args.block_batches - walk["offset"], num_batches - done)
            if timed and cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            session.train_block(hp, tp, walk["pool"][walk["offset"] * B * 2:], n)
            if timed and cuda:
                e1.record()
                kernel_events.append((e0, e1, n))
            done += n
            walk["offset"] += n
            if walk["offset"] == args.block_batches or (leave_block and done == num_batches):
                if grouped and cuda:
                    released[step & 1] = torch.cuda.Event()
                    released[step & 1].record(compute)
                session.exchange(step)
                walk.update(step=step + 1, offset=0, pool=walk["next"], next=None)

    def fence():
        if cuda:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if cuda:
            torch.cuda.synchronize()

    # Residency pass before the W warm-up steps: two batches of every block (code-object load, first touch of every
    # table / pool, runtime pools growing) and the first collective (RCCL communicator + buffers).  One-time costs of
    # tens of milliseconds otherwise land inside a timed region that is itself only tens of milliseconds long.
    for _ in blocks:
        run(min(2, args.block_batches), False, leave_block=True)
    session.wait_exchange()
    # The ceiling of the access pattern is measured here, ahead of the warm-up: 400 launches that read and write back the
    # rows of the pools' batches, tables unchanged (A/B'd against running it after the timed region: no difference).
    probe = None
    if cuda and k == 1 and optimizer.num_moment == 0 and not args.no_access_pattern:
        probe = access_pattern(solver, session, landed, blocks, B, dim)
    # the warm-up ends on a block boundary: the timed region then consists of whole block visits; as in the steady
    # state of the episode loop every visit stages (regroups) the pool of the NEXT visit while it trains, so the
    # region holds exactly one staging pass and one exchange per visit
    run(args.warmup, False, leave_block=True)
    fence()
    timing["on"] = True
    t0 = time.perf_counter()
    run(args.steps, True)
    fence()
    wall = time.perf_counter() - t0
    timing["on"] = False
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if cuda:
        kernel_ms = sum(a.elapsed_time(b) for a, b, _ in kernel_events) / sum(n for _, _, n in kernel_events)
    else:
        kernel_ms = wall / args.steps * 1e3  # dry run: nothing to measure
    final_loss = float(session.loss.mean().item())
    regroup_ms = sum(a.elapsed_time(b) for a, b in regroup_events) if cuda else 0.0
    visits = -(-args.steps // args.block_batches)
    exchange = {"exchanges": session.state.get("exchanges", 0), "bytes_sent_per_gpu": session.state.get("exchanged_bytes", 0)}

    moments = optimizer.num_moment
    bytes_per_launch = (8 * dim * (k + 2) * (1 + moments) + 16) * B  # moment tables are rows read + written too
    achieved = bytes_per_launch / (kernel_ms * 1e-3)
    # HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM); bench.py cannot read PMCs itself
    traffic, pmc_path = None, None
    if world == 1 and dim == 128 and k == 1 and B == 100000 and N == 1000000 and moments == 0:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary_bench_n1.json")), reverse=True):
            traffic, pmc_path = json.load(open(path)).get("traffic_bytes_per_launch"), os.path.relpath(path, ROOT)
            break
    kernel_name = solver.kernels.describe_train(dim, args.optimizer, k, False, B, solver._part_size) if cuda else "stand-in"
    result = {
        "metric": "million edge-samples/sec at dim=%d" % dim,
        "value": world * args.steps * B / wall / 1e6,
        "unit": "million edge-samples/sec",
        "n_gpus": world, "steps": args.steps, "steps_requested": steps_requested, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if cuda else "DRY RUN on the CPU with the test stand-in: logic test, not a result",
        "config": {"workload": "LINE (augmentation_step 1) on synthetic power-law %d nodes / %d edges, dim %d, "
                               "batch %d edge-samples per GPU per step, num_negative %d, SGD lr 0.025 wd 0.005 linear, "
                               "negatives drawn in-kernel, block pools resident in HBM" % (N, E, dim, B, k),
                   "parallelism": "%d GPU(s), %d vertex partition(s), context shards pinned per GPU, asynchronous "
                                  "all-gather of head shards every %d batches" % (world, partitions, args.block_batches),
                   "block_batches": args.block_batches, "block_visits_timed": visits,
                   "negative_table": negative_table,
                   "pair_order": solver.pair_order + (" (gvk_group_pairs once per block visit on the copy stream)"
                                                      if grouped else "")},
        "regroup": {"passes_in_timed_region": len(regroup_events), "ms_per_pass": regroup_ms / max(len(regroup_events), 1),
                    "regroup_ms_per_step": regroup_ms / args.steps,
                    "note": "runs on the copy stream concurrently with the previous block's kernels; ms_per_step "
                            "(wall) already contains whatever of it was not hidden"} if grouped else None,
        "exchange": {"collectives_total": exchange["exchanges"],
                     "bytes_sent_per_gpu_per_collective": exchange["bytes_sent_per_gpu"] // max(exchange["exchanges"], 1),
                     "note": "one in-place all_gather_into_tensor of a head group's slab per schedule step"}
        if world > 1 else None,
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": pmc_path,
                     "kernel": kernel_name, "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": bytes_per_launch},
        "sampler": {"value": sampled / fill_s / 1e6, "unit": "million edge-samples/sec per GPU", "threads": threads,
                    "note": "CPU edge sampler filling this GPU's block pools before the timed region"},
        "final_batch_mean_loss": final_loss,
    }
    if probe is not None:  # how close the training kernel is to what the memory system sustains for ITS access pattern
        result["roofline"]["access_pattern"] = probe
        probe["train_kernel_vs_probe"] = probe["kernel_ms"] / kernel_ms
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        tp = blocks[0][1]
        # the reference's own sampler: one alias slot per row of the tail partition (solver.h:1264-1278)
        from graphvite_amd import hostlib
        from graphvite_amd.kernels import alias_build
        _, _, packed = alias_build(hostlib.negative_weights(graph.vertex_weights, solver._part_ids[tp], 0.75))
        pool0 = pools[blocks[0]].numpy().view(np.uint32).reshape(-1, 2)
        result["cpu_baseline"] = cpu_baseline(args, solver, pool0, packed)
    if cuda and not args.no_end_to_end and args.optimizer == "SGD" and (world == 1 or args.end_to_end):
        session.finish()
        del landed, work, pools, session
        solver.clear()
        torch.cuda.empty_cache()
        try:
            result["end_to_end"] = end_to_end(args, gv, graph, world, threads, partitions)
        except Exception as error:  # the headline measurement above stands on its own; say what happened to this one
            result["end_to_end"] = {"error": "%s: %s" % (type(error).__name__, error)}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
