#!/usr/bin/env python
"""bench.py — million edge-samples/sec of the LINE dim-128 SGD hot path on N MI355X GPUs.

A "step" is one batch (100 000 edge-samples, num_negative 1) of negative-sampling SGD PER GPU on the synthetic
power-law graph BASELINE.json's metric is quoted on (configs[1]: 1M nodes / 10M edges, dim 128, fp32, LINE,
augmentation_step 1), driven through the product path — the native solver engine (include/gvx.h, one process per GPU,
RCCL over xGMI) stepped through its session API: Graph -> GraphSolver.build (degree partition; P = N partitions) -> the
native CPU edge sampler fills the block pools -> pools uploaded to HBM -> per block visit, as in the episode loop:
[regrouping pass gvk_group_pairs on the copy stream while the previous block trains] -> the slot claim -> the block's
`--block-batches` batches (negatives drawn in-kernel, lr schedule per batch): gvk_train_episode_hot wherever a table has hub
rows — the default executor of the headline shape: hub rows by chains, a batch as 8 launches of train_hot_kernel, within
0.002 AUC of the reference's sequential loop —, gvk_train_episode (one launch per batch) otherwise or with
`--fidelity throughput` -> with N > 1 ONE in-place ncclAllGather of the head group's slab (asynchronous).

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
`[time] GraphApplication.train`, python/graphvite/util.py:158-166 — with the CPU samplers, with device-side
sampling, and (`module`) through the pybind11 module `libgraphvite`, the boundary a user of the reference's Python
package loads).  `auc` is the link-prediction AUC of a training of THIS configuration (same graph shape, same worker and
partition count) next to the reference's own loop on the same shape (tests/golden/reference_c2.npz).  `roofline` is for
the training kernel (HBM-bound): achieved = algorithmic bytes per launch (3088 B per edge-sample at dim 128, k = 1;
SURVEY.md §8d; a launch of train_hot_kernel trains batch / `launches_per_step` samples) / the average launch duration
measured with HIP events on the launch stream over the timed region; `roofline.kernel` is what ran; `roofline.traffic` is null
unless a PMC summary of the same round's command sits under profiles/ (`traffic_source`).  N > 1 adds
`single_gpu_same_shards`: the one-GPU rate at the same shard size (`--gpus 1 --partitions N`, run by rank 0 after the
measurement), so that scaling and the cache effect of smaller shards can be told apart.  `cpu_baseline` (N = 1, rank 0) times the
reference's own host-compiled arithmetic (oracle/_ref, Hogwild over all host cores) on a bounded sample of the same
batches — on this workload (configs[1]) and on the BlogCatalog-sized quick-start shape (configs[0]).
"""
import argparse
import json
import logging
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
L2_BYTES, INFINITY_CACHE_BYTES = 8 * (4 << 20), 256 << 20


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
    p.add_argument("--no-end-to-end", action="store_true", help="skip the GraphSolver.train() runs (`end_to_end`, `auc`)")
    p.add_argument("--end-to-end-batches", type=int, default=36000,
                   help="batches per GPU of each end-to-end run: about twenty episodes (the auto episode size is 1750 batches "
                        "here), so that the first pool fill — the one nothing can overlap — is a twentieth of the run, not a "
                        "seventh; real trainings run hundreds of episodes")
    p.add_argument("--auc-epochs", type=int, default=50, help="epochs of the training whose link-prediction AUC is reported")
    p.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                   help="A/B knob: any GVK_TUNE_* key of include/gvk.h by number, e.g. --tune 2=4")
    p.add_argument("--partitions", type=int, default=0,
                   help="vertex partitions (default: the reference's minimum, one per GPU); on one GPU, P > 1 shows the "
                        "kernel at the shard size of a multi-GPU run")
    p.add_argument("--pair-order", choices=["auto", "sampled", "grouped"], default="auto",
                   help="GraphSolver(pair_order=...): auto (the product default) regroups the pairs of a batch by head "
                        "row on the device by table size (DESIGN.md §3.1.1)")
    p.add_argument("--hub-rows", default="default",
                   help="GraphSolver(hub_rows=...): default = the product's rule (off for LINE), 0 = every row trained pair by pair, "
                        "`auto` / N = the hub rows of each partition (auto: expected hits per batch >= 2) trained by chains "
                        "(gvk_train_episode_hot, DESIGN.md §3.1.2)")
    p.add_argument("--fidelity", choices=["auto", "reference", "throughput"], default="auto",
                   help="GraphSolver(fidelity=...): auto (the product's default: hub rows by chains where chains exist), "
                        "throughput = every row pair by pair")
    p.add_argument("--no-fidelity-leg", action="store_true", help="skip auc.fidelity_throughput (the same training with "
                        "GraphSolver(fidelity='throughput'))")
    p.add_argument("--hub-parts", type=int, default=0, help="GraphSolver.hub_parts: with hub rows by chains, a batch as this many parts")
    p.add_argument("--hub-lerp", type=int, default=-1, help="GraphSolver.hub_lerp: -1 the rule, 0 / 1 (the pairs read hub rows along their chains' way)")
    p.add_argument("--hub-cap", type=int, default=0, help="GraphSolver.hub_chain_cap: entries one chain task trains in sequence (0 = default)")
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
    p.add_argument("--no-module", action="store_true", help="skip the run through the pybind11 module (`end_to_end.module`)")
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


def cpu_baseline(args, graph):
    """Reference arithmetic (oracle/_ref/libgvref_fast.so), Hogwild over all host cores, on batches the product's own
    CPU edge sampler drew for this graph (one partition, partition-local ids), negatives from the reference's one-slot-per-
    row alias table, starting from the reference's initial tables; then the same on the quick-start shape (configs[0]: a
    BlogCatalog-sized graph, whose 5 MB tables live in the CPU caches).  Bench infrastructure only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Reference
    try:
        ref = Reference(fast=True)
    except (FileNotFoundError, OSError):
        return None
    import torch
    import graphvite_amd as gv
    from graphvite_amd import hostlib, synthetic
    from graphvite_amd.base import cpu_budget
    from graphvite_amd.kernels import alias_build
    cores = cpu_budget()  # CPUs the container may really use (cgroup quota), not the 256 hardware threads it sees
    B, k = args.batch, args.negatives

    def sample(g, mode, batches, **walk):
        part, local, sizes = hostlib.partition(g.vertex_weights, 1)
        sampler = hostlib.Sampler(g, part, local, 1, args.seed)
        sampler.prepare(mode, 1.0, 1.0, cores)
        pool = np.zeros((batches * B, 2), np.uint32)
        sampler.fill({(0, 0): torch.from_numpy(pool.view(np.int32).reshape(-1))}, batches * B, mode, 4 * cores,
                     sample_batch_size=4000, walk_length=40, walk_batch=100, tail_partition=-1, os_threads=cores, **walk)
        order = np.argsort(local)
        prob, alias, _ = alias_build(hostlib.negative_weights(g.vertex_weights, order, 0.75))
        rng = np.random.default_rng(args.seed)
        v = (rng.random((g.num_vertex, args.dim), dtype=np.float32) - np.float32(0.5)) / np.float32(args.dim)
        return v, pool, prob, alias

    v, pool, prob, alias = sample(graph, "edge", 20, augmentation_step=1, shuffle_base=1)
    rate, done, el = time_reference(ref, cores, v, pool, prob, alias, B, k, args.cpu_seconds, args.seed + 7)
    how = ("%.1f s wall, %d threads = the container's CPU quota on a %d-thread host, Hogwild; -Ofast -march=x86-64-v3 (not -march=native: built "
           "in another container than the one that runs it, oracle/Makefile) host build of the reference's own LINE::forward/backward + sgd_update" % (el, cores, os.cpu_count() or 1))
    out = {"value": rate / 1e6, "unit": "million edge-samples/sec", "cores": cores, "kind": "reference",
           "sample": "%d batches of %d edge-samples drawn by the CPU edge sampler for this graph (%s)" % (done, B, how)}
    # configs[0]: the reference's CPU-runnable case — quick-start hyper-parameters on a BlogCatalog-sized graph
    small = gv.graph.Graph()
    small.load(synthetic.hub_community_edges(10312, 333983, gamma=2.8, num_community=39, seed=args.seed))
    v1, pool1, prob1, alias1 = sample(small, "walk", 20, augmentation_step=2, shuffle_base=2)
    rate1, done1, el1 = time_reference(ref, cores, v1, pool1, prob1, alias1, B, k, args.cpu_seconds, args.seed + 8)
    out["c1"] = {"value": rate1 / 1e6, "unit": "million edge-samples/sec", "cores": cores, "kind": "reference",
                 "sample": "configs[0] shape: %d batches of %d edge-samples (LINE, augmentation_step 2 walk sampler) on a "
                           "synthetic BlogCatalog-sized graph (10 312 nodes / 333 983 edges), %.1f s wall, same build and "
                           "threads" % (done1, B, el1)}
    return out


def auc_episode_size(partitions):
    """Episode size of the trainings whose AUC is compared with the reference's loop at P > 1 partitions: about 512 batches per
    episode (all P x P blocks), like the goldens c2_line_p<P>_e<E> (tests/golden/make_c2_golden.py).  The automatic size makes
    a 50-epoch training shorter than one episode from P = 4 on: every block would be visited once, the last ones under a
    learning rate that has all but decayed — a protocol artefact, not a property of the path."""
    return max(512 // (partitions * partitions), 1)


def train_timed(args, gv, graph, threads, partitions, device_sampling, epochs, fidelity=None, episode_size=None):
    """One GraphSolver.train() as a user calls it, timed by this process."""
    import torch
    solver = gv.solver.GraphSolver(args.dim, num_sampler_per_worker=threads, seed=args.seed, device_sampling=device_sampling,
                                   pair_order=gv.auto if args.pair_order == "auto" else args.pair_order, fidelity=fidelity or args.fidelity,
                                   hub_rows=None if args.hub_rows == "default" else (args.hub_rows if args.hub_rows == "auto" else int(args.hub_rows)))
    solver.hub_parts = args.hub_parts
    solver.hub_lerp = None if args.hub_lerp < 0 else bool(args.hub_lerp)
    solver.hub_chain_cap = args.hub_cap
    solver.negative_table = args.negative_table
    solver.build(graph, optimizer=gv.optimizer.SGD(0.025, 0.005, "linear"), num_partition=partitions,
                 num_negative=args.negatives, batch_size=args.batch, episode_size=episode_size or gv.auto)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.train(model="LINE", num_epoch=epochs, augmentation_step=1, log_frequency=1 << 30)
    wall = time.perf_counter() - t0
    return solver, wall


def end_to_end(args, gv, graph, world, threads, partitions):
    """GraphSolver.train() as a user calls it: the reference's own figure of merit is the wall time of
    GraphApplication.train (python/graphvite/util.py:158-166).  CPU sampler threads feeding the GPU (the north_star
    pipeline; one GPU only: with several, the node's CPUs are the bottleneck — DESIGN.md §4.1) and positives drawn on the
    device.  `value` counts the episode loop (sampling, uploads, regrouping, kernels, exchanges — everything between the
    first and the last batch); `train_seconds` is the whole call including embedding init, table upload and write-back."""
    out = {}
    B = args.batch
    epochs = max(args.end_to_end_batches * world * B // graph.num_edge, 1)
    legs = (("cpu_samplers", False), ("device_sampling", True)) if world == 1 else (("device_sampling", True),)
    for name, device_sampling in legs:
        solver, wall = train_timed(args, gv, graph, threads, partitions, device_sampling, epochs)
        timing = solver.timing
        out[name] = {"value": timing["batches"] * B / timing["episodes"] / 1e6, "unit": "million edge-samples/sec",
                     "batches": timing["batches"], "episode_seconds": timing["episodes"], "train_seconds": wall,
                     "sampler_threads_per_gpu": 0 if device_sampling else threads, "pair_order": solver.pair_order,
                     "episode_size": solver.episode_size, "transport": solver.transport,
                     "exchange_bytes_sent_per_gpu_per_step": (solver.exchange_stats["bytes_sent_per_gpu"] //
                                                              max(solver.exchange_stats["exchanges"], 1))}
        solver.clear()
        del solver
    return out


def link_prediction(args, gv, world, threads, partitions):
    """Link-prediction AUC of a training of THIS configuration — same graph generator and size, a held-out 1 % of the
    edges, the same number of workers and partitions, positives drawn on the device when there are several GPUs — next to
    the reference's own training loop on the same shape (tests/golden/reference_c2.npz: sequential kernel model, one
    partition, --auc-epochs 50)."""
    from graphvite_amd import synthetic
    edges = synthetic.power_law_edges(args.vertices, args.edges, seed=args.seed)
    train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
    graph = gv.graph.Graph()
    graph.load(train)
    episode = auc_episode_size(partitions) if partitions > 1 else None
    solver, wall = train_timed(args, gv, graph, threads, partitions, world > 1, args.auc_epochs, episode_size=episode)
    H, T, Y = (np.asarray(x) for x in test)
    name2id = np.full(args.vertices, -1, np.int64)
    names = np.array([int(x) for x in graph.id2name], np.int64)
    name2id[names] = np.arange(len(names))
    keep = (name2id[H] >= 0) & (name2id[T] >= 0)
    h, t, y = name2id[H[keep]], name2id[T[keep]], Y[keep]

    def auc_of(trained):
        score = np.einsum("ij,ij->i", trained.vertex_embeddings[h], trained.context_embeddings[t])
        ranked = y[np.argsort(-score, kind="stable")]
        return float(np.cumsum(ranked)[ranked == 0].sum()) / (int((ranked == 0).sum()) * int((ranked == 1).sum()))
    auc = auc_of(solver)
    out = {"value": auc, "epochs": args.auc_epochs, "batches": solver.batch_id, "workers": world, "partitions": solver.num_partition,
           "device_sampling": world > 1, "hub_rows": solver.hub_rows, "hub_parts": solver.hub_parts_used, "fidelity": solver.fidelity,
           "episode_size": solver.episode_size,
           "kernel": ("train_hot_kernel: hub rows by chains, a batch as %d parts" % solver.hub_parts_used) if solver.hub_rows else
                     solver.kernels.describe_train(args.dim, "SGD", args.negatives, False, args.batch, solver.partition_rows)}
    golden = os.path.join(ROOT, "tests", "golden", "reference_c2.npz")
    if os.path.exists(golden) and (args.vertices, args.edges, args.seed, args.batch) == (1000000, 10000000, 1024, 100000):
        G = np.load(golden)
        key = "c2_line_sequential" if solver.num_partition == 1 else "c2_line_p%d_e%d" % (solver.num_partition, episode or 0)
        reference = G[key] if key in G else np.zeros(0)
        reference = reference[~np.isnan(reference)]
        if len(reference) and int(G["c2_args"][5]) == args.auc_epochs:
            out["reference_training_loop"] = {"mean": float(reference.mean()), "seeds": len(reference),
                                              "note": "the reference's own GraphSolver::train on this shape, sequential kernel "
                                                      "model, one worker / %d partition(s) (tests/golden/make_c2_golden.py)" % solver.num_partition}
            out["difference"] = auc - float(reference.mean())
            # the reference's kernel is itself concurrent (<<<8192, 512>>>, instance/graph.cuh:487-490): its own training loop
            # under the two chunk-synchronous models of that launch on the card it was written for
            models = {name: float(G["c2_line_" + name][0]) for name in ("lock_step", "reads_at_start") if "c2_line_" + name in G}
            if models and solver.num_partition == 1:
                out["reference_training_loop"]["concurrent_models"] = models
    solver.clear()
    if world == 1 and partitions == 1 and args.hub_rows == "default" and args.fidelity != "throughput" and not args.no_fidelity_leg:
        # the same training pair by pair (GraphSolver(fidelity="throughput"): no chains, every row Hogwild as in the reference's
        # kernel): what the hub rows' lost updates cost on this shape
        plain, wall = train_timed(args, gv, graph, threads, partitions, False, args.auc_epochs, fidelity="throughput")
        value = auc_of(plain)
        timing = plain.timing
        out["fidelity_throughput"] = {"value": value, "hub_rows": plain.hub_rows, "batches": timing["batches"],
                                      "million_edge_samples_per_sec": timing["batches"] * args.batch / timing["episodes"] / 1e6,
                                      "train_seconds": wall}
        if "reference_training_loop" in out:
            out["fidelity_throughput"]["difference"] = value - out["reference_training_loop"]["mean"]
        plain.clear()
    return out


def module_leg(args, world, timeout=240):
    """The same training through the pybind11 module `libgraphvite` — the boundary a user of the reference's Python package
    loads — in a process of its own (one process, device_ids = all GPUs of the job: ncclCommInitAll inside the engine)."""
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "measure_engine.py"), "--bench", "--gpus", str(world),
           "--vertices", str(args.vertices), "--edges", str(args.edges), "--seed", str(args.seed), "--dim", str(args.dim),
           "--batches", str(args.end_to_end_batches)]
    env = dict(os.environ)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK"):
        env.pop(name, None)
    try:
        run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return {"error": "no result within %d s" % timeout}
    lines = [line for line in run.stdout.splitlines() if line.startswith("{")]
    if run.returncode != 0 or not lines:
        return {"error": "exit code %d: %s" % (run.returncode, run.stderr.strip()[-300:])}
    return json.loads(lines[-1])


def expected_curve():
    """The prediction an N-GPU run tests, carried by the line itself: N x the one-GPU rate at the shard size of an N-GPU run, from the
    newest committed `bench.py --partitions N` runs (profiles/r*/bench_by_partitions.jsonl) — what N GPUs deliver if nothing but their
    kernels limits them.  No N-GPU run has been measured so far: this is an expectation, not a curve."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_by_partitions.jsonl")), reverse=True):
        lines = [json.loads(l) for l in open(path) if l.startswith("{")]
        curve = {}
        for line in lines:
            parts = line.get("config", {}).get("shard", {}).get("rows")
            n = round(1000000 / parts) if parts else None
            if n in (2, 4, 8):
                curve[str(n)] = {"per_gpu_at_that_shard_size": line["value"], "expected_whole_job": n * line["value"]}
        one = sorted(glob.glob(os.path.join(os.path.dirname(path), "bench_n1*.json")))
        for name in one:
            rows = [json.loads(l) for l in open(name) if l.startswith("{")]
            if rows:
                curve["1"] = {"per_gpu_at_that_shard_size": rows[-1]["value"], "expected_whole_job": rows[-1]["value"]}
                break
        if curve:
            return {"unit": "million edge-samples/sec", "source": os.path.relpath(path, ROOT), "by_gpus": curve,
                    "note": "one-GPU runs of the shard sizes (bench.py --partitions N), times N; north_star asks for >= 6 x at 8 GPUs"}
    return None


def same_shards_on_one_gpu(args, world, timeout=300):
    """The one-GPU rate at the shard size of this N-GPU run (`bench.py --gpus 1 --partitions N`, the same steps, in a process
    of its own on rank 0's GPU while the other ranks wait): the tables of a block then live where they live in the N-GPU run
    (L2 / Infinity Cache from a few partitions on), so value / (N x this) separates scaling from the cache effect."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--partitions", str(world), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--vertices", str(args.vertices), "--edges", str(args.edges), "--dim", str(args.dim),
           "--batch", str(args.batch), "--negatives", str(args.negatives), "--seed", str(args.seed), "--fidelity", args.fidelity,
           "--block-batches", str(args.block_batches), "--no-end-to-end", "--no-cpu-baseline", "--no-module", "--no-access-pattern"]
    env = dict(os.environ)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK"):
        env.pop(name, None)
    env["HIP_VISIBLE_DEVICES"] = os.environ.get("HIP_VISIBLE_DEVICES", "").split(",")[0] or "0"
    try:
        run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %d s" % timeout}
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    if run.returncode != 0 or not lines:
        return {"error": "exit code %d: %s" % (run.returncode, run.stderr.strip()[-300:])}
    r = json.loads(lines[-1])
    return {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "partitions": world,
            "kernel": r["roofline"]["kernel"], "shard": r["config"]["shard"],
            "note": "bench.py --gpus 1 --partitions %d: the same shard size on ONE GPU" % world}


# True only inside tests/bench_dry_run.py (the loop's logic on the CPU over the host build of the engine; no command line
# and no environment variable of bench.py sets it)
DRY_RUN = False


def main(argv=None):
    args = parse(argv)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with `python -m torch.distributed.run "
                         "--nproc-per-node %d ... bench.py --gpus %d`" % (args.gpus, world, args.gpus, args.gpus))
    import graphvite_amd as gv
    from graphvite_amd import _lib, synthetic
    gv.init_logging(logging.ERROR)
    cuda = not DRY_RUN
    if cuda:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        args.no_cpu_baseline = args.no_end_to_end = args.no_module = args.no_access_pattern = True
        dev = torch.device("cpu")
    if world > 1:
        if cuda:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    N, E, B, k, dim = args.vertices, args.edges, args.batch, args.negatives, args.dim
    # the reference's minimum: one partition per GPU (solver.h:269-276).  More partitions than GPUs (P = 2 x #GPU lets the
    # all-gather of one head group overlap the training on the other) is --partitions.
    partitions = args.partitions or world
    steps_requested = args.steps
    if not args.block_batches:
        auto = max(int(float(N) * 175 / partitions / B), 1)  # the reference's episode size, solver.h:426-436
        if world == 1:
            if partitions == 1:
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
    solver = gv.solver.GraphSolver(dim, num_sampler_per_worker=threads, seed=args.seed, fidelity=args.fidelity,
                                   pair_order=gv.auto if args.pair_order == "auto" else args.pair_order,
                                   hub_rows=None if args.hub_rows == "default" else (args.hub_rows if args.hub_rows == "auto" else int(args.hub_rows)))
    solver.hub_parts = args.hub_parts
    solver.hub_lerp = None if args.hub_lerp < 0 else bool(args.hub_lerp)
    solver.hub_chain_cap = args.hub_cap
    solver.negative_table = args.negative_table
    for item in args.tune:
        key, value = item.split("=")
        solver.kernels.set_tuning(int(key), int(value))
    optimizer = gv.optimizer.SGD(0.025, 0.005, "linear") if args.optimizer == "SGD" else \
        gv.optimizer.Optimizer(args.optimizer, 1e-3, 0.005)
    solver.build(graph, optimizer=optimizer, num_partition=partitions, num_negative=k,
                 batch_size=B, episode_size=args.block_batches)
    blocks_per_rank = partitions * partitions // world
    residency = 2 * blocks_per_rank  # two batches of every block before the warm-up steps
    total_batches = (residency + args.warmup + args.steps) * world
    epochs = total_batches * B // graph.num_edge + 1  # the lr schedule spans about the batches this run trains
    session = solver.session(resident_pools=True, model="LINE", num_epoch=epochs, augmentation_step=1, log_frequency=1 << 30)
    grouped = solver.pair_order == "grouped"
    t0 = time.perf_counter()
    session.fill(0)  # every block pool of this GPU's column: CPU edge sampler, then resident in HBM
    fill_s = time.perf_counter() - t0
    sampled = blocks_per_rank * args.block_batches * B
    stream = torch.cuda.ExternalStream(session.stream(0), device=dev) if cuda else None
    copy_events = []  # (start, end) of every regrouping pass staged while timing

    def event():
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    kernel_events = []
    # The walk over the schedule continues across the residency pass, the warm-up and the timed steps, block by block
    # as the episode loop walks it: block_batches batches of a block, the exchange, the next block — whose pool was
    # staged (regrouped) while this one trained.  A run of K steps simply trains the next K batches of that walk.
    walk = {"visit": 0, "offset": 0, "staged": False}
    timing = {"on": False}

    def stage(visit):
        t = time.perf_counter()
        session.stage(visit % session.steps, 0, visit & 1)
        if timing["on"] and grouped:
            copy_events.append(time.perf_counter() - t)

    def run(num_batches, timed, leave_block=False):
        done = 0
        while done < num_batches:
            visit = walk["visit"]
            if walk["offset"] == 0:  # entering a block
                if not walk["staged"]:
                    stage(visit)
                stage(visit + 1)  # the next visit's pool, while this one trains
                walk["staged"] = True
            n = min(args.block_batches - walk["offset"], num_batches - done)
            e0 = event() if timed and cuda else None
            session.train(visit % session.steps, 0, visit & 1, walk["offset"], n)
            if e0 is not None:
                kernel_events.append((e0, event(), n))
            done += n
            walk["offset"] += n
            if walk["offset"] == args.block_batches or (leave_block and done == num_batches):
                session.exchange(visit % session.steps)
                walk.update(visit=visit + 1, offset=0)

    def fence():
        session.wait()
        session.synchronize()
        if world > 1:
            dist.barrier()
        session.synchronize()

    # Residency pass before the W warm-up steps: two batches of every block (code-object load, first touch of every
    # table / pool, runtime pools growing) and the first collective (RCCL communicator + buffers).  One-time costs of
    # tens of milliseconds otherwise land inside a timed region that is itself only tens of milliseconds long.
    for _ in range(blocks_per_rank):
        run(min(2, args.block_batches), False, leave_block=True)
    fence()
    # The ceiling of the access pattern is measured here, ahead of the warm-up: 2 x 200 launches that read and write back
    # the rows of the pools' batches, tables unchanged.
    probe = None
    if cuda and k == 1 and optimizer.num_moment == 0 and not args.no_access_pattern:
        stage(walk["visit"])
        walk["staged"] = True
        ms = session.probe(walk["visit"] % session.steps, 0, walk["visit"] & 1, 200)
        achieved = algorithmic_bytes(dim, 1) * B / (ms * 1e-3)
        probe = {"kernel": "probe_rows_kernel: the rows of a batch read and written back, nothing else", "kernel_ms": ms,
                 "achieved": achieved / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "launches": 200}
    # the warm-up ends on a block boundary: the timed region then consists of whole block visits; as in the steady
    # state of the episode loop every visit stages (regroups) the pool of the NEXT visit while it trains, so the
    # region holds exactly one staging pass and one exchange per visit
    run(args.warmup, False, leave_block=True)
    fence()
    before = solver._exchange_stats()
    timing["on"] = True
    t0 = time.perf_counter()
    run(args.steps, True)
    fence()
    wall = time.perf_counter() - t0
    timing["on"] = False
    after = solver._exchange_stats()
    # N > 1: one exchange on its own, nothing else in flight (fence before and after, wall clock, max over ranks): next to the
    # kernels' time per visit it says whether a visit is kernel-limited or fabric-limited
    exchange_ms = None
    if world > 1:
        isolated = []
        for _ in range(3):
            t1 = time.perf_counter()
            session.exchange(walk["visit"] % session.steps)
            fence()
            isolated.append(time.perf_counter() - t1)
        t = torch.tensor([min(isolated)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exchange_ms = float(t.item()) * 1e3
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if cuda:
        kernel_ms = sum(a.elapsed_time(b) for a, b, _ in kernel_events) / sum(n for _, _, n in kernel_events)
    else:
        kernel_ms = wall / args.steps * 1e3  # dry run: nothing to measure
    final_loss = session.loss(0)
    visits = -(-args.steps // args.block_batches)
    collectives = after["exchanges"] - before["exchanges"]
    moments = optimizer.num_moment
    # With hub rows trained by chains a batch is `launches` launches of train_hot_kernel, each the pairs of one part of the
    # batch (and the chains of the next part): the roofline is stated per launch, as for the one-launch-per-batch kernels
    launches = max(solver.hub_parts_used, 1) if solver.hub_rows else 1
    kernel_ms_per_batch = kernel_ms  # HIP events bracket whole batches: every launch of a batch, its work lists and mirror copies
    kernel_ms /= launches
    bytes_per_launch = (8 * dim * (k + 2) * (1 + moments) + 16) * B // launches  # moment tables are rows read + written too
    achieved = bytes_per_launch / (kernel_ms * 1e-3)
    # HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM); bench.py cannot read PMCs itself
    traffic, pmc_path = None, None
    same_job = bool(os.environ.get("GVK_BENCH_PMC_SUMMARY"))
    if world == 1 and partitions == 1 and dim == 128 and k == 1 and B == 100000 and N == 1000000 and moments == 0:
        import glob
        wanted = "train_hot_kernel" if solver.hub_rows else "train_kernel"
        candidates = [os.environ["GVK_BENCH_PMC_SUMMARY"]] if same_job else \
            sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary_bench_n1.json")), reverse=True)
        for path in candidates:
            summary = json.load(open(path))  # the committed PMC passes of this command — of the kernel this run launched only
            if wanted in summary.get("kernel", "") and summary.get("launches_per_batch", 1) == launches:
                traffic, pmc_path = summary.get("traffic_bytes_per_launch"), os.path.relpath(path, ROOT)
                break
    rows = solver.partition_rows
    kernel_name = solver.kernels.describe_train(dim, args.optimizer, k, False, B, rows)
    if solver.hub_rows:  # gvk_train_episode_hot: the pairs of a part of a batch + the chains of the next part's hub rows in one launch
        kernel_name = "train_hot_kernel<%d>: the pairs of %d samples (a batch as %d parts) + the chains of the next part over %d hub rows per table%s" % (
            dim, B // launches, launches, solver.hub_rows, ", hub rows read along the chains' way (lerp)" if solver.hub_lerp_used else "")
    shard_bytes = rows * dim * 4 * (1 + moments)
    residency_note = ("both tables of a block fit the 32 MB of L2" if 2 * shard_bytes <= L2_BYTES else
                      "both tables of a block fit the 256 MB Infinity Cache: the kernel is served by the cache, not by HBM — "
                      "compare with the one-GPU rate at the same shard size (bench.py --partitions), not with the 512 MB "
                      "tables of the one-partition run" if 2 * shard_bytes <= INFINITY_CACHE_BYTES else
                      "tables larger than the caches: HBM-bound")
    solver._refresh()
    prefetched = int(getattr(solver, "lists_prefetched", 0))
    work_lists_note = ("%d visit(s) of this process trained lists built while the visit before them trained; a timed visit builds the next visit's "
                       "inside the timed region" % prefetched) if prefetched else "built when a visit begins"
    result = {
        "metric": "million edge-samples/sec at dim=%d" % dim,
        "value": world * args.steps * B / wall / 1e6,
        "unit": "million edge-samples/sec",
        "n_gpus": world, "steps": args.steps, "steps_requested": steps_requested, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if cuda else "DRY RUN on the CPU with the host build of the engine: logic test, not a result",
        "config": {"workload": "LINE (augmentation_step 1) on synthetic power-law %d nodes / %d edges, dim %d, "
                               "batch %d edge-samples per GPU per step, num_negative %d, SGD lr 0.025 wd 0.005 linear, "
                               "negatives drawn in-kernel, block pools resident in HBM" % (N, E, dim, B, k),
                   "parallelism": "%d GPU(s), one process each, %d vertex partition(s), context shards pinned per GPU, one "
                                  "in-place all-gather of a head group's slab every %d batches (%s)"
                                  % (world, partitions, args.block_batches, solver.transport or "one worker: none"),
                   "engine": "native solver engine (include/gvx.h) stepped through its session API",
                   "block_batches": args.block_batches, "block_visits_timed": visits,
                   # the work lists of a visit's first chunk are built while the visit before it trains (gvx_engine.cpp prefetch_lists): every timed visit
                   # builds the lists of the visit after it inside the timed region, as the steady state of the episode loop does
                   "work_lists": work_lists_note,
                   "shard": {"rows": rows, "table_bytes": shard_bytes, "residency": residency_note},
                   "negative_table": solver.negative_table,
                   "pair_order": solver.pair_order + (" (gvk_group_pairs once per block visit on the copy stream)"
                                                      if grouped else "")},
        "regroup": {"passes_in_timed_region": len(copy_events),
                    "note": "runs on the copy stream concurrently with the previous block's kernels; ms_per_step "
                            "(wall) already contains whatever of it was not hidden"} if grouped else None,
        "exchange": {"collectives_timed": collectives,
                     "bytes_sent_per_gpu_per_collective": (after["bytes_sent_per_gpu"] - before["bytes_sent_per_gpu"]) // max(collectives, 1),
                     "transport": solver.transport,
                     "isolated_ms": exchange_ms,  # one exchange with nothing else in flight (it overlaps the next visit's kernels in the timed region)
                     "kernels_ms_per_visit": kernel_ms_per_batch * args.block_batches,
                     "note": "one in-place ncclAllGather of a head group's slab per schedule step"}
        if world > 1 else None,
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK,
                     # a PMC pass cannot run inside this process: `traffic` is a number only when the job that runs this command also ran
                     # the PMC passes and says so (GVK_BENCH_PMC_SUMMARY=<the summary it wrote>); otherwise null, with the committed
                     # passes of the same kernel and launch count beside it
                     "traffic": traffic if same_job else None,
                     "traffic_profiled": None if traffic is None or same_job else {"bytes_per_launch": traffic, "source": pmc_path,
                                         "note": "rocprofv3 --pmc passes of this command on another box (profiles/), not of this run"},
                     "traffic_source": pmc_path if same_job else None,
                     "kernel": kernel_name, "kernel_ms": kernel_ms, "launches_per_step": launches,
                     "algorithmic_bytes_per_launch": bytes_per_launch},
        "sampler": {"value": sampled / fill_s / 1e6, "unit": "million edge-samples/sec per GPU", "threads": threads,
                    "note": "CPU edge sampler filling this GPU's block pools (+ their upload) before the timed region"},
        "final_batch_mean_loss": final_loss,
    }
    if probe is not None:  # how close the training kernel is to what the memory system sustains for ITS access pattern
        result["roofline"]["access_pattern"] = probe
        probe["train_kernel_vs_probe"] = probe["kernel_ms"] / (kernel_ms * launches)  # per batch, both
    session.close()
    solver.clear()
    del session, solver
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, graph)
    if cuda and not args.no_end_to_end and args.optimizer == "SGD":
        # every rank takes part in these trainings; an error on one rank would leave the others in a collective, so
        # they run after the headline measurement is complete and are reported as they are
        try:
            result["end_to_end"] = end_to_end(args, gv, graph, world, threads, partitions)
            result["auc"] = link_prediction(args, gv, world, threads, partitions)
        except Exception as error:  # the headline measurement above stands on its own; say what happened to these
            result.setdefault("end_to_end", {})["error"] = "%s: %s" % (type(error).__name__, error)
        if world > 1:
            dist.barrier()
        if rank == 0 and not args.no_module:
            torch.cuda.empty_cache()
            result["end_to_end"]["module"] = module_leg(args, world)
    if world > 1 and cuda:
        if rank == 0:
            result["expected_curve"] = expected_curve()
            same = result["single_gpu_same_shards"] = same_shards_on_one_gpu(args, world)
            if "value" in same:  # what N GPUs deliver if nothing but their kernels limits them, and what was measured
                expected = world * same["value"]
                kernels, exchange = result["exchange"]["kernels_ms_per_visit"], result["exchange"]["isolated_ms"]
                result["scaling_readout"] = {
                    "expected_if_kernel_limited": expected, "measured": result["value"], "measured_over_expected": result["value"] / expected,
                    "limited_by": "kernels" if result["value"] >= 0.9 * expected else
                                  ("fabric: one exchange takes longer than a visit's kernels" if exchange and exchange > kernels else
                                   "neither alone: imbalance between ranks, host pacing or exchanges that did not overlap"),
                    "note": "expected = N x single_gpu_same_shards.value (the same shard size on ONE GPU: no exchange, no imbalance)"}
        dist.barrier()
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
