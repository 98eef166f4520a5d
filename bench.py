#!/usr/bin/env python
"""bench.py — million edge-samples/sec of the LINE dim-128 SGD hot path on N MI355X GPUs.

A "step" is one batch (100 000 edge-samples, num_negative 1) of negative-sampling SGD per GPU on the synthetic
power-law graph BASELINE.json's metric is quoted on (configs[1]: 1M nodes / 10M edges, dim 128, fp32, LINE),
with the episode's sample pool, the alias table and both embedding tables resident in HBM when the timed
region starts.  Negatives are drawn inside the kernel; the learning-rate schedule is applied per batch.

    python bench.py [--steps K] [--warmup W]                                                     (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (N > 1)

Rank 0 prints ONE JSON line.  `roofline` is for the training kernel (HBM-bound): achieved = algorithmic
bytes per launch (3088 B per edge-sample at dim 128, k = 1) / average launch duration measured
with HIP events on the launch stream over the timed region.  `cpu_baseline` times the reference's own host-
compiled arithmetic (oracle/_ref, Hogwild over all host cores) on a bounded sample of the same batches.
"""
import argparse
import json
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


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=400)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--vertices", type=int, default=1000000)
    p.add_argument("--edges", type=int, default=10000000)
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--batch", type=int, default=100000)
    p.add_argument("--negatives", type=int, default=1)
    p.add_argument("--pool-batches", type=int, default=200, help="batches in the HBM-resident sample pool")
    p.add_argument("--lanes", type=int, default=0, help="A/B knob: lanes per pair (0 = per-dim default)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=3.0, help="wall seconds given to the CPU baseline")
    p.add_argument("--seed", type=int, default=1024)
    return p.parse_args()


def cpu_baseline(args, vertex, context, pool, negs):
    """Reference arithmetic (oracle/_ref/libgvref_fast.so), Hogwild over all host cores, on the first batches
    of the same pool.  Test/bench infrastructure: the product never loads it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Reference
    try:
        ref = Reference(fast=True)
        kind = "reference"
    except (FileNotFoundError, OSError):
        return None
    cores = os.cpu_count() or 1
    v, c = vertex.copy(), context.copy()
    done, t0 = 0, time.perf_counter()
    nb = pool.shape[0] // args.batch
    while True:
        b = done % nb
        ref.train_mt(v, c, pool[b * args.batch:(b + 1) * args.batch], negs[b % len(negs)], 0.025, 0.005, 5.0, cores)
        done += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds and done >= 2:
            break
    return {"value": done * args.batch / el / 1e6, "unit": "million edge-samples/sec", "cores": cores, "kind": kind,
            "sample": "%d batches of %d edge-samples of the same pool (%.1f s wall, %d threads, Hogwild, "
                      "-Ofast x86-64-v3 host build of the reference's LINE::forward/backward + sgd_update)"
                      % (done, args.batch, el, cores)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from graphvite_amd import kernels as K
    from graphvite_amd import synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if world > 1:
        raise SystemExit("multi-GPU bench path lands with the episode-partitioned solver")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    hip = K.HipKernels()
    if args.lanes:
        hip.set_lanes_per_pair(args.lanes)

    # ---- synthetic workload (host) ----
    N, E, B, k, dim = args.vertices, args.edges, args.batch, args.negatives, args.dim
    edges = synthetic.power_law_edges(N, E, seed=args.seed)
    deg = synthetic.degrees(edges, N)
    # single partition: local id = rank in (degree desc, id asc) (SolverMixin::partition, solver.h:873-887)
    order = np.lexsort((np.arange(N), -deg))
    local = np.empty(N, np.uint32)
    local[order] = np.arange(N, dtype=np.uint32)
    rng = np.random.default_rng(args.seed + 1)
    # LINE, augmentation_step 1: edges drawn with probability ∝ weight (unit weights -> uniform over the 2E
    # directed edges); records are {tail, head}
    n_pool = args.pool_batches * B
    pick = rng.integers(0, E, n_pool)
    flip = rng.integers(0, 2, n_pool).astype(bool)
    head = np.where(flip, edges[pick, 1], edges[pick, 0])
    tail = np.where(flip, edges[pick, 0], edges[pick, 1])
    pool = np.stack([local[tail], local[head]], 1).astype(np.uint32)
    neg_w = (deg[order] ** 0.75).astype(np.float32)  # solver.h:1264-1278, in partition (local id) order
    prob, alias, packed = K.alias_build(neg_w)
    vertex = rng.uniform(-0.5 / dim, 0.5 / dim, (N, dim)).astype(np.float32)  # graph.cuh:724-731
    context = np.zeros((N, dim), np.float32)

    # ---- HBM-resident state ----
    t_vertex, t_context = torch.from_numpy(vertex).to(dev), torch.from_numpy(context).to(dev)
    t_pool = torch.from_numpy(pool.view(np.int32)).to(dev)
    t_table = K.packed_to_device(packed, dev)
    t_loss = torch.zeros(B, device=dev)
    spec = K.OptimizerSpec("SGD", 0.025, 0.005, schedule="linear")
    total_batches = args.warmup + args.steps

    def run(first, count):
        done = 0
        while done < count:
            start = (first + done) % args.pool_batches
            n = min(count - done, args.pool_batches - start)
            hip.train_episode(t_vertex, t_context, t_pool[start * B:], t_loss, spec, k, 5.0, t_table, args.seed,
                              first + done, total_batches, n, B)
            done += n

    run(0, args.warmup)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.warmup, args.steps)
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # launches are back to back on this stream
    final_loss = float(t_loss.mean().item())

    samples = args.steps * B
    bytes_per_launch = algorithmic_bytes(dim, k) * B
    achieved = bytes_per_launch / (kernel_ms * 1e-3)
    result = {
        "metric": "million edge-samples/sec at dim=%d" % dim,
        "value": samples / wall / 1e6,
        "unit": "million edge-samples/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "LINE on synthetic power-law %d nodes / %d edges, dim %d, batch %d, num_negative %d, "
                               "SGD lr 0.025 wd 0.005 linear, on-device negatives, pool resident in HBM"
                               % (N, E, dim, B, k),
                   "parallelism": "1 GPU, 1 partition", "lanes_per_pair": args.lanes or "default"},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": None, "kernel": "train_kernel<128,16,SGD>",
                     "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch},
        "final_batch_mean_loss": final_loss,
    }
    if not args.no_cpu_baseline:
        # negatives for the CPU run: drawn from the same alias table (numpy, vectorised)
        negs = []
        for _ in range(4):
            idx = rng.integers(0, N, (B, k))
            u = rng.random((B, k)).astype(np.float32)
            negs.append(np.where(u < prob[idx], idx, alias[idx]).astype(np.uint32))
        result["cpu_baseline"] = cpu_baseline(args, vertex, context, pool, negs)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
