"""BASELINE configs[4] as a RUN on one MI355X: LINE, dim 96, on a Friendster-scale synthetic graph — 65M nodes (the real
graph's node count, config/graph/line_friendster.yaml), as many edges as the box's host memory comfortably holds (--edges,
default 2e8: the real graph has 1.8e9; say so next to the number) — cut into the 8 partitions of the 8-GPU configuration and
trained for a few episodes on ONE GPU with the positives drawn on the device: every table of the 8-GPU run exists (8 head
partitions of 8.2M rows = 3.1 GB each, 8 context shards, 64 block edge tables), so the run exercises what that scale adds —
64-bit offsets in upload / write-back (the vertex table alone is 25 GB), partitions of more than 2^23 rows, the episode-size
arithmetic the reference does in 32 bits (num_vertex * kSamplePerVertex = 1.1e10, include/core/solver.h:429-431), chunked
table moves — and gives a whole-path rate for the configuration.

    python scripts/measure_friendster.py [--vertices 65000000] [--edges 200000000] [--episodes 2] >> profiles/r3/configs_4.jsonl
"""
import argparse
import json
import logging
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphvite_amd as gv  # noqa: E402
from graphvite_amd import synthetic  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--vertices", type=int, default=65000000)
    p.add_argument("--edges", type=int, default=200000000)
    p.add_argument("--partitions", type=int, default=8)
    p.add_argument("--episodes", type=int, default=2)
    p.add_argument("--episode-size", type=int, default=250,
                   help="batches per block pool; the reference's automatic size for this graph (65M * 175 / 8 / 100 000 = 14 218 "
                        "batches = 11 GB per block pool) does not fit any GPU and is halved until it does (solver.h:437-455); with the "
                        "positives drawn on the device every block pool of an episode is resident twice, so the size is given")
    p.add_argument("--dim", type=int, default=96)
    p.add_argument("--cpu-samplers", action="store_true", help="CPU sampler threads instead of device-side sampling")
    args = p.parse_args()
    gv.init_logging(logging.WARNING)
    t0 = time.perf_counter()
    edges = synthetic.power_law_edges(args.vertices, args.edges, seed=2026)
    t1 = time.perf_counter()
    graph = gv.graph.Graph()
    graph.load(edges)
    del edges
    t2 = time.perf_counter()
    solver = gv.solver.GraphSolver(args.dim, device_sampling=not args.cpu_samplers, seed=1)
    solver.build(graph, optimizer=gv.optimizer.SGD(0.025, 0.005), num_partition=args.partitions, num_negative=1, batch_size=100000,
                 episode_size=args.episode_size)
    per_episode = solver.num_partition ** 2 * solver.episode_size
    epochs = max(args.episodes * per_episode * solver.batch_size // graph.num_edge, 1)
    t3 = time.perf_counter()
    solver.train(model="LINE", num_epoch=epochs, augmentation_step=1, log_frequency=1 << 30)
    t4 = time.perf_counter()
    timing = solver.timing
    v, c = solver.vertex_embeddings, solver.context_embeddings
    sample = np.random.default_rng(0).integers(0, graph.num_vertex, 100000)
    out = {"config": "configs[4]: LINE dim %d, %d nodes / %d edges (synthetic power-law; Friendster: 65.6M / 1.8e9), %d partitions "
                     "on one GPU" % (args.dim, graph.num_vertex, graph.num_edge, solver.num_partition),
           "device_sampling": solver.device_sampling, "episode_size": solver.episode_size, "partition_rows": solver.partition_rows,
           "table_bytes_per_partition": solver.partition_rows * args.dim * 4, "vertex_table_bytes": graph.num_vertex * args.dim * 4,
           "gpu_memory_cost": solver.gpu_memory_cost, "batches": timing["batches"], "episodes": timing["batches"] / per_episode,
           "million_edge_samples_per_sec": timing["batches"] * solver.batch_size / timing["episodes"] / 1e6,
           "episode_loop_seconds": timing["episodes"], "train_seconds": t4 - t3, "generate_seconds": t1 - t0, "load_seconds": t2 - t1,
           "build_seconds": t3 - t2, "kernel": solver.kernels.describe_train(args.dim, "SGD", 1, False, solver.batch_size, solver.partition_rows),
           "pair_order": solver.pair_order,
           "finite": bool(np.isfinite(v[sample]).all() and np.isfinite(c[sample]).all()),
           "context_rows_trained_of_sample": float((np.abs(c[sample]).max(1) > 0).mean()),
           "last_row_trained": bool(np.abs(c[-1000:]).max() > 0 or np.abs(v[-1000:]).max() > 0)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
