"""BASELINE.json configs[2..4] where they fit one GPU (run on the MI355X box; summaries go to profiles/<round>/):

  configs[2]  DeepWalk on a Youtube-sized graph (1 138 499 nodes / 4 945 382 edges, config/graph/deepwalk_youtube.yaml:
              augmentation_step 5, walk length 40, batch 100 000, episode 500), end to end through GraphSolver.train():
              CPU samplers feeding the GPU, and positives drawn on the device;
  configs[3]  node2vec p = q = 0.25 on the same graph (config/graph/node2vec_youtube.yaml): per-edge alias tables when
              they fit the limit, rejection sampling otherwise, device sampling; and with 4 partitions on the one GPU —
              the per-GPU shape of the 4-GPU run (blocks of a quarter of the table, walks binned per block);
  configs[4]  the dim-96 kernel on one Friendster shard (8.2M rows = 65M / 8, config/graph/line_friendster.yaml) with
              its roofline fraction: bench.py --dim 96 --vertices 8200000.

    python scripts/measure_configs.py [--epochs 200] > gpurun_out/configs_2_4.jsonl
"""
import argparse
import json
import logging
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphvite_amd as gv  # noqa: E402
from graphvite_amd import synthetic  # noqa: E402


def run(graph, name, model, epochs, threads, **kw):
    import torch
    solver_kw = {k: kw.pop(k) for k in ("device_sampling", "num_partition", "table_limit") if k in kw}
    s = gv.solver.GraphSolver(128, num_sampler_per_worker=threads, device_sampling=solver_kw.get("device_sampling", False))
    if "table_limit" in solver_kw:
        s.node2vec_table_limit = solver_kw["table_limit"]
    s.build(graph, optimizer=gv.optimizer.SGD(0.025, 0.005), num_partition=solver_kw.get("num_partition", gv.auto),
            num_negative=1, batch_size=100000, episode_size=500)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.train(model=model, num_epoch=epochs, negative_weight=5, augmentation_step=5, random_walk_length=40,
            random_walk_batch_size=100, log_frequency=1 << 30, **kw)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    t = s.timing
    out = {"config": name, "model": model, "mode": s._mode, "pair_order": s.pair_order, "num_partition": s.num_partition,
           "episode_size": s.episode_size, "device_sampling": s.device_sampling,
           "sampler_threads": 0 if s.device_sampling else threads, "batches": t["batches"],
           "million_edge_samples_per_sec": t["batches"] * 1e5 / t["episodes"] / 1e6, "episode_seconds": t["episodes"],
           "train_seconds": wall, "context_norm": float(np.abs(s.context_embeddings).mean())}
    print(json.dumps(out), flush=True)
    s.clear()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--epochs", type=int, default=200)
    p.add_argument("--skip-friendster", action="store_true")
    args = p.parse_args()
    gv.init_logging(logging.ERROR)
    from graphvite_amd.base import cpu_budget
    threads = max(cpu_budget() - 1, 1)
    graph = gv.graph.Graph()
    t0 = time.perf_counter()
    graph.load(synthetic.power_law_edges(1138499, 4945382, seed=2024))
    print(json.dumps({"config": "graph", "note": "synthetic power-law stand-in for Youtube", "num_vertex": graph.num_vertex,
                      "num_edge": graph.num_edge, "load_seconds": time.perf_counter() - t0}), flush=True)
    e = args.epochs
    run(graph, "configs[2]", "DeepWalk", e, threads)
    run(graph, "configs[2]", "DeepWalk", e, threads, device_sampling=True)
    run(graph, "configs[3]", "node2vec", e, threads, p=0.25, q=0.25)
    run(graph, "configs[3]", "node2vec", e, threads, p=0.25, q=0.25, table_limit=1)         # forced rejection sampling
    run(graph, "configs[3]", "node2vec", e, threads, p=0.25, q=0.25, device_sampling=True)
    run(graph, "configs[3] per-GPU shape of the 4-GPU run", "node2vec", e, threads, p=0.25, q=0.25, num_partition=4)
    # (an episode is 16 blocks x 500 batches here: 5 x the epochs = 7 episodes instead of 2, whose first pool nothing overlaps)
    run(graph, "configs[3] per-GPU shape of the 4-GPU run", "node2vec", 5 * e, threads, p=0.25, q=0.25, num_partition=4,
        device_sampling=True)
    run(graph, "configs[2] over 4 partitions on the one GPU", "DeepWalk", 5 * e, threads, num_partition=4, device_sampling=True)
    del graph
    if not args.skip_friendster:
        for order in ("sampled", "grouped"):
            line = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dim", "96", "--vertices", "8200000",
                                   "--edges", "41000000", "--no-cpu-baseline", "--no-end-to-end", "--pair-order", order],
                                  capture_output=True, text=True).stdout.strip().splitlines()[-1]
            r = json.loads(line)
            print(json.dumps({"config": "configs[4]: one Friendster shard (8.2M rows), dim 96, LINE kernel", "pair_order": order,
                              "million_edge_samples_per_sec": r["value"], "kernel_ms": r["roofline"]["kernel_ms"],
                              "roofline_frac": r["roofline"]["frac"], "kernel": r["roofline"]["kernel"],
                              "regroup": r.get("regroup")}), flush=True)


if __name__ == "__main__":
    main()
