"""End-to-end training rate of the NATIVE engine (include/gvx.h) through the pybind11 module `libgraphvite`, the way a
user of the reference's Python package runs it — one process, device_ids=[0] — on the graph shapes of BASELINE.json
(synthetic stand-ins): LINE / DeepWalk / node2vec on a Youtube-sized graph, CPU sampler threads vs positives drawn on
the device (GraphSolver(..., device_sampling=True)), one partition and the 4-partition per-GPU shape.

    python scripts/measure_engine.py [--epochs 100] > gpurun_out/engine_e2e.jsonl

--bench (what bench.py's `end_to_end.module` leg runs in a process of its own): LINE on bench.py's own graph (synthetic
power-law, written to an edge-list file and loaded with the module's Graph.load(file_name) like a user's dataset), one
process, device_ids = [0 .. gpus - 1] — with several GPUs the engine creates its RCCL communicators with ncclCommInitAll —
positives drawn on the device; prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graphvite_amd", "lib"))
from graphvite_amd import synthetic  # noqa: E402
from graphvite_amd.base import cpu_budget  # noqa: E402
import libgraphvite as lib  # noqa: E402


def run(graph, name, model, epochs, threads, device_sampling, num_partition=lib.auto, dim=128, **kw):
    solver = getattr(lib.solver, "GraphSolver_%d_f_j" % dim)(device_ids=[0], num_sampler_per_worker=threads,
                                                            device_sampling=device_sampling)
    solver.build(graph, lib.optimizer.SGD(0.025, 0.005), num_partition=num_partition, num_negative=1, batch_size=100000)
    t0 = time.perf_counter()
    solver.train(model=model, num_epoch=epochs, negative_weight=5, log_frequency=1 << 30, **kw)
    wall = time.perf_counter() - t0
    out = {"config": name, "model": model, "dim": dim, "num_partition": solver.num_partition,
           "episode_size": solver.episode_size, "augmentation_step": solver.augmentation_step,
           "device_sampling": device_sampling, "sampler_threads": 0 if device_sampling else threads,
           "batches": solver.batch_id, "episode_loop_seconds": solver.train_seconds,
           "million_edge_samples_per_sec": solver.batch_id * 1e5 / solver.train_seconds / 1e6, "train_seconds": wall}
    print(json.dumps(out), flush=True)
    solver.clear()


def bench(args):
    import tempfile
    edges = synthetic.power_law_edges(args.vertices, args.edges, seed=args.seed)
    with tempfile.TemporaryDirectory() as directory:
        path = os.path.join(directory, "edges.txt")
        try:
            import pandas
            pandas.DataFrame(edges).to_csv(path, sep=" ", header=False, index=False)
        except ImportError:
            import numpy
            numpy.savetxt(path, edges, fmt="%d")
        graph = lib.graph.Graph_j()
        t0 = time.perf_counter()
        graph.load(path)
        load_seconds = time.perf_counter() - t0
    batch = 100000
    solver = getattr(lib.solver, "GraphSolver_%d_f_j" % args.dim)(device_ids=list(range(args.gpus)), device_sampling=True,
                                                                 seed=args.seed)
    solver.build(graph, lib.optimizer.SGD(0.025, 0.005), num_negative=1, batch_size=batch)
    epochs = max(args.batches * args.gpus * batch // graph.num_edge, 1)
    t0 = time.perf_counter()
    solver.train(model="LINE", num_epoch=epochs, augmentation_step=1, negative_weight=5, log_frequency=1 << 30)
    wall = time.perf_counter() - t0
    print(json.dumps({"value": solver.batch_id * batch / solver.train_seconds / 1e6, "unit": "million edge-samples/sec",
                      "binding": "libgraphvite.solver.GraphSolver_%d_f_j(device_ids=%s, device_sampling=True)" % (
                          args.dim, list(range(args.gpus))),
                      "batches": solver.batch_id, "episode_seconds": solver.train_seconds, "train_seconds": wall,
                      "num_worker": solver.num_worker, "num_partition": solver.num_partition,
                      "episode_size": solver.episode_size, "graph_load_seconds": load_seconds}), flush=True)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--quick", action="store_true", help="LINE on one partition only (profiling runs)")
    p.add_argument("--bench", action="store_true", help="bench.py's `end_to_end.module` leg (see the docstring)")
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--vertices", type=int, default=1000000)
    p.add_argument("--edges", type=int, default=10000000)
    p.add_argument("--seed", type=int, default=1024)
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--batches", type=int, default=36000)
    args = p.parse_args()
    lib.init_logging(lib.ERROR)
    if args.bench:
        return bench(args)
    threads = max(cpu_budget() - 1, 1)
    graph = lib.graph.Graph_j()
    t0 = time.perf_counter()
    edges = synthetic.power_law_edges(1138499, 4945382, seed=2024)
    graph.load([(str(u), str(v)) for u, v in edges.tolist()])
    print(json.dumps({"config": "graph", "note": "synthetic power-law stand-in for Youtube", "num_vertex": graph.num_vertex,
                      "num_edge": graph.num_edge, "load_seconds": time.perf_counter() - t0}), flush=True)
    e = args.epochs
    walk = dict(augmentation_step=5, random_walk_length=40, random_walk_batch_size=100)
    for sampling in (False, True):
        run(graph, "configs[1]", "LINE", e, threads, sampling, augmentation_step=1)
        if args.quick:
            continue
        run(graph, "configs[2]", "DeepWalk", e, threads, sampling, **walk)
        run(graph, "configs[3]", "node2vec", e, threads, sampling, p=0.25, q=0.25, **walk)
        run(graph, "configs[1] over 4 partitions", "LINE", e, threads, sampling, num_partition=4, augmentation_step=1)
        run(graph, "configs[3] over 4 partitions", "node2vec", e, threads, sampling, num_partition=4, p=0.25, q=0.25, **walk)


if __name__ == "__main__":
    main()
