"""Rounds of gvk_sample_walks_blocks per episode and the end-to-end rate of device-sampled walks over 4 partitions on
one GPU (the per-GPU shape of configs[3]); repeated, to see the run-to-run spread of a two-episode run."""
import logging
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import graphvite_amd as gv  # noqa: E402
from graphvite_amd import synthetic  # noqa: E402

gv.init_logging(logging.ERROR)
graph = gv.graph.Graph()
graph.load(synthetic.power_law_edges(1138499, 4945382, seed=2024))
for model, kw in (("DeepWalk", {}), ("node2vec", dict(p=0.25, q=0.25))) * 2:
    for epochs in (200, 1000):
        s = gv.solver.GraphSolver(128, device_sampling=True)
        s.build(graph, optimizer=gv.optimizer.SGD(0.025, 0.005), num_partition=4, num_negative=1, batch_size=100000,
                episode_size=500)
        rounds = []
        inner = s.kernels.sample_walks_blocks

        def spy(*a, **k):
            t0 = time.perf_counter()
            used = inner(*a, **k)
            rounds.append((list(s.kernels.walk_rounds), round(time.perf_counter() - t0, 3)))
            return used
        s.kernels.sample_walks_blocks = spy
        s.train(model=model, num_epoch=epochs, negative_weight=5, augmentation_step=5, random_walk_length=40,
                random_walk_batch_size=100, log_frequency=1 << 30, **kw)
        t = s.timing
        print(model, "epochs", epochs, "batches", t["batches"], "%.0f M edge-samples/s" % (t["batches"] * 1e5 / t["episodes"] / 1e6),
              "episodes %.2f s" % t["episodes"], "sampling calls:", rounds[:4], flush=True)
        s.clear()
