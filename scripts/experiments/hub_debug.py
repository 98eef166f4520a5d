"""Bring-up (GPU): norms of the hub rows after 100 batches of 100 000 samples on a 200k-node power-law graph (the top hub
heads 1 400 samples of a batch) with the hub chains in several forms, next to the sequential host build's:
sequential, 100 batches: top 8: v 9.4217 c 0.1885 | top 64: v 26.0269 c 1.7924 | top 512: v 53.3115 c 8.7469 | top 4096: v 71.1321 c 33.3239 | rest v 39.6985 c 105.3164"""
import logging
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import graphvite_amd as gv  # noqa: E402
from graphvite_amd import synthetic  # noqa: E402
from graphvite_amd.kernels import HipKernels  # noqa: E402

gv.init_logging(logging.ERROR)
N, E, B, epochs = 200000, 2000000, 100000, 5
edges = synthetic.power_law_edges(N, E, seed=5)
g = gv.graph.Graph()
g.load(edges)
deg = np.bincount(edges.reshape(-1), minlength=N)
names = np.array([int(x) for x in g.id2name])
order = np.argsort(-deg[names], kind="stable")
tune = HipKernels()
for label, hub, serialized, cap in (("plain", 0, 0, 0), ("chains auto, cap 256", "auto", 0, 0), ("chains auto, cap 64", "auto", 0, 64),
                                    ("chains auto, cap 1024", "auto", 0, 1024), ("chains auto, cap 8192 (no parts)", "auto", 0, 8192),
                                    ("chains auto, no parts, three launches", "auto", 1, 8192), ("chains 512, no parts", 512, 0, 8192),
                                    ("chains 16384, no parts", 16384, 0, 8192)):
    tune.set_tuning(9, serialized)
    tune.set_tuning(8, cap)
    s = gv.solver.GraphSolver(128, num_sampler_per_worker=8, seed=3, hub_rows=hub, pair_order="sampled")
    s.build(g, batch_size=B, episode_size=20)
    s.train(model="LINE", num_epoch=epochs, augmentation_step=1, log_frequency=1 << 30)
    v, c = s.vertex_embeddings, s.context_embeddings
    print("%-40s %5d hub rows, %d batches: " % (label, s.hub_rows, s.batch_id) + " | ".join(
        "top %d: v %.4f c %.4f" % (K, np.linalg.norm(v[order[:K]]), np.linalg.norm(c[order[:K]])) for K in (8, 64, 512, 4096)) +
        " | rest v %.4f c %.4f" % (np.linalg.norm(v[order[4096:]]), np.linalg.norm(c[order[4096:]])), flush=True)
tune.set_tuning(9, 0)
tune.set_tuning(8, 0)
