"""Experiment (GPU): link-prediction AUC on the headline shape (configs[1]: power-law 1M / 10M, 50 epochs) per kernel / pair
order, next to the reference's loop (tests/golden/reference_c2.npz: 0.668).

    python scripts/experiments/c2_auc.py [key=value ...]     variant= run_cap= split= order=sampled|grouped|auto epochs= seeds=
"""
import logging
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import graphvite_amd as gv  # noqa: E402
from graphvite_amd import synthetic  # noqa: E402
from oracle_lib import link_prediction_auc  # noqa: E402

extra = dict(kv.split("=", 1) for kv in sys.argv[1:])
gv.init_logging(logging.ERROR)
edges = synthetic.power_law_edges(1000000, 10000000, seed=1024)
train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
g = gv.graph.Graph()
g.load(train)
H, T, Y = (np.asarray(x) for x in test)
name2id = np.full(1000000, -1, np.int64)
names = np.array([int(x) for x in g.id2name], np.int64)
name2id[names] = np.arange(len(names))
keep = (name2id[H] >= 0) & (name2id[T] >= 0)
from graphvite_amd.kernels import HipKernels  # noqa: E402
tune = HipKernels()
configs = extra.get("configs", "variant=0").split(";")
for config in configs:
    kw = dict(kv.split("=") for kv in config.split(",") if kv)
    tune.set_variant(int(kw.get("variant", 0)))
    tune.set_run_cap(int(kw.get("run_cap", 0)))
    tune.set_split_hits(int(kw.get("split", 2)))
    tune.set_tuning(8, int(kw.get("chain_cap", 0)))  # GVK_TUNE_CHAIN_CAP
    tune.set_tuning(10, int(kw.get("whole_pairs", 0)))  # GVK_TUNE_HOT_WHOLE_PAIRS
    order = kw.get("order", "auto")
    aucs = []
    for seed in [int(x) for x in extra.get("seeds", "1024").split(",")]:
        t0 = time.time()
        s = gv.solver.GraphSolver(128, num_sampler_per_worker=15, seed=seed, pair_order=gv.auto if order == "auto" else order,
                                  device_sampling=kw.get("device", "0") == "1",
                                  hub_rows=kw.get("hub", "0") if kw.get("hub", "0") == "auto" else int(kw.get("hub", "0")))
        s.hub_parts = int(kw.get("parts", 0))
        s.build(g, batch_size=int(kw.get("batch", 100000)), num_partition=int(kw.get("partitions", 0)))
        s.train(model="LINE", num_epoch=int(extra.get("epochs", 50)), augmentation_step=1, log_frequency=1 << 30)
        aucs.append(link_prediction_auc(s.vertex_embeddings, s.context_embeddings, name2id[H[keep]], name2id[T[keep]], Y[keep]))
        rate = s.timing["batches"] * s.batch_size / s.timing["episodes"] / 1e6
    print("C2 [%s] %s %s, %d hub rows: AUC %s mean %.6f | %.0f M edge-samples/s" % (config, s.pair_order, tune.describe_train(
        128, "SGD", 1, False, s.batch_size, s.partition_rows), s.hub_rows, " ".join("%.6f" % a for a in aucs), np.mean(aucs), rate), flush=True)
