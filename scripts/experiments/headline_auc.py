"""Experiment (GPU): link-prediction AUC of the headline shape (configs[1]) under the solver options given by the environment
(GVX_HUB_EXECUTOR, GVX_HUB_PAIR_LAUNCHES, ...) next to the reference's own loop (tests/golden/reference_c2.npz).

    GVX_HUB_EXECUTOR=ahead python scripts/experiments/headline_auc.py seeds=1024,5,6,7 [partitions=8 episode=8]
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
G = np.load(os.path.join(ROOT, "tests", "golden", "reference_c2.npz"))
n, e, graph_seed, batch, episode, epochs = [int(x) for x in G["c2_args"]]
partitions, ep = int(extra.get("partitions", 1)), int(extra.get("episode", 0))
key = "c2_line_sequential" if partitions == 1 else "c2_line_p%d" % partitions + ("_e%d" % ep if ep else "")
reference = G[key][~np.isnan(G[key])]
edges = synthetic.power_law_edges(n, e, seed=graph_seed)
train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
gv.init_logging(logging.ERROR)
g = gv.graph.Graph()
g.load(train)
H, T, Y = (np.asarray(x) for x in test)
name2id = np.full(n, -1, np.int64)
name2id[np.array([int(x) for x in g.id2name], np.int64)] = np.arange(g.num_vertex)
keep = (name2id[H] >= 0) & (name2id[T] >= 0)
aucs = []
for seed in [int(x) for x in extra.get("seeds", "1024,5").split(",")]:
    s = gv.solver.GraphSolver(128, num_sampler_per_worker=8, seed=seed, device_sampling=extra.get("device") == "1")
    s.build(g, batch_size=batch, num_partition=partitions if partitions > 1 else gv.auto, episode_size=ep or gv.auto)
    t0 = time.time()
    s.train(model="LINE", num_epoch=epochs, augmentation_step=1, log_frequency=1 << 30)
    el = time.time() - t0
    aucs.append(link_prediction_auc(s.vertex_embeddings, s.context_embeddings, name2id[H[keep]], name2id[T[keep]], Y[keep]))
    rate = s.batch_id * batch / el / 1e6
    s.clear()
print("headline %s [executor %s, pair launches %s]: AUC %s mean %.6f | reference %.6f | difference %+.6f | last training %.1f M edge-samples/s end to end" % (
    key, os.environ.get("GVX_HUB_EXECUTOR", "default"), os.environ.get("GVX_HUB_PAIR_LAUNCHES", "-"), " ".join("%.6f" % a for a in aucs), np.mean(aucs),
    reference.mean(), np.mean(aucs) - reference.mean(), rate), flush=True)
