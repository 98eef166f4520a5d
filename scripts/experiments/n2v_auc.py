"""Experiment (GPU): node2vec p = q = 0.25 on the Youtube-sized graph of tests/golden/make_configs_golden.py (yt_p4_node2vec), CPU samplers against
positives drawn on the device, in P partitions — where does the device sampler's +0.002 come from (P = 1: one pool, every walk owns its slots;
P > 1: pairs binned into block pools, full pools drop what arrives later)?

    python scripts/experiments/n2v_auc.py partitions=1,4 seeds=1024,5,6 [model=node2vec p=0.25 q=0.25 epochs=100 episode=30]
"""
import logging
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import graphvite_amd as gv  # noqa: E402
from graphvite_amd import synthetic  # noqa: E402
from make_configs_golden import graph_edges  # noqa: E402
from oracle_lib import link_prediction_auc  # noqa: E402

extra = dict(kv.split("=", 1) for kv in sys.argv[1:])
edges = graph_edges("youtube_n2v")
train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
gv.init_logging(logging.ERROR)
g = gv.graph.Graph()
g.load(train)
H, T, Y = (np.asarray(x) for x in test)
name2id = np.full(int(max(edges.max(), H.max(), T.max())) + 1, -1, np.int64)
name2id[np.array([int(x) for x in g.id2name], np.int64)] = np.arange(g.num_vertex)
keep = (name2id[H] >= 0) & (name2id[T] >= 0)
model, p, q = extra.get("model", "node2vec"), float(extra.get("p", 0.25)), float(extra.get("q", 0.25))
for partitions in [int(x) for x in extra.get("partitions", "1,4").split(",")]:
    for device in ((True,) if "sb" in extra else (False, True)):
        aucs = []
        for seed in [int(x) for x in extra.get("seeds", "1024,5,6").split(",")]:
            s = gv.solver.GraphSolver(128, num_sampler_per_worker=8, seed=seed, device_sampling=device)
            s.build(g, batch_size=100000, num_partition=partitions, episode_size=int(extra.get("episode", 30)) if partitions > 1 else 500)
            s.train(model=model, num_epoch=int(extra.get("epochs", 100)), augmentation_step=5, random_walk_length=40, random_walk_batch_size=100, shuffle_base=int(extra.get("sb", 1)),
                    p=p, q=q, log_frequency=1 << 30)
            aucs.append(link_prediction_auc(s.vertex_embeddings, s.context_embeddings, name2id[H[keep]], name2id[T[keep]], Y[keep]))
            parts = s.hub_parts_used
            s.clear()
        print("%s p=%g q=%g, %d partition(s), %s samplers (%d parts): AUC %s mean %.6f" % (model, p, q, partitions, "device" if device else "CPU", parts,
                                                                                           " ".join("%.6f" % a for a in aucs), np.mean(aucs)), flush=True)
