"""pair_order = "sampled" vs "grouped": link-prediction AUC (does grouping same-head pairs hurt what is learned?) and
end-to-end throughput.  Run on the GPU box: python scripts/experiments/pair_order.py"""
import logging
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import graphvite_amd as gv
from graphvite_amd import synthetic
from oracle_lib import link_prediction_auc

gv.init_logging(logging.ERROR)


def auc_of(g, s, split):
    H, T, Y = split
    n2i = g.name2id
    keep = [(n2i[str(h)], n2i[str(t)], y) for h, t, y in zip(H, T, Y) if str(h) in n2i and str(t) in n2i]
    return link_prediction_auc(s.vertex_embeddings, s.context_embeddings, [k[0] for k in keep], [k[1] for k in keep],
                               [k[2] for k in keep])


def one(name, edges, batch, episode, epochs, aug=1, seeds=(1, 2, 3)):
    train, (valid, test) = synthetic.link_prediction_split(edges, (100, 3, 3))
    g = gv.graph.Graph()
    g.load(train)
    for order in ("sampled", "grouped"):
        aucs, rates = [], []
        for seed in seeds:
            s = gv.solver.GraphSolver(128, seed=seed, pair_order=order)
            s.build(g, batch_size=batch, episode_size=episode)
            s.train(model="LINE", num_epoch=epochs, augmentation_step=aug, log_frequency=1 << 30)
            aucs.append(auc_of(g, s, test))
            rates.append(s.timing["batches"] * batch / s.timing["episodes"] / 1e6)
        print("%-34s %-8s AUC %s  mean %.4f   %.0f M edge-samples/s" % (
            name, order, " ".join("%.4f" % a for a in aucs), np.mean(aucs), np.mean(rates)), flush=True)


one("power-law 4k/80k batch 1000", synthetic.power_law_edges(4000, 80000, seed=3), 1000, 100, 100)
one("power-law 10k/334k batch 100k", synthetic.power_law_edges(10312, 333983, seed=1024), 100000, 100, 1000)
one("power-law 100k/2M batch 100k", synthetic.power_law_edges(100000, 2000000, seed=5), 100000, 50, 400)
one("power-law 1M/10M batch 100k", synthetic.power_law_edges(1000000, 10000000, seed=0), 100000, 250, 100, seeds=(1,))
one("community 20k/400k batch 500", synthetic.community_edges(20000, 400000, num_community=100, seed=3), 500, 200, 50)


def rate(name, edges, partitions, epochs):
    g = gv.graph.Graph()
    g.load(edges)
    for order in ("sampled", "grouped"):
        s = gv.solver.GraphSolver(128, seed=1, pair_order=order)
        s.build(g, batch_size=100000, num_partition=partitions)
        s.train(model="LINE", num_epoch=epochs, augmentation_step=1, log_frequency=1 << 30)
        print("%-34s %-8s P=%-2d episode %d  %.0f M edge-samples/s" % (
            name, order, partitions, s.episode_size, s.timing["batches"] * 100000 / s.timing["episodes"] / 1e6), flush=True)


big = synthetic.power_law_edges(1000000, 10000000, seed=0)
for P in (1, 4, 16):
    rate("power-law 1M/10M", big, P, 200)
