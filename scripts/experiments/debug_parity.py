"""Experiment: LINE augmentation_step 2 (walk sampler + pseudo shuffle) — HIP vs sequential oracle AUC at several
conflict densities."""
import logging
import sys
import time

import numpy as np

sys.path.insert(0, "tests"); sys.path.insert(0, ".")  # run from the repo root
import graphvite_amd as gv
from fake_kernels import OracleKernels
from graphvite_amd import synthetic
from oracle_lib import link_prediction_auc

gv.init_logging(logging.ERROR)


def auc_of(g, s, split):
    H, T, Y = split
    n2i = g.name2id
    keep = [(n2i[str(h)], n2i[str(t)], y) for h, t, y in zip(H, T, Y) if str(h) in n2i and str(t) in n2i]
    return link_prediction_auc(s.vertex_embeddings, s.context_embeddings, [k[0] for k in keep], [k[1] for k in keep],
                               [k[2] for k in keep])


def run(train, kernels, batch, episode, epochs, aug):
    g = gv.graph.Graph()
    g.load(train)
    s = gv.solver.GraphSolver(128, kernels=kernels, num_sampler_per_worker=4, seed=17)
    s.build(g, batch_size=batch, episode_size=episode)
    s.train(model="LINE", num_epoch=epochs, augmentation_step=aug, random_walk_length=10, random_walk_batch_size=20,
            log_frequency=1 << 30)
    return g, s


for N, E, C, batch, episode, epochs in ((20000, 400000, 100, 100, 1000, 50), (20000, 400000, 100, 250, 400, 50),
                                        (60000, 1200000, 300, 250, 400, 17)):
    edges = synthetic.community_edges(N, E, num_community=C, seed=3)
    train, (valid, test) = synthetic.link_prediction_split(edges, (100, 3, 3))
    t = time.time()
    g1, hip = run(train, None, batch, episode, epochs, 2)
    t1 = time.time() - t
    g2, ora = run(train, OracleKernels(), batch, episode, epochs, 2)
    a, b = auc_of(g1, hip, test), auc_of(g2, ora, test)
    print("N %d batch %d: AUC hip %.6f oracle %.6f diff %.5f (hip %.1fs oracle %.1fs)" % (N, batch, a, b, abs(a - b), t1,
                                                                                       time.time() - t - t1), flush=True)
