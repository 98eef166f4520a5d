"""Debug: full train() loop, DeepWalk mode, HIP vs oracle, loss trajectory + AUC."""
import logging
import sys

import numpy as np
import torch

sys.path.insert(0, "tests"); sys.path.insert(0, ".")  # run from the repo root
import graphvite_amd as gv
from fake_kernels import OracleKernels
from graphvite_amd import synthetic
from oracle_lib import link_prediction_auc

edges = synthetic.community_edges(20000, 400000, num_community=100, seed=3)
train, (valid, test) = synthetic.link_prediction_split(edges, (100, 3, 3))


def auc_of(g, s, split):
    H, T, Y = split
    n2i = g.name2id
    keep = [(n2i[str(h)], n2i[str(t)], y) for h, t, y in zip(H, T, Y) if str(h) in n2i and str(t) in n2i]
    return link_prediction_auc(s.vertex_embeddings, s.context_embeddings, [k[0] for k in keep], [k[1] for k in keep],
                               [k[2] for k in keep])


def run(kernels, model, aug, epochs, sync=False, threads=4):
    gv.init_logging(logging.ERROR)
    g = gv.graph.Graph()
    g.load(train)
    gv.init_logging(logging.INFO)
    s = gv.solver.GraphSolver(128, kernels=kernels, num_sampler_per_worker=threads, seed=17)
    s.build(g, batch_size=BATCH, episode_size=EPISODE)
    if sync:
        orig = s._train_episode
        def wrapped(state, pools):
            r = orig(state, pools)
            torch.cuda.synchronize()
            return r
        s._train_episode = wrapped
    logging.getLogger("graphvite_amd").handlers[0].setFormatter(logging.Formatter("  %(message)s"))
    import io as _io
    s.train(model=model, num_epoch=epochs, augmentation_step=aug, random_walk_length=10, random_walk_batch_size=20,
            log_frequency=1000)
    return g, s


from graphvite_amd.kernels import HipKernels
for BATCH, EPISODE, EPOCHS in ((100000, 10, 500), (20000, 20, 200)):
    for label, variant in (("hip-strided", 0), ("hip-poolorder", 2)):
        print("=====", label, BATCH)
        HipKernels().set_variant(variant)
        g, s = run(model="DeepWalk", aug=2, epochs=EPOCHS, kernels=None)
    print("=====", label, "AUC", auc_of(g, s, test), "|v|", np.abs(s.vertex_embeddings).mean(), "|c|",
          np.abs(s.context_embeddings).mean())
