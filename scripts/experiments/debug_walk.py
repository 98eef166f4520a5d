"""Debug: DeepWalk-mode pools through the HIP kernel vs the oracle, batch by batch."""
import logging
import sys

import numpy as np
import torch

sys.path.insert(0, "tests"); sys.path.insert(0, ".")  # run from the repo root
import graphvite_amd as gv
from fake_kernels import OracleKernels
from graphvite_amd import synthetic

gv.init_logging(logging.ERROR)
edges = synthetic.community_edges(20000, 400000, num_community=100, seed=3)
train, _ = synthetic.link_prediction_split(edges, (100, 3, 3))


def make(kernels):
    g = gv.graph.Graph()
    g.load(train)
    s = gv.solver.GraphSolver(128, kernels=kernels, num_sampler_per_worker=4, seed=17)
    s.build(g, batch_size=500, episode_size=200)
    s._configure_training("DeepWalk", 50, False, 2, 10, 20, gv.auto, 1, 1, 1, 0.75, 5.0, 1 << 30)
    return g, s


g1, hip = make(None)
g2, ora = make(OracleKernels())
for name, s in (("hip", hip), ("ora", ora)):
    s._state = s._upload_state()
    s._pools = s._host_pools()
    s._fill(s._pools[0])
p_h = hip._pools[0][(0, 0)].numpy().copy()
p_o = ora._pools[0][(0, 0)].numpy().copy()
print("pools equal:", (p_h == p_o).all(), "max id", p_h.max(), "N", hip.num_vertex)
print("first pairs", p_h[:16].reshape(-1, 2).tolist())
for episode in range(3):
    for name, s in (("hip", hip), ("ora", ora)):
        s._train_episode(s._state, s._pools[0])
        if name == "hip":
            torch.cuda.synchronize()
        v = s._state["vertex"].cpu().numpy()
        c = s._state["context"].cpu().numpy()
        print(episode, name, "batch_id", s.batch_id, "loss", float(s._state["loss"].mean()), "|v|", np.abs(v).mean(),
              "|c|", np.abs(c).mean(), "dev pool == host pool",
              bool((s._state["pool_dev"][0].cpu().numpy() == s._pools[0][(0, 0)].numpy()).all()))
    dv = np.linalg.norm(hip._state["vertex"].cpu().numpy() - ora._state["vertex"].numpy()) / np.linalg.norm(
        ora._state["vertex"].numpy())
    dc = np.linalg.norm(hip._state["context"].cpu().numpy() - ora._state["context"].numpy()) / (np.linalg.norm(
        ora._state["context"].numpy()) + 1e-30)
    print(episode, "rel diff vertex %.4f context %.4f" % (dv, dc))
