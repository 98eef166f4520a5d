"""Where does regrouping start to pay end to end?  Small graphs, alternating pair orders in one process."""
import logging, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import graphvite_amd as gv
from graphvite_amd import synthetic
gv.init_logging(logging.ERROR)
for V in (10000, 20000, 40000, 100000):
    g = gv.graph.Graph()
    g.load(synthetic.power_law_edges(V, 20 * V, seed=1))
    rates = {"sampled": [], "grouped": []}
    for rep in range(3):
        for order in ("sampled", "grouped"):
            s = gv.solver.GraphSolver(128, seed=1, pair_order=order)
            s.build(g, batch_size=100000, episode_size=100)
            s.train(model="LINE", num_epoch=int(4e9 / (20 * V)) // 10, augmentation_step=1, log_frequency=1 << 30)
            tm = s.timing
            rates[order].append(tm["batches"] * 100000 / tm["episodes"] / 1e6)
    print("%7d vertices (%5.1f MB tables): sampled %s  grouped %s" % (
        V, V * 512 / 1e6, " ".join("%.0f" % x for x in rates["sampled"]), " ".join("%.0f" % x for x in rates["grouped"])), flush=True)
