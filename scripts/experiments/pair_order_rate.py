"""End-to-end rate of pair_order "sampled" vs "grouped", alternating the two in one process (box load varies)."""
import logging, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import graphvite_amd as gv
from graphvite_amd import synthetic
gv.init_logging(logging.ERROR)
g = gv.graph.Graph()
g.load(synthetic.power_law_edges(1000000, 10000000, seed=0))
for P in (1, 16):
    for rep in range(3):
        for order in ("sampled", "grouped"):
            s = gv.solver.GraphSolver(128, seed=1, pair_order=order)
            s.build(g, batch_size=100000, num_partition=P, episode_size=250 if P == 1 else gv.auto)
            s.train(model="LINE", num_epoch=150, augmentation_step=1, log_frequency=1 << 30)
            tm = s.timing
            print("P=%-2d %-8s episode %4d  %5.0f M edge-samples/s   %s" % (
                P, order, s.episode_size, tm["batches"] * 100000 / tm["episodes"] / 1e6,
                ", ".join("%s %.2f" % kv for kv in (tm.get("loop") or {}).items())), flush=True)
