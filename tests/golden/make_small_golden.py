"""Adds `train_small_w1_p4_aucs` to tests/golden/reference_solver.npz: the reference's OWN training loop (GraphSolver::train as
written, compiled for the host: oracle/ref_solver_harness.cpp, sequential kernel model) on the small community graph of
`train_small_args` (4000 nodes / 80 000 edges, LINE, 150 epochs, batch 200, episode 40) with ONE worker and FOUR partitions,
seeds 3 .. 7 — what the multi-process gloo test of tests/test_solver_cpu.py is held to (+-0.002 between means).

    python tests/golden/make_small_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.dirname(HERE)]
from graphvite_amd import synthetic  # noqa: E402  (graph generator only; nothing of the product trains here)
from oracle_lib import Oracle, ReferenceSolver, link_prediction_auc, reference_train  # noqa: E402

PATH = os.path.join(HERE, "reference_solver.npz")


def main():
    G = dict(np.load(PATH))
    n, e, communities, graph_seed, batch, episode, epochs = [int(x) for x in G["train_small_args"]]
    small = synthetic.community_edges(n, e, num_community=communities, seed=graph_seed)
    train, (valid, test) = synthetic.link_prediction_split(small, (100, 3, 3))
    oracle = Oracle()
    for W, P in ((1, 4), (1, 1)):
        aucs = []
        for seed in (3, 4, 5, 6, 7):
            rs = ReferenceSolver(oracle, seed, train.astype(np.uint32), None, True, W, 2, P, 1, batch, episode)
            vertex, context, batch_id = reference_train(rs, "LINE", epochs, 1)
            labels = rs.partition()[0]
            name2id = {int(label): i for i, label in enumerate(labels)}
            keep = [(name2id[int(h)], name2id[int(t)], y) for h, t, y in zip(*test) if int(h) in name2id and int(t) in name2id]
            aucs.append(link_prediction_auc(vertex, context, [k[0] for k in keep], [k[1] for k in keep], [k[2] for k in keep]))
            print("reference training loop, %d worker / %d partitions, seed %d: AUC %.6f" % (W, P, seed, aucs[-1]), flush=True)
        G["train_small_w%d_p%d_aucs" % (W, P)] = np.array(aucs, np.float64)
    np.savez_compressed(PATH, **G)


if __name__ == "__main__":
    main()
