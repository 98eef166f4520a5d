"""Generates tests/golden/reference_partitions.npz: link-prediction AUC of the REFERENCE's own training loop
(GraphSolver::train as written, compiled for the host: oracle/ref_solver_harness.cpp, sequential kernel model) when the
tables are split into P > 1 partitions and trained by W worker threads — the configurations every multi-GPU run lives
in.  The reference fills every (head, tail) block pool with the same number of samples whatever the block's share of
the edges (include/core/solver.h:1045-1052) and walks the blocks by get_schedule (solver.h:519-575); whatever that does
to learning on a hub-heavy graph whose communities correlate with the degree partition is the behaviour a drop-in has
to reproduce.

Shape "hub100k" (scripts/experiments/reference_concurrency.py SHAPES): 100k nodes / 2M edges, gamma 2.3, 100
communities, LINE, batch 100 000, 200 epochs — the P = 1 golden of reference_concurrency.npz with episode 35; here
episode = EPISODE[P] batches per block pool.

    python tests/golden/make_partition_golden.py 1,4 4,4            # "W,P" pairs; several processes may run side by side
Finished trainings are kept (file lock + atomic replace), so the script can be resumed.
"""
import fcntl
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.dirname(HERE), os.path.join(ROOT, "scripts", "experiments")]
from graphvite_amd import synthetic  # noqa: E402  (graph generator only; nothing of the product trains here)
from oracle_lib import Oracle, ReferenceSolver, link_prediction_auc, reference_train  # noqa: E402
from reference_concurrency import SHAPES  # noqa: E402

PATH = os.path.join(HERE, "reference_partitions.npz")
SHAPE = os.environ.get("SHAPE", "hub100k")
EPOCHS = int(os.environ.get("EPOCHS", "200"))
EPISODE = {1: 35, 2: 18, 4: 9, 8: 5, 16: 2}  # ~35 / P: an episode stays P * 35 batches
CONFIGS = ((1, 4), (1, 8), (1, 16), (4, 4), (8, 8))
SEEDS = (17, 18, 19, 20, 21, 22)  # the single-worker configurations: six seeds (the comparison is between means)


def update(key, index, value, extra):
    with open(PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        out = dict(np.load(PATH)) if os.path.exists(PATH) else {}
        values = out.get(key, np.full(len(SEEDS), np.nan))
        if len(values) < len(SEEDS):
            values = np.concatenate([values, np.full(len(SEEDS) - len(values), np.nan)])
        values[index] = value
        out[key] = values
        out.update(extra)
        np.savez_compressed(PATH + ".tmp.npz", **out)
        os.replace(PATH + ".tmp.npz", PATH)


def main():
    configs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or list(CONFIGS)
    oracle = Oracle()
    kw, batch, _, train_kw = SHAPES[SHAPE]
    edges = synthetic.hub_community_edges(**kw)
    train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
    for W, P in configs:
        key = "%s_w%d_p%d" % (SHAPE, W, P)
        for i, seed in enumerate(SEEDS):
            done = dict(np.load(PATH)) if os.path.exists(PATH) else {}
            if key in done and i < len(done[key]) and not np.isnan(done[key][i]):
                continue
            t0 = time.time()
            samplers = 4 if W == 1 else 1
            rs = ReferenceSolver(oracle, seed, train.astype(np.uint32), None, True, W, samplers, P, 1, batch, EPISODE[P])
            vertex, context, batch_id = reference_train(rs, "LINE", EPOCHS, **train_kw)
            labels = rs.partition()[0]
            name2id = {int(label): j for j, label in enumerate(labels)}
            keep = [(name2id[int(h)], name2id[int(t)], y) for h, t, y in zip(*test) if int(h) in name2id and int(t) in name2id]
            auc = link_prediction_auc(vertex, context, [k[0] for k in keep], [k[1] for k in keep], [k[2] for k in keep])
            print("%s W %d P %d episode %d seed %d: %d batches, AUC %.6f, %.0f s" % (SHAPE, W, P, EPISODE[P], seed, batch_id,
                                                                                   auc, time.time() - t0), flush=True)
            update(key, i, auc, {
                SHAPE + "_args": np.array([kw["num_vertex"], kw["num_edge"], kw["num_community"], kw["seed"], batch, EPOCHS,
                                           train_kw["augmentation_step"]], np.int64),
                SHAPE + "_gamma_p_in": np.array([kw["gamma"], kw["p_in"]], np.float64),
                key + "_episode": np.int64(EPISODE[P]),
                "seeds": np.array(SEEDS, np.int64)})


if __name__ == "__main__":
    main()
