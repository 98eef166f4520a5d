"""Experiment (CPU, this container only — needs oracle/_ref): the reference's own training loop with its kernel
emulated sequentially vs chunk-synchronously (oracle/ref_solver_harness.cpp), link-prediction AUC and run time.

    python scripts/experiments/reference_concurrency.py blog 200 0,5120,-5120 17,18      (-C: all reads at chunk start)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from graphvite_amd import synthetic  # noqa: E402
from oracle_lib import Oracle, ReferenceSolver, link_prediction_auc, reference_train  # noqa: E402

SHAPES = {
    # name: (generator kwargs, batch, episode, train kwargs)
    "blog": (dict(num_vertex=10312, num_edge=333983, gamma=2.8, num_community=39, p_in=0.7, seed=1024), 100000, 500,
             dict(augmentation_step=2, walk_length=40, walk_batch=100, shuffle_base=2)),
    "hub100k": (dict(num_vertex=100000, num_edge=2000000, gamma=2.3, num_community=100, p_in=0.7, seed=1024), 100000, 35,
                dict(augmentation_step=1)),
    # Youtube-like (BASELINE configs[2] / [3]: 1.1M nodes / 4.9M edges, maximum degree 28 754): a fifth of the nodes and edges, the
    # largest hub 7 % of the nodes, sum of squared degrees 8.9e8 — node2vec's per-edge tables still fit (2^30 entries)
    "tube": (dict(num_vertex=200000, num_edge=1000000, gamma=2.3, num_community=200, p_in=0.7, seed=1024), 100000, 200,
             dict(augmentation_step=5, walk_length=40, walk_batch=100, shuffle_base=1)),
}


def main():
    shape, epochs = sys.argv[1], int(sys.argv[2])
    chunks = [int(x) for x in sys.argv[3].split(",")]
    seeds = [int(x) for x in sys.argv[4].split(",")]
    kw, batch, episode, train_kw = SHAPES[shape]
    edges = synthetic.hub_community_edges(**kw)
    train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
    oracle = Oracle()
    for chunk in chunks:
        for seed in seeds:
            t0 = time.time()
            rs = ReferenceSolver(oracle, seed, train.astype(np.uint32), None, True, 1, 4, 1, 1, batch, episode)
            vertex, context, batch_id = reference_train(rs, "LINE", epochs, kernel_chunk=abs(chunk), reads_at_start=chunk < 0, threads=int(os.environ.get("THREADS", "8")), **train_kw)
            labels = rs.partition()[0]
            name2id = {int(label): i for i, label in enumerate(labels)}
            keep = [(name2id[int(h)], name2id[int(t)], y) for h, t, y in zip(*test) if int(h) in name2id and int(t) in name2id]
            auc = link_prediction_auc(vertex, context, [k[0] for k in keep], [k[1] for k in keep], [k[2] for k in keep])
            print("%s epochs %d chunk %d seed %d: %d batches, AUC %.6f, %.1f s" % (shape, epochs, chunk, seed, batch_id,
                                                                                   auc, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
