"""Generates tests/golden/reference_walks.npz: link-prediction AUC of the REFERENCE's own training loop
(GraphSolver::train as written, compiled for the host: oracle/ref_solver_harness.cpp, sequential kernel model) for the
random-walk models — DeepWalk and node2vec through its own sample_random_walk / sample_biased_random_walk with the
per-edge alias tables of build_edge_edge (include/instance/graph.cuh:298-450,656-721) — on the "blog" shape
(BASELINE configs[0]: 10 312 nodes / 333 983 edges, hub-heavy with communities; small enough for node2vec's per-edge
tables: 135M entries) with the walk hyper-parameters the reference ships for these models
(config/graph/deepwalk_youtube.yaml, node2vec_youtube.yaml: augmentation_step 5, random_walk_length 40,
random_walk_batch_size 100, batch 100 000, episode 500) and the quick start's 2000 epochs; and — SHAPE=tube EPOCHS=300 — on
a Youtube-like graph (scripts/experiments/reference_concurrency.py SHAPES: 200k nodes / 1M edges, the largest hub 7 % of the
nodes; 3 000 batches in episodes of 200), the scale BASELINE configs[2] / [3] run at.

    python tests/golden/make_walk_golden.py deepwalk node2vec_p0.25_q0.25 node2vec_p4_q2     # any subset; resumable
    SHAPE=tube EPOCHS=300 SEEDS=4 python tests/golden/make_walk_golden.py deepwalk node2vec_p0.25_q0.25
    SHAPE=tube EPOCHS=300 SEEDS=2 PARTITIONS=4 EPISODE=12 python tests/golden/make_walk_golden.py deepwalk node2vec_p0.25_q0.25
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

PATH = os.path.join(HERE, "reference_walks.npz")
SHAPE = os.environ.get("SHAPE", "blog")
PARTITIONS = int(os.environ.get("PARTITIONS", "1"))  # > 1: keys <shape>_p<P>_<model>, episode EPISODE batches per block
EPOCHS = int(os.environ.get("EPOCHS", "2000"))
WALK = dict(augmentation_step=5, walk_length=40, walk_batch=100, shuffle_base=1)
MODELS = {
    "deepwalk": ("DeepWalk", 1.0, 1.0),
    "node2vec_p0.25_q0.25": ("node2vec", 0.25, 0.25),   # BASELINE configs[3]
    "node2vec_p4_q2": ("node2vec", 4.0, 2.0),           # config/graph/node2vec_youtube.yaml:30-31
}
SEEDS = (17, 18, 19, 20, 21, 22)[:int(os.environ.get("SEEDS", "6"))]  # 3 at first; 6 since the comparison is between means of different random streams


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
    names = sys.argv[1:] or list(MODELS)
    oracle = Oracle()
    kw, batch, episode, _ = SHAPES[SHAPE]
    edges = synthetic.hub_community_edges(**kw)
    train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
    for name in names:
        model, p, q = MODELS[name]
        key = "%s_%s" % (SHAPE if PARTITIONS == 1 else "%s_p%d" % (SHAPE, PARTITIONS), name)
        for i, seed in enumerate(SEEDS):
            done = dict(np.load(PATH)) if os.path.exists(PATH) else {}
            if key in done and i < len(done[key]) and not np.isnan(done[key][i]):
                continue
            t0 = time.time()
            if PARTITIONS > 1:
                episode = int(os.environ.get("EPISODE", "12"))
            rs = ReferenceSolver(oracle, seed, train.astype(np.uint32), None, True, 1, 4, PARTITIONS, 1, batch, episode)
            vertex, context, batch_id = reference_train(rs, model, EPOCHS, p=p, q=q, **WALK)
            labels = rs.partition()[0]
            name2id = {int(label): j for j, label in enumerate(labels)}
            keep = [(name2id[int(h)], name2id[int(t)], y) for h, t, y in zip(*test) if int(h) in name2id and int(t) in name2id]
            auc = link_prediction_auc(vertex, context, [k[0] for k in keep], [k[1] for k in keep], [k[2] for k in keep])
            print("%s %s seed %d: %d batches, AUC %.6f, %.0f s" % (SHAPE, name, seed, batch_id, auc, time.time() - t0),
                  flush=True)
            update(key, i, auc, {
                SHAPE + "_args": np.array([kw["num_vertex"], kw["num_edge"], kw["num_community"], kw["seed"], batch, episode,
                                           EPOCHS, WALK["augmentation_step"], WALK["walk_length"], WALK["walk_batch"]], np.int64),
                SHAPE + "_gamma_p_in": np.array([kw["gamma"], kw["p_in"]], np.float64),
                key + "_p_q": np.array([p, q], np.float64),
                key + "_episode": np.int64(episode),
                ("seeds" if SHAPE == "blog" else (SHAPE if PARTITIONS == 1 else "%s_p%d" % (SHAPE, PARTITIONS)) + "_seeds"): np.array(SEEDS, np.int64)})


if __name__ == "__main__":
    main()
