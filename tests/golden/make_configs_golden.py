"""Generates tests/golden/reference_configs.npz: link-prediction AUC of the REFERENCE's own training loop (GraphSolver::train as
written, compiled for the host: oracle/ref_solver_harness.cpp, sequential kernel model) at the SHAPES BASELINE configs[2..4]
state, for the optimizers the reference ships beside SGD, and on a hub-heavy graph none of the product's constants was tuned on:

  fs_line_p8      configs[4]'s shape: a Friendster-like power-law graph (2M nodes / 40M edges; the real one has 65M / 1.8B and
                  does not fit a host training loop), **dim 96** (oracle/_ref/libgvref_solver_96.so), LINE with augmentation_step 2
                  (random walks of 40, pools in walk order, shuffle_base 2), 8 partitions on one worker, episodes of 8 batches per
                  block, SGD 0.025 / 0.005 (config/graph/line_friendster.yaml:7-27), 50 epochs = 20 000 batches of 100 000 (13 epochs
                  leave the reference's own loop at AUC 0.506: nothing learnt yet, nothing to compare)
  yt_deepwalk     configs[2]'s shape AT ITS STATED SIZE: a Youtube-sized hub / community graph (1 138 499 nodes / 4 945 382 edge
  yt_p4_deepwalk  lines), DeepWalk, augmentation_step 5, walks of 40 (config/graph/deepwalk_youtube.yaml:7-27), 100 epochs = 4 900
                  batches in episodes of 500 — one partition, and the 4 partitions of configs[3]'s per-GPU shape (episodes of 30)
  c2_adam         the headline shape (configs[1]: power-law 1M / 10M, LINE, dim 128, one partition, 50 epochs) under
  c2_momentum     train_2_moment<kAdam> / train_1_moment<kMomentum> (instance/gpu/graph.cuh:104-242): Adam 1e-3, Momentum 0.005 (at SGD's 0.025 the
                  reference's own loop ends at AUC 0.246: momentum 0.999 overshoots and the ranking inverts — nothing to compare)
  held_p1         a held-out hub-heavy graph: power-law exponent 2.0 (the headline graph: 2.3 — node weights rank^-1 instead of rank^-0.77: a
                  heavier head) from another generator seed,
  held_p8_e8      1.5M nodes / 12M edges, LINE, dim 128, 42 epochs = 5 040 batches; one partition and 8 (episodes of 8)

Every job: SEEDS seeds.  ~10-25 minutes of host time per training.

    python tests/golden/make_configs_golden.py [job ...]         # resumable, lock-protected: several processes may run side by side
    SEED_INDEX=1 python tests/golden/make_configs_golden.py fs_line_p8     # one seed of a job per process
"""
import fcntl
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.dirname(HERE)]
from graphvite_amd import synthetic  # noqa: E402  (graph generators only; nothing of the product trains here)
from oracle_lib import Oracle, ReferenceSolver, link_prediction_auc, reference_train  # noqa: E402

PATH = os.path.join(HERE, "reference_configs.npz")
SEEDS = (17, 18, 19, 20, 21, 22, 23, 24)[:int(os.environ.get("SEEDS", "3"))]
BATCH = 100000
WALK = dict(walk_length=40, walk_batch=100)

# name: (graph, dim, model, train kwargs, partitions, episode (0 = automatic), epochs, optimizer)
JOBS = {
    "fs_line_p8": ("friendster_like", 96, "LINE", dict(augmentation_step=2, shuffle_base=2, **WALK), 8, 8, 50, None),
    # diagnostics for the open line of DESIGN.md section 7.11 (a): the same shape at dim 128, and in one partition
    "fs128_line_p8": ("friendster_like", 128, "LINE", dict(augmentation_step=2, shuffle_base=2, **WALK), 8, 8, 50, None),
    "fs_line_p1": ("friendster_like", 96, "LINE", dict(augmentation_step=2, shuffle_base=2, **WALK), 1, 0, 50, None),
    "yt_deepwalk": ("youtube_like", 128, "DeepWalk", dict(augmentation_step=5, shuffle_base=1, **WALK), 1, 500, 100, None),
    "yt_p4_deepwalk": ("youtube_like", 128, "DeepWalk", dict(augmentation_step=5, shuffle_base=1, **WALK), 4, 30, 100, None),
    "c2_adam": ("headline", 128, "LINE", dict(augmentation_step=1), 1, 0, 50, ("Adam", 1e-3, 0.005)),
    "c2_momentum": ("headline", 128, "LINE", dict(augmentation_step=1), 1, 0, 50, ("Momentum", 0.005, 0.005)),
    # Momentum with the coefficient 0.9 at SGD's learning rate: where the reference's loop learns (with the class's default 0.999 a row's
    # moment needs about a thousand of ITS OWN updates to warm up, more than most rows of this graph meet in 50 epochs: the loop ends below 0.5)
    "c2_momentum09": ("headline", 128, "LINE", dict(augmentation_step=1), 1, 0, 50, ("Momentum", 0.025, 0.005, 0.9)),
    # configs[3] as BASELINE words it: node2vec p = q = 0.25 at Youtube's size in 4 partitions.  The graph: the Youtube-sized generator
    # with exponent 2.5 (largest degree 19 115; the real graph's: 28 754) — the reference's per-edge tables (graph.cuh:656-677) hold
    # sum deg^2 = 2.3e9 entries = 18 GB here; on the exponent-2.3 graph of yt_deepwalk they would hold 8.3e9 = 66 GB, more than this host has
    "yt_p4_node2vec": ("youtube_n2v", 128, "node2vec", dict(augmentation_step=5, shuffle_base=1, p=0.25, q=0.25, **WALK), 4, 30, 100, None),
    # two graphs between the headline shape (its largest vertex takes 1.0 % of the degree: long chains in one round) and the held-out one (6.8 %:
    # rounds), either side of the 2 % at which configure() switches rounds on (gvx_engine.cpp kHubRoundShare): 1.8 % and 2.5 %
    "mid18_p1": ("mid18", 128, "LINE", dict(augmentation_step=1), 1, 0, 42, None),
    "mid25_p1": ("mid25", 128, "LINE", dict(augmentation_step=1), 1, 0, 42, None),
    "held_p1": ("held_out", 128, "LINE", dict(augmentation_step=1), 1, 0, 42, None),
    "held_p8_e8": ("held_out", 128, "LINE", dict(augmentation_step=1), 8, 8, 42, None),
}


def graph_edges(name):
    """The edge list of a shape — also what tests/test_solver_gpu.py trains (same function, imported from here)."""
    if name == "friendster_like":
        return synthetic.power_law_edges(2000000, 40000000, seed=65)
    if name == "youtube_like":
        return synthetic.hub_community_edges(num_vertex=1138499, num_edge=4945382, gamma=2.3, num_community=400, p_in=0.7, seed=1024)
    if name == "youtube_n2v":
        return synthetic.hub_community_edges(num_vertex=1138499, num_edge=4945382, gamma=2.5, num_community=400, p_in=0.7, seed=1024)
    if name == "headline":
        return synthetic.power_law_edges(1000000, 10000000, seed=1024)
    if name == "mid18":
        return synthetic.power_law_edges(1200000, 11000000, gamma=2.2, seed=777)
    if name == "mid25":
        return synthetic.power_law_edges(1200000, 11000000, gamma=2.15, seed=778)
    if name == "held_out":
        return synthetic.power_law_edges(1500000, 12000000, gamma=2.0, seed=4711)
    raise KeyError(name)


def update(key, index, value, extra):
    with open(PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        out = dict(np.load(PATH)) if os.path.exists(PATH) else {}
        values = out.get(key, np.full(len(SEEDS), np.nan))
        if len(values) <= index:
            values = np.concatenate([values, np.full(index + 1 - len(values), np.nan)])
        values[index] = value
        out[key] = values
        out.update(extra)
        np.savez_compressed(PATH + ".tmp.npz", **out)
        os.replace(PATH + ".tmp.npz", PATH)


def main():
    names = sys.argv[1:] or list(JOBS)
    oracle = Oracle()
    graphs = {}
    for name in names:
        graph, dim, model, train_kw, partitions, episode, epochs, optimizer = JOBS[name]
        for i, seed in enumerate(SEEDS):
            if os.environ.get("SEED_INDEX") and i != int(os.environ["SEED_INDEX"]):
                continue  # one seed per process: several processes of one job side by side
            done = dict(np.load(PATH)) if os.path.exists(PATH) else {}
            if name in done and i < len(done[name]) and not np.isnan(done[name][i]):
                continue
            if graph not in graphs:
                graphs.clear()
                edges = graph_edges(graph)
                graphs[graph] = synthetic.link_prediction_split(edges, (100, 1, 1))
            train, (valid, test) = graphs[graph]
            t0 = time.time()
            rs = ReferenceSolver(oracle, seed, train.astype(np.uint32), None, True, 1, 4, partitions, 1, BATCH, episode, dim=dim,
                                 optimizer=optimizer)
            vertex, context, batch_id = reference_train(rs, model, epochs, **train_kw)
            labels = rs.partition()[0]
            name2id = np.full(int(max(labels.max(), np.asarray(test[0]).max(), np.asarray(test[1]).max())) + 1, -1, np.int64)
            name2id[labels] = np.arange(len(labels))
            H, T, Y = (np.asarray(x) for x in test)
            ok = (name2id[H] >= 0) & (name2id[T] >= 0)
            auc = link_prediction_auc(vertex, context, name2id[H[ok]], name2id[T[ok]], Y[ok])
            print("%s seed %d: %d partitions, episode %d, %d batches, AUC %.6f, %.0f s" % (name, seed, rs.num_partition, rs.episode_size,
                                                                                          batch_id, auc, time.time() - t0), flush=True)
            update(name, i, auc, {name + "_args": np.array([dim, partitions, rs.episode_size, epochs, batch_id], np.int64),
                                  "seeds": np.array(SEEDS, np.int64)})
            del rs, vertex, context


if __name__ == "__main__":
    main()
