"""Experiment (GPU): link-prediction AUC of the product on the hub-heavy parity shapes, per pair order and seed.

    python scripts/experiments/auc_shapes.py blog 2000 sampled,grouped 17,18,19
"""
import logging
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import graphvite_amd as gv  # noqa: E402
from graphvite_amd import synthetic  # noqa: E402
from oracle_lib import link_prediction_auc  # noqa: E402
from reference_concurrency import SHAPES  # noqa: E402


def main():
    shape, epochs = sys.argv[1], int(sys.argv[2])
    orders = sys.argv[3].split(",")
    seeds = [int(x) for x in sys.argv[4].split(",")]
    extra = dict(kv.split("=") for kv in sys.argv[5:])
    kw, batch, episode, train_kw = SHAPES[shape]
    edges = synthetic.hub_community_edges(**kw)
    train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
    gv.init_logging(logging.ERROR)
    g = gv.graph.Graph()
    g.load(train)
    H, T, Y = test
    n2i = g.name2id
    keep = [(n2i[str(h)], n2i[str(t)], y) for h, t, y in zip(H, T, Y) if str(h) in n2i and str(t) in n2i]
    from graphvite_amd.kernels import HipKernels
    tune = HipKernels()
    tune.set_variant(int(extra.get("variant", 0)))
    tune.set_run_cap(int(extra.get("run_cap", 0)))
    tune.set_generation(int(extra.get("generation", 0)))
    tune.set_segment_steps(int(extra.get("steps", 0)))
    if "split" in extra:
        tune.set_split_hits(int(extra["split"]))
    hub = extra.get("hub", "default")
    tag = " ".join("%s=%s" % kv for kv in sorted(extra.items()))
    for order in orders:
        aucs = []
        for seed in seeds:
            t0 = time.time()
            s = gv.solver.GraphSolver(128, num_sampler_per_worker=int(extra.get("samplers", 8)), seed=seed, pair_order=gv.auto if order == "auto" else order,
                                      device_sampling=extra.get("device_sampling", "0") == "1",
                                      hub_rows=None if hub == "default" else (hub if hub == "auto" else int(hub)),
                                      fidelity=extra.get("fidelity", "auto"))
            s.hub_parts, s.hub_chain_cap = int(extra.get("parts", 0)), int(extra.get("cap", 0))
            s.hub_lerp = None if "lerp" not in extra else bool(int(extra["lerp"]))
            s.build(g, batch_size=batch, episode_size=int(extra.get("episode", episode)), num_partition=int(extra.get("partitions", 0)))
            s.train(model=extra.get("model", "LINE"), num_epoch=epochs,
                    augmentation_step=int(extra.get("aug", train_kw["augmentation_step"])),
                    random_walk_length=train_kw.get("walk_length", 40), random_walk_batch_size=train_kw.get("walk_batch", 100),
                    p=float(extra.get("p", 1)), q=float(extra.get("q", 1)), log_frequency=1 << 30)
            aucs.append(link_prediction_auc(s.vertex_embeddings, s.context_embeddings, [k[0] for k in keep],
                                            [k[1] for k in keep], [k[2] for k in keep]))
            print("%s epochs %d order %s seed %d: %d batches AUC %.6f (%.1f s)" % (shape, epochs, order, seed, s.batch_id,
                                                                                 aucs[-1], time.time() - t0), flush=True)
        print("%s epochs %d order %s [%s] %s, %d hub rows: mean %.6f sd %.6f" % (shape, epochs, order, tag, tune.describe_train(
            128, "SGD", 1, False, batch, s.partition_rows), s.hub_rows, np.mean(aucs), np.std(aucs)), flush=True)


if __name__ == "__main__":
    main()
