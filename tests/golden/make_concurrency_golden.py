"""Generates tests/golden/reference_concurrency.npz: link-prediction AUC of the REFERENCE's own training loop
(GraphSolver::train as written — sampler threads, schedule, negative sampler, lr schedule — compiled for the host,
oracle/ref_solver_harness.cpp) on the two hub-heavy parity shapes, under three execution models of its kernel launch:

    sequential       the samples of a batch one after the other (what a CPU solver does; no update is lost)
    lock_step        chunk-synchronous, 5120 resident warps (a V100), lock step over the kernel's phases, last writer wins
    reads_at_start   chunk-synchronous, every row of a chunk read before any is written (the harsher bracket)

Shapes (scripts/experiments/reference_concurrency.py SHAPES): "blog" = BASELINE configs[0]'s shape — 10 312 nodes /
333 983 edges, hub-heavy with communities, config/demo/quick_start.yaml hyper-parameters (LINE, 2000 epochs,
augmentation_step 2, batch 100 000, episode 500); "hub100k" = 100k nodes / 2M edges, batch 100 000, 200 epochs.

Run here (the container that has /root/reference; 18 trainings of 6-9 minutes on 4 threads each):
    python tests/golden/make_concurrency_golden.py [shape ...]
Finished trainings are kept in the .npz, so the script can be run per shape / resumed.  Results do not depend on THREADS.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.dirname(HERE), os.path.join(ROOT, "scripts", "experiments")]
from graphvite_amd import synthetic  # noqa: E402  (graph generator only; nothing of the product trains here)
from oracle_lib import Oracle, ReferenceSolver, link_prediction_auc, reference_train  # noqa: E402
from reference_concurrency import SHAPES  # noqa: E402

PATH = os.path.join(HERE, "reference_concurrency.npz")
EPOCHS = {"blog": 2000, "hub100k": 200}
MODELS = {"sequential": (0, False), "lock_step": (5120, False), "reads_at_start": (5120, True)}
SEEDS = (17, 18, 19)


def main():
    shapes = sys.argv[1:] or list(EPOCHS)
    oracle = Oracle()
    threads = int(os.environ.get("THREADS", "4"))
    for shape in shapes:
        kw, batch, episode, train_kw = SHAPES[shape]
        edges = synthetic.hub_community_edges(**kw)
        train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
        for model, (chunk, reads_at_start) in MODELS.items():
            for i, seed in enumerate(SEEDS):
                out = dict(np.load(PATH)) if os.path.exists(PATH) else {}
                key = "%s_%s" % (shape, model)
                values = out.get(key, np.full(len(SEEDS), np.nan))
                if not np.isnan(values[i]):
                    continue
                rs = ReferenceSolver(oracle, seed, train.astype(np.uint32), None, True, 1, 4, 1, 1, batch, episode)
                vertex, context, batch_id = reference_train(rs, "LINE", EPOCHS[shape], kernel_chunk=chunk, threads=threads,
                                                            reads_at_start=reads_at_start, **train_kw)
                labels = rs.partition()[0]
                name2id = {int(label): j for j, label in enumerate(labels)}
                keep = [(name2id[int(h)], name2id[int(t)], y) for h, t, y in zip(*test) if int(h) in name2id and int(t) in name2id]
                values[i] = link_prediction_auc(vertex, context, [k[0] for k in keep], [k[1] for k in keep], [k[2] for k in keep])
                print("%s %s seed %d: %d batches, AUC %.6f" % (shape, model, seed, batch_id, values[i]), flush=True)
                out = dict(np.load(PATH)) if os.path.exists(PATH) else {}  # another shape may be running beside this one
                out[key] = values
                out[shape + "_args"] = np.array([kw["num_vertex"], kw["num_edge"], kw["num_community"], kw["seed"], batch, episode,
                                                 EPOCHS[shape], train_kw["augmentation_step"]], np.int64)
                out[shape + "_gamma_p_in"] = np.array([kw["gamma"], kw["p_in"]], np.float64)
                out["seeds"] = np.array(SEEDS, np.int64)
                np.savez_compressed(PATH + ".tmp.npz", **out)
                os.replace(PATH + ".tmp.npz", PATH)


if __name__ == "__main__":
    main()
