"""Experiment: a small stand-in for the headline shape — power-law 200k nodes / 2M edges, batch 100 000 (the top hub heads
1 400 samples of a batch, as hub-heavy per launch as configs[1]), LINE, 50 epochs — link-prediction AUC per executor.
With GVK_LIBRARY = the host build (tests/hostdev) the kernels are the sequential oracle: the value to match.  The host build
is also an executor SIMULATOR: with hub rows (hub=…, parts=…, cap=…, lerp=…) and GVH_EXECUTOR=units it trains every part the
way the device path does (oracle/gv_oracle.c gvo_train_hot: chains of both families from the part's start state, long chains
as tasks composed in order, then the pairs — hub rows as the chains left them or along their way), with
GVH_EXECUTOR=pipelined the chains of part u + 1 are computed before the pairs of part u write — what a change of the device
path would do to learning, measured without a GPU (the Hogwild losses of the other rows are not simulated).

    GVK_ALLOW_TEST_LIBRARY=1 GVK_LIBRARY=tests/hostdev/build/libgvk_host.so GVH_EXECUTOR=units \
        python scripts/experiments/c2mini.py seeds=3,4 configs="hub=0;hub=auto,parts=8;hub=auto,parts=5,lerp=1" [epochs=50]
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

extra = dict(kv.split("=", 1) for kv in sys.argv[1:])
gv.init_logging(logging.ERROR)
N, E, B = int(extra.get("nodes", 200000)), int(extra.get("edges", 2000000)), int(extra.get("batch", 100000))
edges = synthetic.power_law_edges(N, E, gamma=float(extra.get("gamma", 2.3)), seed=int(extra.get("graph_seed", 5)))  # gamma=2.0: a heavier head (the held-out graph of tests/golden/make_configs_golden.py)  # nodes=1000000 edges=10000000 graph_seed=1024: the headline shape itself
train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
g = gv.graph.Graph()
g.load(train)
H, T, Y = (np.asarray(x) for x in test)
name2id = np.full(N, -1, np.int64)
names = np.array([int(x) for x in g.id2name], np.int64)
name2id[names] = np.arange(len(names))
keep = (name2id[H] >= 0) & (name2id[T] >= 0)
host = "host" in os.environ.get("GVK_LIBRARY", "")
if not host:
    from graphvite_amd.kernels import HipKernels
    tune = HipKernels()
for config in extra.get("configs", "hub=0").split(";"):
    kw = dict(kv.split("=") for kv in config.split(",") if kv)
    if not host:
        tune.set_variant(int(kw.get("variant", 0)))
    aucs = []
    for seed in [int(x) for x in extra.get("seeds", "3").split(",")]:
        t0 = time.time()
        hub = kw.get("hub", "0")
        s = gv.solver.GraphSolver(128, num_sampler_per_worker=6, seed=seed, hub_rows=None if hub == "default" else (hub if hub == "auto" else int(hub)),
                                  pair_order=kw.get("order", "sampled") if kw.get("order", "sampled") != "auto" else gv.auto)
        s.hub_parts, s.hub_chain_cap = int(kw.get("parts", 0)), int(kw.get("cap", 0))
        s.hub_lerp = None if "lerp" not in kw else bool(int(kw["lerp"]))
        optimizer = {"sgd": lambda: gv.auto, "adam": lambda: gv.optimizer.Adam(1e-3, 0.005), "momentum": lambda: gv.optimizer.Momentum(0.025, 0.005, 0.9)}[kw.get("opt", "sgd")]()
        s.build(g, optimizer=optimizer, batch_size=B, episode_size=int(kw.get("episode", 20)), num_partition=int(kw.get("partitions", 0)))
        s.train(model="LINE", num_epoch=int(extra.get("epochs", 50)), augmentation_step=1, log_frequency=1 << 30)
        aucs.append(link_prediction_auc(s.vertex_embeddings, s.context_embeddings, name2id[H[keep]], name2id[T[keep]], Y[keep]))
        print("    seed %d: AUC %.6f (%d batches, %.0f s)" % (seed, aucs[-1], s.batch_id, time.time() - t0), flush=True)
    print("c2mini [%s]%s, %d hub rows: AUC %s mean %.6f" % (config, " sequential host build" if host else "", s.hub_rows,
                                                           " ".join("%.6f" % a for a in aucs), np.mean(aucs)), flush=True)
