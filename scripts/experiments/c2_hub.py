"""Experiment (GPU): link-prediction AUC and training rate on the headline shape (configs[1]: power-law 1M / 10M, 50 epochs)
per executor configuration — hub rows by chains, parts per batch, chain cap, lerp, partitions — next to the reference's loop
(tests/golden/reference_c2.npz).

    python scripts/experiments/c2_hub.py configs="hub=auto,parts=8;hub=auto,parts=5,lerp=1" [epochs=50] [seeds=1024] [nodes= edges= graph_seed=]
"""
import logging
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import graphvite_amd as gv  # noqa: E402
from graphvite_amd import synthetic  # noqa: E402
from oracle_lib import link_prediction_auc  # noqa: E402

extra = dict(kv.split("=", 1) for kv in sys.argv[1:])
gv.init_logging(logging.ERROR)
N, E = int(extra.get("nodes", 1000000)), int(extra.get("edges", 10000000))
edges = synthetic.power_law_edges(N, E, seed=int(extra.get("graph_seed", 1024)))
train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
g = gv.graph.Graph()
g.load(train)
H, T, Y = (np.asarray(x) for x in test)
name2id = np.full(N, -1, np.int64)
names = np.array([int(x) for x in g.id2name], np.int64)
name2id[names] = np.arange(len(names))
keep = (name2id[H] >= 0) & (name2id[T] >= 0)
golden = os.path.join(ROOT, "tests", "golden", "reference_c2.npz")
reference = dict(np.load(golden)) if os.path.exists(golden) else {}
for config in extra.get("configs", "hub=auto").split(";"):
    kw = dict(kv.split("=") for kv in config.split(",") if kv)
    aucs = []
    if "GVK_LIBRARY" not in os.environ or "host" not in os.environ["GVK_LIBRARY"]:  # tune<key>=<value>: gvk_set_tuning (include/gvk.h), e.g. tune9=1
        from graphvite_amd.kernels import HipKernels
        for key, default in ((9, 0), (10, 1), (12, -1)):  # GVK_TUNE_HOT_SERIALIZED, GVK_TUNE_HOT_ORDER, GVK_TUNE_ROUND_STEPS
            HipKernels().set_tuning(key, int(kw.get("tune%d" % key, default)))
    for seed in [int(x) for x in extra.get("seeds", "1024").split(",")]:
        hub = kw.get("hub", "default")
        s = gv.solver.GraphSolver(128, num_sampler_per_worker=15, seed=seed, device_sampling=kw.get("device", "0") == "1",
                                  hub_rows=None if hub == "default" else (hub if hub == "auto" else int(hub)),
                                  fidelity=kw.get("fidelity", "auto"))
        s.hub_parts = int(kw.get("parts", 0))
        s.hub_lerp = None if "lerp" not in kw else bool(int(kw["lerp"]))
        s.hub_chain_cap = int(kw.get("cap", 0))
        s.build(g, batch_size=int(kw.get("batch", 100000)), num_partition=int(kw.get("partitions", 0)),
                episode_size=int(kw.get("episode", 0)) or gv.auto)
        s.train(model="LINE", num_epoch=int(extra.get("epochs", 50)), augmentation_step=1, log_frequency=1 << 30)
        aucs.append(link_prediction_auc(s.vertex_embeddings, s.context_embeddings, name2id[H[keep]], name2id[T[keep]], Y[keep]))
        rate = s.timing["batches"] * s.batch_size / s.timing["episodes"] / 1e6
    P = int(kw.get("partitions", 0)) or 1
    key = "c2_line_sequential" if P == 1 else "c2_line_p%d" % P + ("_e%s" % kw["episode"] if "episode" in kw else "")
    want = float(np.nanmean(reference[key])) if key in reference and N == 1000000 else float("nan")
    print("C2 [%s] %d hub rows, %d partitions: AUC %s mean %.6f (reference loop %.6f: %+.6f) | %.0f M edge-samples/s" % (
        config, s.hub_rows, s.num_partition, " ".join("%.6f" % a for a in aucs), np.mean(aucs), want, np.mean(aucs) - want, rate), flush=True)
