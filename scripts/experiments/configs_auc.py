"""Experiment (GPU): AUC of a job of tests/golden/make_configs_golden.py under overrides of the hub rule, next to the reference's loop.

    python scripts/experiments/configs_auc.py job=held_p1 seeds=1024,5 variants="default;parts=64;hub_rows=8000;rounds=0;device=1"
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_configs_gpu as T  # noqa: E402

extra = dict(kv.split("=", 1) for kv in sys.argv[1:])
job = extra["job"]
seeds = [int(x) for x in extra.get("seeds", "1024,5").split(",")]
for variant in extra.get("variants", "default").split(";"):
    kw = dict(kv.split("=") for kv in variant.split(",") if "=" in kv)

    def tweak(s):
        if "parts" in kw:
            s.hub_parts = int(kw["parts"])
        if "rounds" in kw:
            s.hub_rounds = bool(int(kw["rounds"]))
        if "lerp" in kw:
            s.hub_lerp = bool(int(kw["lerp"]))
    solver_kw = {}
    if "hub_rows" in kw:
        solver_kw["hub_rows"] = int(kw["hub_rows"])
    if kw.get("device") == "1":
        solver_kw["device_sampling"] = True
    if "fidelity" in kw:
        solver_kw["fidelity"] = kw["fidelity"]
    if "tune" in kw:  # gvk_set_tuning key:value, e.g. tune=12:2 (rounds of two entries per task whatever the engine says)
        from graphvite_amd.kernels import HipKernels
        HipKernels().set_tuning(int(kw["tune"].split(":")[0]), int(kw["tune"].split(":")[1]))
    base = T.JOBS[job][3].get("shuffle_base")
    if "sb" in kw:  # the pseudo shuffle's base (graph.cuh:362-364,439-441) — a large one mixes the CPU samplers' walk-ordered pools
        T.JOBS[job][3]["shuffle_base"] = int(kw["sb"])
    aucs = []
    for seed in seeds:
        spec = T.JOBS[job][7]  # the job's optimizer (None: the solver's default SGD)
        optimizer = None if spec is None else getattr(T.gv.optimizer, spec[0])(*spec[1:])
        auc, reference, info = T.train(job, seed, tweak=tweak, optimizer=optimizer, **solver_kw)
        aucs.append(auc)
    if "sb" in kw:
        T.JOBS[job][3]["shuffle_base"] = base
    reference = reference[~np.isnan(reference)]
    print("%s [%s]: AUC %s mean %.6f | reference %.6f | difference %+.6f | %s" % (job, variant, " ".join("%.6f" % a for a in aucs), np.mean(aucs),
                                                                                 reference.mean(), np.mean(aucs) - reference.mean(), info), flush=True)
