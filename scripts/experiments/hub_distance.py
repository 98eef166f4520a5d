"""Experiment (GPU): how far the executor's rows end from the reference's SEQUENTIAL loop (gpu/graph.cuh:54-94 in sample order = the oracle's
gvo_train) at batch level — per hub rank: distance / the sequential row's own movement, the ratio of the two movements (> 1: the
executor moves a row further than the sequential loop does) and their cosine — after `batches` batches from a state trained for `warm`
batches by the executor itself.  Same setting as tests/test_hub_chains_gpu.py::test_hub_rows_of_headline_batches_stay_with_the_sequential_loop.

    python scripts/experiments/hub_distance.py [gamma=2.3 nodes=1000000 edges=10000000 graph_seed=1024 warm=2000 batches=20 parts=8,32 hub_hits=1.0]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from graphvite_amd import kernels as K, synthetic  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

extra = dict(kv.split("=", 1) for kv in sys.argv[1:])
DEV, SEED = "cuda:0", 5
dim, B, k = 128, 100000, 1
n, e = int(extra.get("nodes", 1000000)), int(extra.get("edges", 10000000))
warm, batches = int(extra.get("warm", 2000)), int(extra.get("batches", 20))
total = int(extra.get("total", 5000))
hip, oracle = K.HipKernels(), Oracle()
edges = synthetic.power_law_edges(n, e, gamma=float(extra.get("gamma", 2.3)), seed=int(extra.get("graph_seed", 1024)))
degree = synthetic.degrees(edges, n)
order = np.argsort(-degree, kind="stable")
local = np.empty(n, np.int64)
local[order] = np.arange(n)
chunk = 20


def samples(first, count):
    out = np.empty((count * B, 2), np.uint32)
    for i in range(count):
        rng = np.random.default_rng(1000 + first + i)
        pick, flip = rng.integers(0, e, B), rng.random(B) < 0.5
        out[i * B:(i + 1) * B, 1] = local[np.where(flip, edges[pick, 0], edges[pick, 1])]
        out[i * B:(i + 1) * B, 0] = local[np.where(flip, edges[pick, 1], edges[pick, 0])]
    return out


w = degree[order] ** np.float32(0.75)
table = K.packed_to_device(K.alias_build(w)[2], DEV)
share = degree[order] / degree.sum()
opt = K.OptimizerSpec("SGD", 0.025, 0.005)
print("top hub: %.2f %% of the endpoints = %d head samples of a batch" % (100 * share[0], B * share[0]), flush=True)


def executor(parts, hits):
    kv = kc = int(min(16384, np.count_nonzero(B * share >= hits)))
    ws = torch.zeros(hip.hot_plan(dim, B, k, kv, kc, chunk, parts), dtype=torch.uint8, device=DEV)
    loss = torch.zeros(B, device=DEV)

    def run(tv, tc, first_batch, count):
        for at in range(first_batch, first_batch + count, chunk):
            m = min(chunk, first_batch + count - at)
            dpool = torch.from_numpy(samples(at, m).view(np.int32)).to(DEV)
            hip.hot_build(dim, ws, dpool, B, m, k, table, SEED, at, kv, kc, parts=parts)
            hip.train_episode_hot(tv, tc, dpool, loss, opt, k, 5.0, table, SEED, at, total, m, B, ws, kv, kc, workspace_batches=m, parts=parts)
        torch.cuda.synchronize()
    return run, kv


for item in extra.get("tune", "").split(","):  # gvk_set_tuning keys (include/gvk.h GVK_TUNE_*), e.g. tune=12:4 (rounds of four entries per task)
    if item:
        hip.set_tuning(int(item.split(":")[0]), int(item.split(":")[1]))
parts_list = [int(x) for x in extra.get("parts", "8").split(",")]
hits = float(extra.get("hub_hits", 1.0))
rng = np.random.default_rng(11)
v = rng.uniform(-0.5 / dim, 0.5 / dim, (n, dim)).astype(np.float32)
c = np.zeros((n, dim), np.float32)
tv, tc = torch.from_numpy(v).to(DEV), torch.from_numpy(c).to(DEV)
run, kv = executor(parts_list[0], hits)
t0 = time.time()
run(tv, tc, 0, warm)
print("warm-up: %d batches as %d parts, %d hub rows per table, %.1f s" % (warm, parts_list[0], kv, time.time() - t0), flush=True)
v0, c0 = tv.cpu().numpy(), tc.cpu().numpy()
sv, sc = v0.copy(), c0.copy()
negs = torch.zeros(B * k, dtype=torch.int32, device=DEV)
t0 = time.time()
for b in range(batches):
    hip.negative_draw(table, SEED, warm + b, negs, B, k)
    nb = negs.cpu().numpy().view(np.uint32).reshape(B, k)
    oracle.train(sv, sc, samples(warm + b, 1), nb, oracle.lr(0.025, True, warm + b, total), 0.005, 5.0)
print("sequential loop: %d batches, %.1f s" % (batches, time.time() - t0), flush=True)
tail = np.arange(kv + 1000, kv + 3000)  # rows that are not hub rows
for parts in parts_list:
    run, kv = executor(parts, hits)
    dv, dc = torch.from_numpy(v0).to(DEV), torch.from_numpy(c0).to(DEV)
    run(dv, dc, warm, batches)
    gv_, gc_ = dv.cpu().numpy(), dc.cpu().numpy()
    for name, got, want, start in (("head", gv_, sv, v0), ("context", gc_, sc, c0)):
        for label, rows in (("hub rows 0-9", np.arange(10)), ("hub rows 10-99", np.arange(10, 100)), ("hub rows 100-999", np.arange(100, 1000)), ("other rows", tail)):
            moved_seq = np.linalg.norm(want[rows] - start[rows], axis=1)
            moved = np.linalg.norm(got[rows] - start[rows], axis=1)
            off = np.linalg.norm(got[rows] - want[rows], axis=1) / np.maximum(moved_seq, 1e-30)
            cos = ((got[rows] - start[rows]) * (want[rows] - start[rows])).sum(1) / np.maximum(moved * moved_seq, 1e-30)
            keep = moved_seq > 0
            print("parts %3d | %-7s | %-16s | distance / sequential movement: median %.3f max %.3f | movement ratio median %.3f | cosine median %.3f | norm ratio %.4f" % (
                parts, name, label, np.median(off[keep]), off[keep].max(), np.median(moved[keep] / moved_seq[keep]), np.median(cos[keep]),
                np.median(np.linalg.norm(got[rows], axis=1) / np.maximum(np.linalg.norm(want[rows], axis=1), 1e-30))), flush=True)
