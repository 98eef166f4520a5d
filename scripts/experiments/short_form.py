"""Experiment (GPU): why the driver's short form (5 warm-up + 20 timed launches) shows a slower kernel than 400
back-to-back launches.  Per-launch durations (HIP events around every single launch) of the benchmark batch shape:
a cold sequence right after a device synchronize, the same after a host sleep, and a long back-to-back sequence.

    python scripts/experiments/short_form.py
"""
import logging
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import graphvite_amd as gv  # noqa: E402
from graphvite_amd import synthetic  # noqa: E402

gv.init_logging(logging.ERROR)
N, E, B = 1000000, 10000000, 100000
g = gv.graph.Graph()
g.load(synthetic.power_law_edges(N, E, seed=1024))
s = gv.solver.GraphSolver(128, num_sampler_per_worker=16, seed=1024)
s.build(g, optimizer=gv.optimizer.SGD(0.025, 0.005, "linear"), num_partition=1, batch_size=B, episode_size=250)
session = s.session(model="LINE", num_epoch=100, augmentation_step=1, log_frequency=1 << 30)
pools = session.new_host_pools()
session.fill(pools)
dev = session.upload(pools, group=False)
pool = dev[(0, 0)]


def launches(n, first=0):
    events = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b = (first + i) % 250
        e0.record()
        session.train_block(0, 0, pool[b * B * 2:], 1)
        e1.record()
        events.append((e0, e1))
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) * 1e3 for a, b in events])


def block(n, first=0):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    session.train_block(0, 0, pool[first * B * 2:], n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


launches(4)
for label, nap in (("after synchronize", 0.0), ("after 50 ms of idle", 0.05), ("after 1 s of idle", 1.0)):
    for rep in range(3):
        torch.cuda.synchronize()
        time.sleep(nap)
        us = launches(25, first=rep * 25)
        print("%s, rep %d: first 5 %s | launches 6-25 mean %.2f us | all 25 mean %.2f" % (
            label, rep, " ".join("%.1f" % x for x in us[:5]), us[5:].mean(), us.mean()), flush=True)
for n in (20, 20, 20, 50, 100, 200, 200):
    torch.cuda.synchronize()
    print("one gvk_train_episode of %d batches after synchronize: %.2f us per batch" % (n, block(n)), flush=True)
# the driver's form: warm-up 5, synchronize, 20 timed — against 200 warm-up launches first
for warm in (5, 5, 200, 200):
    block(warm)
    torch.cuda.synchronize()
    print("warm-up %d, synchronize, 20 timed: %.2f us per batch" % (warm, block(20, 30)), flush=True)
