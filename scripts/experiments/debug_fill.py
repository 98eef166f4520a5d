"""Debug: CPU edge-sampler fill rate — standalone, repeated, pinned vs pageable pools."""
import logging, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import graphvite_amd as gv
from graphvite_amd import synthetic, hostlib
gv.init_logging(logging.ERROR)
g = gv.graph.Graph(); g.load(synthetic.power_law_edges(1000000, 10000000, seed=1024))
part, local, _ = hostlib.partition(g.vertex_weights, 1)
s = hostlib.Sampler(g, part, local, 1, seed=1)
t = time.time(); s.prepare("edge", num_thread=256); print("prepare %.2f s" % (time.time() - t))
n = 25000000
pool = torch.empty(2 * n, dtype=torch.int32, pin_memory=True)
for label, OFF in (("unpinned", -1),):
    for T in (255, 128):
        times = []
        for rep in range(8):
            t = time.time(); s.fill({(0, 0): pool}, n, "edge", 4 * T, sample_batch_size=4000, os_threads=T, cpu_offset=OFF); times.append(time.time() - t)
        print("%-9s threads %3d: " % (label, T) + " ".join("%.0f M/s" % (n / x / 1e6) for x in times), flush=True)
