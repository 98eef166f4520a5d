"""CPU sampler fill rate on its own (no GPU): edge sampler and walk sampler on the benchmark graph."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from graphvite_amd import graph as G, hostlib, synthetic
from graphvite_amd.base import cpu_budget

N, E = 1_000_000, 10_000_000
threads = int(sys.argv[1]) if len(sys.argv) > 1 else cpu_budget()
g = G.Graph(); g.load(synthetic.power_law_edges(N, E, seed=0))
V = g.num_vertex; part = np.zeros(V, np.int32); local = np.arange(V, dtype=np.uint32)
s = hostlib.Sampler(g, part, local, 1, 0)
pool_size = 20_000_000
pool = np.zeros((pool_size, 2), np.uint32)
for mode, kw in (("edge", {}), ("walk", dict(walk_length=2, augmentation_step=2)), ("walk", dict(walk_length=40, augmentation_step=5))):
    s.prepare(mode, num_thread=threads)
    for rep in range(3):
        t = time.perf_counter()
        s.fill([pool], pool_size, mode, threads * 4, os_threads=threads, **kw)
        dt = time.perf_counter() - t
    print("mode", mode, kw, "threads", threads, "%.1f M samples/s" % (pool_size / dt / 1e6), "(%.1f per thread)" % (pool_size / dt / 1e6 / threads))
