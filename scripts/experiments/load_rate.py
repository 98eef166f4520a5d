"""Graph load time (ndarray and text file) and sampler rate, with and without MADV_HUGEPAGE (GVS_NO_HUGEPAGE=1)."""
import logging, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import graphvite_amd as gv
from graphvite_amd import synthetic
gv.init_logging(logging.ERROR)
e = synthetic.power_law_edges(1_000_000, 10_000_000, seed=0)
path = "/tmp/gv_edges.txt"
if not os.path.exists(path):
    np.savetxt(path, e, fmt="%d")
g = gv.graph.Graph()
times = []
for i in range(4):
    t = time.time(); g.load(e); times.append(time.time() - t)
print("huge pages %s | load ndarray 10M edges: %s s" % ("off" if os.environ.get("GVS_NO_HUGEPAGE") else "on",
      " ".join("%.2f" % x for x in times)), flush=True)
times = []
for i in range(2):
    t = time.time(); g.load(file_name=path); times.append(time.time() - t)
print("   load text file 10M edges: %s s" % " ".join("%.2f" % x for x in times), flush=True)
print("   thp:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:",
      open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip(), flush=True)
