set -x
python - <<'PY'
import json, logging, sys, time
sys.path.insert(0, "scripts")
import measure_configs as m
import graphvite_amd as gv
import torch
from graphvite_amd import synthetic
from graphvite_amd.kernels import HipKernels
gv.init_logging(logging.ERROR)
graph = gv.graph.Graph()
graph.load(synthetic.power_law_edges(1138499, 4945382, seed=2024))
orig = HipKernels.sample_walks_blocks
def timed(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    used = orig(self, *a, **k)
    torch.cuda.synchronize()
    print("sample_walks_blocks: %.3f s, rounds %s, walks %d" % (time.perf_counter() - t0, self.walk_rounds, used), flush=True)
    return used
HipKernels.sample_walks_blocks = timed
m.run(graph, "configs[2] P=4 on one GPU, device sampling", "DeepWalk", 100, 4, num_partition=4, device_sampling=True)
m.run(graph, "configs[3] P=4 on one GPU, device sampling", "node2vec", 100, 4, p=0.25, q=0.25, num_partition=4, device_sampling=True)
m.run(graph, "configs[2] P=4 on one GPU, device sampling", "DeepWalk", 300, 4, num_partition=4, device_sampling=True)
PY
