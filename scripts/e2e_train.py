"""End-to-end GraphSolver.train() throughput on the benchmark graph (configs[1]): CPU samplers feeding the GPU
(the north_star pipeline) vs device-side positive sampling, and the positive_reuse lever.  Prints one line per
configuration; numbers go into DESIGN.md §6."""
import logging
import sys
import time

sys.path.insert(0, ".")
import graphvite_amd as gv
from graphvite_amd import synthetic

gv.init_logging(logging.ERROR)
N, E = 1000000, 10000000
graph = gv.graph.Graph()
t = time.time()
graph.load(synthetic.power_law_edges(N, E, seed=1024))
print("graph build+load %.1f s" % (time.time() - t), flush=True)
for label, kw, train_kw in (("cpu-samplers", {}, {}), ("cpu-samplers reuse=4", {}, {"positive_reuse": 4}),
                            ("device-sampling", {"device_sampling": True}, {}),
                            ("cpu-samplers aug=2 (walks)", {}, {"augmentation_step": 2}),
                            ("DeepWalk aug=5", {}, {"model": "DeepWalk", "augmentation_step": 5}),
                            ("device-sampling DeepWalk aug=5", {"device_sampling": True},
                             {"model": "DeepWalk", "augmentation_step": 5}),
                            ("device-sampling node2vec aug=5", {"device_sampling": True},
                             {"model": "node2vec", "augmentation_step": 5, "p": 0.25, "q": 0.25})):
    solver = gv.solver.GraphSolver(128, **kw)
    solver.build(graph, batch_size=100000, episode_size=250)
    cfg = dict(model="LINE", num_epoch=100, augmentation_step=1, log_frequency=1 << 30)
    cfg.update(train_kw)
    t = time.time()
    solver.train(**cfg)
    el = time.time() - t
    tm = solver.timing
    print("%-28s %8.1f M edge-samples/s in the episode loop | train() %.2f s = configure %.2f + upload %.2f + "
          "%d batches %.2f + write-back %.2f (%d sampler threads)" % (
              label, tm["batches"] * 100000 / tm["episodes"] / 1e6, el, tm["configure"], tm["upload"], tm["batches"],
              tm["episodes"], tm["write_back"], solver.num_sampler_per_worker), flush=True)
    if tm.get("loop"):
        print("      host loop: " + ", ".join("%s %.2f s" % kv for kv in tm["loop"].items()), flush=True)
