"""Where do ~30 ms go in a P=16 single-GPU run? Per-block CPU time and GPU event time."""
import logging, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import graphvite_amd as gv
from graphvite_amd import synthetic
gv.init_logging(logging.ERROR)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = gv.graph.Graph(); g.load(synthetic.community_edges(1000000, 10000000, num_community=1000, seed=1024))
s = gv.solver.GraphSolver(128, num_sampler_per_worker=16, seed=1)
s.build(g, optimizer=gv.optimizer.SGD(0.025, 0.005), num_partition=P, batch_size=100000, episode_size=8)
session = s.session(model="LINE", num_epoch=100, augmentation_step=1, log_frequency=1 << 30)
pools = session.new_host_pools(); session.fill(pools); dev = session.upload(pools)
blocks = session.blocks
for rep in range(2):
    cpu, evs = [], []
    torch.cuda.synchronize()
    for step in range(64):
        hp, tp = blocks[step % len(blocks)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter(); e0.record(); session.train_block(hp, tp, dev[(hp, tp)], 8); e1.record()
        cpu.append(time.perf_counter() - t); evs.append((e0, e1))
    torch.cuda.synchronize()
    gpu = [a.elapsed_time(b) for a, b in evs]
    print("P", P, "rep", rep, "cpu ms per block: median %.3f max %.3f at %d | gpu ms per block: median %.3f max %.3f at %d" % (
        np.median(cpu) * 1e3, max(cpu) * 1e3, int(np.argmax(cpu)), np.median(gpu), max(gpu), int(np.argmax(gpu))))
    print("   gpu first 24:", [round(x, 2) for x in gpu[:24]])
