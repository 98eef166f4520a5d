"""BASELINE.json configs[0]: the reference's quick start (config/demo/quick_start.yaml — LINE, dim 128, 2000 epochs,
augmentation_step 2, batch 100000, episode 500) on the BlogCatalog-sized stand-in of the parity tests (10 312 nodes /
333 983 edges, hub-heavy with communities; the real dataset needs the network).  Prints stage times and the
link-prediction AUC."""
import logging
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphvite_amd as gv
from graphvite_amd import synthetic

gv.init_logging(logging.WARNING)
edges = synthetic.hub_community_edges(10312, 333983, gamma=2.8, num_community=39, p_in=0.7, seed=1024)
train, (valid, test) = synthetic.link_prediction_split(edges, (100, 1, 1))
app = gv.application.GraphApplication(dim=128)
t0 = time.time()
app.load(edge_list=train)
t1 = time.time()
app.build(optimizer=gv.optimizer.SGD(0.025, 0.005), num_partition=gv.auto, num_negative=1, batch_size=100000,
          episode_size=500)
t2 = time.time()
app.train(model="LINE", num_epoch=2000, negative_weight=5, augmentation_step=2, random_walk_length=40,
          random_walk_batch_size=100, log_frequency=1000)
t3 = time.time()
H, T, Y = test
result = app.evaluate("link prediction", H=[str(h) for h in H], T=[str(t) for t in T], Y=Y.tolist(),
                      filter_H=[str(h) for h in train[:, 0]], filter_T=[str(t) for t in train[:, 1]])
t4 = time.time()
print("quick start: load %.2f s, build %.2f s, train %.2f s (%d batches, %.1f M edge-samples/s), evaluate %.2f s, %s"
      % (t1 - t0, t2 - t1, t3 - t2, app.solver.batch_id, app.solver.batch_id * 1e5 / (t3 - t2) / 1e6, t4 - t3, result))
print("timing", app.solver.timing)
