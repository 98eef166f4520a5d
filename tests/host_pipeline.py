"""TEST INFRASTRUCTURE: one training through the HOST build of the engine (tests/hostdev/build/libgvk_host.so: the engine's
own sources over a host stand-in for HIP, its kernels being the SEQUENTIAL CPU oracle) in a process of its own — the
pipeline the GPU parity tests compare the HIP path with: same graph, same sampler streams, same negatives (RNG contract),
same init, every batch applied one sample after the other.

    python tests/host_pipeline.py config.json out.npz          (the caller sets GVK_LIBRARY)
config: {"edges": path to an .npy edge list, "dim": 128, "solver": {...GraphSolver kwargs}, "build": {...}, "train": {...}}
"""
import json
import logging
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HOST_LIBRARY = os.path.join(ROOT, "tests", "hostdev", "build", "libgvk_host.so")


def run_in_subprocess(edges, dim, solver, build, train, tmp_dir, timeout=3600):
    """Called by the tests: trains through the host build, returns (vertex, context, batch_id, num_batch)."""
    import subprocess
    np.save(os.path.join(tmp_dir, "edges.npy"), np.asarray(edges))
    config = {"edges": os.path.join(tmp_dir, "edges.npy"), "dim": dim, "solver": solver, "build": build, "train": train}
    with open(os.path.join(tmp_dir, "config.json"), "w") as f:
        json.dump(config, f)
    out = os.path.join(tmp_dir, "out.npz")
    env = dict(os.environ, GVK_LIBRARY=HOST_LIBRARY, GVK_ALLOW_TEST_LIBRARY="1")
    run = subprocess.run([sys.executable, os.path.abspath(__file__), os.path.join(tmp_dir, "config.json"), out],
                         capture_output=True, text=True, timeout=timeout, env=env)
    if run.returncode != 0:
        raise RuntimeError("host pipeline failed: %s" % run.stderr[-2000:])
    data = np.load(out)
    return data["vertex"], data["context"], int(data["batch_id"]), int(data["num_batch"])


def main():
    config = json.load(open(sys.argv[1]))
    import graphvite_amd as gv
    assert gv._lib.lib().gvh_is_host_build(), "GVK_LIBRARY must point to the host build"
    gv.init_logging(logging.ERROR)
    g = gv.graph.Graph()
    g.load(np.load(config["edges"]))
    kw = dict(config["solver"])
    if kw.get("pair_order") == "auto":
        kw["pair_order"] = gv.auto
    s = gv.solver.GraphSolver(config["dim"], **kw)
    s.build(g, **config["build"])
    s.train(**config["train"])
    np.savez(sys.argv[2], vertex=s.vertex_embeddings, context=s.context_embeddings, batch_id=s.batch_id, num_batch=s.num_batch)


if __name__ == "__main__":
    main()
