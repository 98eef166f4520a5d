"""Generates tests/golden/reference_application.npz: macro / micro F1 of the REFERENCE's own node-classification routine —
python/graphvite/application/application.py:456-533 `linear_classification` with application/network.py `NodeClassifier`,
imported from /root/reference where they lie — on fixed embeddings and labels (drawn here from a numpy seed), for a few
portions, with and without normalization.  The reference's module is loaded as package `graphvite` over small stand-ins for
what it imports besides numpy / torch: `easydict`, `future.builtins` (python-2 compatibility names), the compiled library
`lib`, and `Tensor.cuda` / `Module.cuda` (the routine moves its tensors to a GPU; this container has none: the arithmetic is
the same on the CPU).  `np.int` (removed from numpy 1.24) is given back its meaning for the one line that uses it.

    python tests/golden/make_application_golden.py
"""
import builtins
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference/python/graphvite"
PATH = os.path.join(HERE, "reference_application.npz")
CASES = ((0.1, False), (0.3, False), (0.3, True), (0.5, False))  # (portion, normalization)
TIMES, PATIENCE, SEED = 2, 100, 7


def fixed_problem():
    """600 nodes, 32 dims, 4 overlapping classes: what tests/test_host_cpu.py rebuilds from the same seed."""
    rng = np.random.default_rng(5)
    n, dim, classes = 600, 32, 4
    membership = rng.random((n, classes)) < 0.3
    membership[np.arange(n), rng.integers(0, classes, n)] = True  # every node has at least one label
    centres = rng.normal(0, 1, (classes, dim))
    embeddings = (membership.astype(np.float64) @ centres + rng.normal(0, 2.5, (n, dim))).astype(np.float32)
    return embeddings, membership.astype(np.int64)


def reference_application():
    easydict = types.ModuleType("easydict")
    easydict.EasyDict = dict
    future = types.ModuleType("future")
    future.builtins = types.ModuleType("future.builtins")
    future.builtins.str, future.builtins.map, future.builtins.range = builtins.str, builtins.map, builtins.range
    sys.modules.update({"easydict": easydict, "future": future, "future.builtins": future.builtins})
    package = types.ModuleType("graphvite")
    package.__path__ = [REFERENCE]
    package.lib = types.SimpleNamespace(auto=0)
    class AnyConfig(object):  # cfg.<anything>: default values of keyword arguments this script never uses
        def __getattr__(self, name):
            return 0
    package.auto, package.cfg = 0, AnyConfig()
    package.graph = types.ModuleType("graphvite.graph")
    package.solver = types.ModuleType("graphvite.solver")
    sys.modules.update({"graphvite": package, "graphvite.graph": package.graph, "graphvite.solver": package.solver})
    for name in ("util", "application.network", "application.application"):
        path = os.path.join(REFERENCE, *name.split(".")) + ".py"
        if name.startswith("application.") and "graphvite.application" not in sys.modules:
            sub = types.ModuleType("graphvite.application")
            sub.__path__ = [os.path.join(REFERENCE, "application")]
            sys.modules["graphvite.application"] = sub
        spec = importlib.util.spec_from_file_location("graphvite." + name, path)
        module = importlib.util.module_from_spec(spec)
        sys.modules["graphvite." + name] = module
        spec.loader.exec_module(module)
    return sys.modules["graphvite.application.application"]


def main():
    np.int = int  # application.py:469
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    application = reference_application()
    embeddings, labels = fixed_problem()
    out = {"cases": np.array(CASES, np.float64), "times_patience_seed": np.array([TIMES, PATIENCE, SEED], np.int64)}
    for i, (portion, normalization) in enumerate(CASES):
        np.random.seed(SEED)
        torch.manual_seed(SEED)
        result = application.linear_classification((embeddings, np.asmatrix(labels), portion, bool(normalization), TIMES, PATIENCE, 0))
        macro, micro = result["macro-F1@%g%%" % (portion * 100)], result["micro-F1@%g%%" % (portion * 100)]
        print("portion %g normalization %d: macro-F1 %.6f micro-F1 %.6f" % (portion, normalization, macro, micro))
        out["f1_%d" % i] = np.array([macro, micro], np.float64)
    np.savez_compressed(PATH, **out)


if __name__ == "__main__":
    main()
