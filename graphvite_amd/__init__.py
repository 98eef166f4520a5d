"""graphvite_amd — MI355X-native node-embedding training (LINE / DeepWalk / node2vec): a drop-in for the
GraphSolver / GraphApplication.train path of GraphVite (`import graphvite_amd as gv`).

    gv.graph.Graph, gv.solver.GraphSolver, gv.optimizer.{Optimizer, SGD, ...}, gv.application.GraphApplication,
    gv.dtype / gv.uint32 / gv.float32, gv.auto, gv.KiB/MiB/GiB, gv.init_logging, gv.io

The training arithmetic lives in libgvk.so (hand-written gfx950 HIP kernels behind the C ABI of include/gvk.h);
the CPU samplers and the graph store in the same library behind include/gvs.h.  See DESIGN.md, INTEGRATION.md.
"""
import logging

from .base import GiB, KiB, MiB, auto, dtype, dtype2name, float32, float64, init_logging, io, uint32, uint64

__version__ = "0.1.0"

init_logging(logging.INFO)

from . import graph, optimizer, solver  # noqa: E402
from . import application  # noqa: E402

__all__ = ["graph", "optimizer", "solver", "application", "dtype", "auto", "init_logging", "io", "KiB", "MiB", "GiB",
           "uint32", "uint64", "float32", "float64", "dtype2name", "__version__"]
