"""The pybind11 module `libgraphvite` (graphvite_amd/lib/libgraphvite.so — the module the reference's Python package
loads, src/graphvite.cu:28-106 + include/bind.h) on a machine without a GPU: surface, graph store, optimizers, and the
reference's own template dispatch (python/graphvite/helper.py) run against it where the reference tree is present."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODULE = os.path.join(ROOT, "graphvite_amd", "lib", "libgraphvite.so")
REFERENCE = "/root/reference/python/graphvite"


def load_module():
    if "libgraphvite" in sys.modules:
        return sys.modules["libgraphvite"]
    spec = importlib.util.spec_from_file_location("libgraphvite", MODULE)
    lib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lib)
    sys.modules["libgraphvite"] = lib
    return lib


def reference_package(lib):
    """The reference's helper / graph / solver / optimizer modules, loaded from where they lie as package `graphvite`
    over `lib` (nothing is copied; `cfg` is what its base.load_global_config() yields by default)."""
    package = types.ModuleType("graphvite")
    package.__path__ = [REFERENCE]
    package.lib = lib
    package.auto = lib.auto
    package.cfg = types.SimpleNamespace(float_type=lib.dtype.float32, index_type=lib.dtype.uint32)
    sys.modules["graphvite"] = package
    out = {}
    for name in ("helper", "graph", "solver", "optimizer"):
        spec = importlib.util.spec_from_file_location("graphvite." + name, os.path.join(REFERENCE, name + ".py"))
        module = importlib.util.module_from_spec(spec)
        sys.modules["graphvite." + name] = module
        spec.loader.exec_module(module)
        setattr(package, name, module)
        out[name] = module
    return out


def test_module_surface():
    lib = load_module()
    assert lib.__version__ == "0.2.2" and lib.auto == 0
    assert [lib.dtype2name[t] for t in (lib.dtype.uint32, lib.dtype.uint64, lib.dtype.float32, lib.dtype.float64)] == list("jmfd")
    assert (lib.KiB(2), lib.MiB(1), lib.GiB(1)) == (2048, 1 << 20, 1 << 30)
    assert (lib.INFO, lib.WARNING, lib.ERROR, lib.FATAL) == (0, 1, 2, 3)
    assert lib.io.size_string(3 << 20) == "3 MiB" and lib.io.yes_no(True) == "yes"
    assert lib.io.header("Graph").center(40).count("-") == 33 and lib.io.block("x").count("<") == 40
    for dim in (32, 64, 96, 128, 256, 512):  # src/graphvite.cu:52-59
        cls = getattr(lib.solver, "GraphSolver_%d_f_j" % dim)
        assert cls.__name__ == "GraphSolver" and "dim (int)" in cls.__doc__
        for member in ("num_partition", "num_negative", "optimizer", "negative_sample_exponent", "negative_weight",
                       "model", "num_epoch", "resume", "episode_size", "batch_size", "augmentation_step",
                       "random_walk_length", "random_walk_batch_size", "shuffle_base", "p", "q", "positive_reuse",
                       "log_frequency", "num_worker", "num_sampler", "gpu_memory_limit", "gpu_memory_cost",
                       "vertex_embeddings", "context_embeddings", "build", "train", "predict", "clear"):  # bind.h:408-503
            assert hasattr(cls, member), member
    lib.init_logging(lib.WARNING)


def test_graph_store(tmp_path):
    lib = load_module()
    g = lib.graph.Graph_j()
    g.load([("a", "b"), ("b", "c"), ("c", "a"), ("a", "d")])
    assert (g.num_vertex, g.num_edge, g.as_undirected, g.normalization) == (4, 4, True, False)
    assert g.name2id == {"a": 0, "b": 1, "c": 2, "d": 3} and g.id2name == ["a", "b", "c", "d"]
    assert "#vertex: 4, #edge: 4" in repr(g) and type(g).__name__ == "Graph"
    path = str(tmp_path / "graph.txt")
    g.save(path, weighted=True, anonymous=False)
    h = lib.graph.Graph_j()
    h.load(path, as_undirected=False)
    assert h.num_vertex == 4 and h.num_edge == 8  # the saved graph holds both directions
    w = lib.graph.Graph_j()
    w.load(weighted_edge_list=[("x", "y", 2.0), ("y", "z", 0.5)], as_undirected=False, normalization=False)
    assert w.num_edge == 2 and not w.as_undirected
    with pytest.raises(ValueError):
        lib.graph.Graph_j().load(str(tmp_path / "missing.txt"))
    corpus = tmp_path / "corpus.txt"
    corpus.write_text("the cat sat on the mat\nthe dog sat on the log\n" * 3)
    words = lib.graph.WordGraph_j()
    words.load(str(corpus), window=2, min_count=2)
    assert words.num_vertex == 7 and "WordGraph" in repr(words) and isinstance(words, lib.graph.Graph_j)


def test_optimizers():
    lib = load_module()
    o = lib.optimizer
    sgd = o.SGD(0.025, 0.005)
    assert (sgd.type, sgd.schedule.type) == ("SGD", "linear") and sgd.lr == pytest.approx(0.025)
    assert o.Momentum().momentum == pytest.approx(0.999) and o.AdaGrad().epsilon == pytest.approx(1e-10)
    rms, adam = o.RMSprop(alpha=0.9), o.Adam(1e-3, 0, 0.9, 0.99, 1e-8, "constant")
    assert rms.alpha == pytest.approx(0.9) and (adam.beta1, adam.schedule.type) == (pytest.approx(0.9), "constant")
    custom = o.SGD(schedule=lambda batch_id, num_batch: 1 - batch_id / num_batch)
    assert custom.schedule.type == "custom" and custom.schedule.schedule_function(1, 4) == pytest.approx(0.75)
    assert o.Optimizer(lib.auto).type == "Default" and o.Optimizer(0.5).lr == pytest.approx(0.5)
    assert isinstance(sgd, o.Optimizer) and "weight decay" in repr(sgd)
    with pytest.raises(ValueError):
        o.LRSchedule("cosine")


def test_solver_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = load_module()
    with pytest.raises(ValueError, match="No GPU devices found"):  # solver.h:176; there is no CPU training path
        lib.solver.GraphSolver_128_f_j()
    with pytest.raises(ValueError, match="No GPU devices found"):  # the keyword beyond the reference's is accepted
        lib.solver.GraphSolver_128_f_j(device_ids=[0], num_sampler_per_worker=1, device_sampling=True)
    with pytest.raises(TypeError):
        lib.solver.GraphSolver_128_f_j(sample_on_device=True)
    import ctypes as C
    from graphvite_amd import _lib
    gvk = _lib.lib()
    gvk.gvx_solver_set.restype, gvk.gvx_solver_set.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int64]
    assert gvk.gvx_solver_set(None, 1, 1) == _lib.GVK_EINVAL and b"null solver" in gvk.gvk_last_error()


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is not on this machine")
def test_reference_template_dispatch_resolves_to_this_module():
    """python/graphvite/helper.py as written: find_all_templates / make_helper_class over lib.graph and lib.solver, then
    `graphvite.solver.GraphSolver(dim, float_type, index_type)` -> `lib.solver.GraphSolver_<dim>_f_j`."""
    lib = load_module()
    ref = reference_package(lib)
    assert sorted(ref["helper"].find_all_templates(lib.solver)) == ["GraphSolver"]
    assert sorted(ref["helper"].find_all_templates(lib.graph)) == ["Graph", "WordGraph"]
    graph = ref["graph"].Graph()  # TemplateHelper.__new__ -> lib.graph.Graph_j()
    assert type(graph) is lib.graph.Graph_j
    graph.load([("0", "1"), ("1", "2")])
    assert graph.num_vertex == 3
    assert type(ref["graph"].WordGraph(lib.dtype.uint32)) is lib.graph.WordGraph_j
    assert "**dim**: 32, 64, 96, 128, 256, 512" in ref["solver"].GraphSolver.__doc__
    with pytest.raises(AttributeError, match="Can't find an instantiation"):
        ref["solver"].GraphSolver(100)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(ValueError, match="No GPU devices found"):  # resolved to GraphSolver_128_f_j, whose constructor ran
            ref["solver"].GraphSolver(128, device_ids=[0])
    assert type(ref["optimizer"].Optimizer("SGD", 0.1)) is lib.optimizer.SGD
    assert ref["optimizer"].Optimizer().type == "Default"


def build_c_client(out_dir):
    """tests/c/abi_client.c — a plain C host of include/gvs.h + include/gvx.h — compiled with gcc against libgvk.so."""
    import subprocess
    binary = os.path.join(str(out_dir), "abi_client")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_client.c"), "-L", os.path.join(ROOT, "graphvite_amd"), "-lgvk",
                           "-Wl,-rpath," + os.path.join(ROOT, "graphvite_amd"), "-lm", "-o", binary])
    return binary


def test_plain_c_host_links_and_fails_loudly_without_a_gpu(tmp_path):
    """The boundary is a C ABI: a C11 program with nothing but the three headers and -lgvk loads a graph and asks for
    a solver; without a GPU the library says so and the program exits with its "no GPU" code."""
    import subprocess
    import torch
    binary = build_c_client(tmp_path)
    run = subprocess.run([binary, "2000", "20000"], capture_output=True, text=True, timeout=300)
    assert "graph: 2000 vertices, 20000 edges" in run.stdout
    if not torch.cuda.is_available():
        assert run.returncode == 3 and "No GPU devices found" in run.stderr
