"""The product's host runtime (libgvk.so: include/gvk.h host entries + include/gvs.h) against the CPU oracle —
integer / index work, so the bar is bit-exact.  No GPU needed: nothing here launches a kernel."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import graphvite_amd as gv
from graphvite_amd import _lib, hostlib, synthetic
from graphvite_amd import kernels as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = set()
    for header in ("gvk.h", "gvs.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(gv[ks]_[a-z0-9_]+)\s*\(", text))
    assert len(names) >= 40
    for name in sorted(names):
        assert hasattr(lib, name), "libgvk.so does not export %s" % name
    assert b"gfx950" in lib.gvk_version()


def test_error_codes_not_aborts():
    lib = _lib.lib()
    w = np.ones(4, np.float32)
    out = np.zeros(4, np.float32)
    alias = np.zeros(4, np.uint32)
    assert lib.gvk_alias_build(w.ctypes.data, 0, out.ctypes.data, alias.ctypes.data, 4, None) == _lib.GVK_EINVAL
    assert b"empty" in lib.gvk_last_error()
    assert lib.gvk_alias_build(w.ctypes.data, 4, out.ctypes.data, alias.ctypes.data, 3, None) == _lib.GVK_EINVAL
    assert lib.gvk_set_tuning(99, 0) == _lib.GVK_EINVAL
    assert lib.gvk_set_tuning(_lib.TUNE_LANES_PER_PAIR, 7) == _lib.GVK_EINVAL
    # argument validation happens before any launch, so these are safe without a GPU
    opt = _lib.Optimizer(0, 0.025, 0.005, 0, 0, 0)
    tables = _lib.Tables()
    neg = _lib.NegativeSource()
    assert lib.gvk_train(None, 100, C.byref(opt), C.byref(tables), None, C.byref(neg), 0, None, 1, 1, 5.0) \
        == _lib.GVK_EDIM
    assert lib.gvk_train(None, 128, C.byref(opt), C.byref(tables), None, C.byref(neg), 0, None, 1, 1, 5.0) \
        == _lib.GVK_EINVAL
    assert lib.gvk_train(None, 128, C.byref(opt), C.byref(tables), None, C.byref(neg), 0, None, 0, 1, 5.0) \
        == _lib.GVK_OK  # empty batch
    assert lib.gvk_predict(None, 7, None, None, None, None, 1) == _lib.GVK_EDIM
    with pytest.raises(ValueError):
        _lib.check(_lib.GVK_EINVAL, "x")


@pytest.mark.parametrize("n", [1, 2, 3, 17, 1000, 50000])
def test_alias_build_bit_exact(oracle, n):
    rng = np.random.default_rng(n)
    w = (rng.pareto(1.1, n) + 1e-3).astype(np.float32)
    for index_bytes in (4, 8):
        prob, alias, packed = K.alias_build(w, index_bytes)
        oprob, oalias = oracle.alias_build(w, index_bytes)
        assert (prob == oprob).all() and (alias == oalias).all()
        if packed is not None:
            assert (packed["prob"] == prob).all() and (packed["alias"] == alias).all()


def test_alias_build_degenerate_weights(oracle):
    for w in (np.zeros(3, np.float32) + 1e-30, np.array([0, 0, 1], np.float32), np.array([5], np.float32)):
        prob, alias, _ = K.alias_build(w)
        oprob, oalias = oracle.alias_build(w)
        assert np.array_equal(prob, oprob, equal_nan=True) and (alias == oalias).all()


def test_host_uniform_stream_bit_exact(oracle):
    for seed, stream, first, n in ((1, 0, 0, 1000), (2 ** 63 + 5, 77, 12345, 999), (0, 2 ** 31, 2 ** 40 + 1, 10)):
        assert (hostlib.host_uniforms(seed, stream, first, n) == oracle.host_uniforms(seed, stream, first, n)).all()


@pytest.mark.parametrize("env", [{"GVS_NO_AVX512": "1"}, {"GVS_NO_AVX512": "1", "GVS_NO_AVX2": "1"}])
def test_host_uniform_stream_same_on_every_simd_path(env):
    """The Philox block generator has AVX-512, AVX2 and scalar forms chosen at run time; all yield the same doubles."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np, sys; from graphvite_amd import hostlib; "
            "sys.stdout.buffer.write(hostlib.host_uniforms(2**63 + 5, 77, 12345, 5001).tobytes())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, check=True,
                         env=dict(os.environ, **env)).stdout
    assert np.frombuffer(out, np.float64).tobytes() == hostlib.host_uniforms(2 ** 63 + 5, 77, 12345, 5001).tobytes()


@pytest.mark.parametrize("P", [1, 2, 3, 8])
def test_partition_bit_exact(oracle, P):
    rng = np.random.default_rng(P)
    w = np.floor(rng.pareto(1.5, 4321)).astype(np.float32)  # many ties
    part, local, sizes = hostlib.partition(w, P)
    opart, olocal, osizes = oracle.partition(w, P)
    assert (part == opart).all() and (local == olocal).all() and (sizes == osizes).all()


def test_schedule_bit_exact(oracle):
    for P, W in ((1, 1), (1, 4), (2, 2), (4, 4), (8, 8), (4, 2), (8, 4), (16, 8), (3, 3)):
        assert hostlib.schedule(P, W).tolist() == oracle.schedule(P, W).tolist()
    with pytest.raises(ValueError):
        hostlib.schedule(6, 4)


# ---- graph store ------------------------------------------------------------------------------------------

def python_graph(lines, as_undirected, normalization):
    """Pure-Python restatement of Graph::add_edge / normalize / flatten (graph.cuh:103-153, core/graph.h:87-101)."""
    name2id, adj, vw = {}, [], []
    for u, v, w in lines:
        for name in (u, v):
            if name not in name2id:
                name2id[name] = len(name2id)
                adj.append([])
                vw.append(np.float32(0))
        ui, vi = name2id[u], name2id[v]
        adj[ui].append([vi, np.float32(w)])
        vw[ui] = np.float32(vw[ui] + np.float32(w))
        if as_undirected and ui != vi:
            adj[vi].append([ui, np.float32(w)])
            vw[vi] = np.float32(vw[vi] + np.float32(w))
    if normalization:
        cw = [np.float32(0)] * len(adj)
        for u in range(len(adj)):
            for v, w in adj[u]:
                cw[v] = np.float32(cw[v] + w)
        for u in range(len(adj)):
            total = np.float32(0)
            for e in adj[u]:
                e[1] = np.float32(e[1] / np.sqrt(np.float32(vw[u] * cw[e[0]])))
                total = np.float32(total + e[1])
            vw[u] = total
    edges = [(u, v) for u in range(len(adj)) for v, _ in adj[u]]
    weights = [w for u in range(len(adj)) for _, w in adj[u]]
    offsets = np.cumsum([0] + [len(a) for a in adj])
    return name2id, np.array(edges, np.uint32).reshape(-1, 2), np.array(weights, np.float32), offsets, \
        np.array(vw, np.float32)


@pytest.mark.parametrize("as_undirected", [True, False])
@pytest.mark.parametrize("normalization", [False, True])
def test_graph_load_matches_reference_semantics(tmp_path, as_undirected, normalization):
    rng = np.random.default_rng(0)
    lines = [("n%d" % a, "n%d" % b, round(float(w), 3)) for a, b, w in
             zip(rng.integers(0, 30, 200), rng.integers(0, 30, 200), rng.uniform(0.5, 3, 200))]
    lines += [("n1", "n1", 2.0), ("n2", "n3", 1.0), ("n2", "n3", 1.0)]  # self loop, duplicate edge
    name2id, edges, weights, offsets, vw = python_graph(lines, as_undirected, normalization)

    g = gv.graph.Graph()
    g.load([(u, v, w) for u, v, w in lines], as_undirected=as_undirected, normalization=normalization)
    assert g.num_vertex == len(name2id) and g.num_edge == len(lines)
    assert g.as_undirected == as_undirected and g.normalization == normalization
    assert dict(g.name2id) == name2id and list(g.id2name) == sorted(name2id, key=name2id.get)
    assert (g.edges == edges).all() and (g.flat_offsets == offsets).all()
    np.testing.assert_array_equal(g.edge_weights, weights)
    np.testing.assert_array_equal(g.vertex_weights, vw)

    # the same graph through a text file with comments, blank lines and mixed delimiters
    path = tmp_path / "graph.txt"
    with open(path, "w") as f:
        f.write("# header comment\n\n")
        for i, (u, v, w) in enumerate(lines):
            f.write("%s\t%s %s%s\n" % (u, v, w, "  # trailing" if i % 7 == 0 else ""))
    g2 = gv.graph.Graph()
    g2.load(str(path), as_undirected=as_undirected, normalization=normalization)
    assert (g2.edges == edges).all() and dict(g2.name2id) == name2id
    np.testing.assert_array_equal(g2.edge_weights, weights)

    # save -> load round trip keeps the directed structure
    out = tmp_path / "saved.txt"
    g2.save(str(out))
    g3 = gv.graph.Graph()
    g3.load(str(out), as_undirected=False)
    assert g3.num_vertex == g2.num_vertex and g3.num_directed_edge == g2.num_directed_edge


def test_graph_load_errors_and_edge_cases(tmp_path):
    g = gv.graph.Graph()
    with pytest.raises(ValueError):
        g.load("/nonexistent/file.txt")
    bad = tmp_path / "bad.txt"
    bad.write_text("a b 1 extra\n")
    with pytest.raises(ValueError, match="Invalid format at line 1"):
        g.load(str(bad))
    bad.write_text("a b\nlonely\n")
    with pytest.raises(ValueError, match="line 2"):
        g.load(str(bad))
    with pytest.raises(AttributeError):
        gv.graph.Graph(gv.dtype.uint64)
    # unweighted pairs default to weight 1; integer arrays name nodes by their decimal label
    g.load(np.array([[7, 3], [3, 9], [7, 9]], np.uint32))
    assert g.num_vertex == 3 and g.name2id["7"] == 0 and g.id2name[2] == "9" and "5" not in g.name2id
    assert g.edge_weights.tolist() == [1.0] * 6
    with pytest.raises(KeyError):
        g.name2id["nope"]


# ---- samplers ---------------------------------------------------------------------------------------------

def small_graph(seed=1, n=400, e=3000, weighted=True, as_undirected=True):
    edges = synthetic.power_law_edges(n, e, seed=seed)
    g = gv.graph.Graph()
    if weighted:
        w = np.random.default_rng(seed).uniform(0.5, 2, e).astype(np.float32)
        g.load([(str(a), str(b), float(c)) for (a, b), c in zip(edges, w)], as_undirected=as_undirected)
    else:
        g.load(edges, as_undirected=as_undirected)
    return g


@pytest.mark.parametrize("P,T", [(1, 1), (1, 3), (2, 2), (3, 4)])
def test_edge_sampler_bit_exact(oracle, P, T):
    g = small_graph()
    part, local, _ = hostlib.partition(g.vertex_weights, P)
    s = hostlib.Sampler(g, part, local, P, seed=11)
    pool_size = 2500
    pools = {(h, t): np.zeros(pool_size * 2, np.uint32) for h in range(P) for t in range(P)}
    for episode in range(2):  # stream positions carry over between fills
        start = [s.stream_position(t) for t in range(T)]
        s.fill(pools, pool_size, "edge", T, sample_batch_size=300)
        want = [np.zeros(pool_size * 2, np.uint32) for _ in range(P * P)]
        work = (pool_size + T - 1) // T
        for t in range(T):
            rnd = oracle.host_uniforms(11, t, start[t], 400000)
            used = oracle.sample_edges(g.edges, s.edge_prob, s.edge_alias, part, local, P, want, work * t,
                                       min(work * (t + 1), pool_size), 300, rnd)
            assert start[t] + used == s.stream_position(t)
        for h in range(P):
            for t in range(P):
                assert (pools[(h, t)] == want[h * P + t]).all()


@pytest.mark.parametrize("mode,as_undirected", [("walk", True), ("biased_walk", True), ("walk", False),
                                                ("biased_walk", False)])
def test_walk_samplers_bit_exact(oracle, mode, as_undirected):
    # the directed graph has nodes without out-edges: walks stop there (graph.cuh:346-349,421-424)
    g = small_graph(seed=3, as_undirected=as_undirected)
    if not as_undirected:
        assert (np.diff(g.flat_offsets.astype(np.int64)) == 0).any()
    P, T, pool_size, L, nb, aug = 2, 3, 3000, 12, 9, 4
    sb = 1 if mode == "biased_walk" else 4
    part, local, _ = hostlib.partition(g.vertex_weights, P)
    s = hostlib.Sampler(g, part, local, P, seed=5)
    s.prepare(mode, p=0.25, q=2.0, num_thread=3)
    D = g.num_directed_edge
    fo = g.flat_offsets
    if mode == "biased_walk":
        eo = s.edge_edge_offsets
        nbp, nba = s.neighbor_tables(int(eo[-1]))
        for e in range(0, D, 41):  # table contents: node2vec weights (graph.cuh:656-677) through alias build
            if eo[e + 1] == eo[e]:
                continue  # the edge ends in a node without out-edges: no table
            prob, alias = oracle.alias_build(oracle.edge_edge_weights(g.edges, g.edge_weights, fo, e, 0.25, 2.0))
            assert (prob == nbp[eo[e]:eo[e + 1]]).all() and (alias == nba[eo[e]:eo[e + 1]]).all()
    else:
        eo = None
        nbp, nba = s.neighbor_tables(D)
        for u in range(0, g.num_vertex, 13):
            if fo[u + 1] > fo[u]:
                prob, alias = oracle.alias_build(g.edge_weights[fo[u]:fo[u + 1]])
                assert (prob == nbp[fo[u]:fo[u + 1]]).all() and (alias == nba[fo[u]:fo[u + 1]]).all()
    pools = {(h, t): np.zeros(pool_size * 2, np.uint32) for h in range(P) for t in range(P)}
    s.fill(pools, pool_size, mode, T, walk_length=L, walk_batch=nb, augmentation_step=aug, shuffle_base=sb)
    want = [np.zeros(pool_size * 2, np.uint32) for _ in range(P * P)]
    work = (pool_size + T - 1) // T
    for t in range(T):
        rnd = oracle.host_uniforms(5, t, 0, 1500000)
        used = oracle.sample_walks(mode == "biased_walk", g.edges, s.edge_prob, s.edge_alias, fo, nbp, nba, eo, part,
                                   local, P, want, pool_size, work * t, min(work * (t + 1), pool_size), L, nb, aug, sb,
                                   rnd)
        assert used == s.stream_position(t)
    for h in range(P):
        for t in range(P):
            assert (pools[(h, t)] == want[h * P + t]).all()
    # one GPU's column only (tail_partition filter) reproduces exactly that column
    s2 = hostlib.Sampler(g, part, local, P, seed=5)
    s2.prepare(mode, p=0.25, q=2.0, num_thread=2)
    column = {(h, 1): np.zeros(pool_size * 2, np.uint32) for h in range(P)}
    s2.fill(column, pool_size, mode, T, walk_length=L, walk_batch=nb, augmentation_step=aug, shuffle_base=sb,
            tail_partition=1)
    for h in range(P):
        assert (column[(h, 1)] == want[h * P + 1]).all()


def test_node2vec_rejection_sampler(oracle):
    """GVS_MODE_BIASED_REJECT: bit-exact against the oracle, and the same transition distribution as the reference's
    per-edge tables (weights w/p, w, w/q of graph.cuh:664-669) without their sum-of-deg^2 memory."""
    g = small_graph(seed=5, n=80, e=700)
    P, T, pool_size, L, nb, aug, p, q = 1, 2, 3 * 40000, 2, 50, 2, 0.25, 4.0
    part, local, _ = hostlib.partition(g.vertex_weights, P)
    s = hostlib.Sampler(g, part, local, P, seed=8)
    s.prepare("biased_reject", p=p, q=q, num_thread=2)
    D, fo, E = g.num_directed_edge, g.flat_offsets, g.edges
    nbp, nba = s.neighbor_tables(D)
    sorted_nb = np.ascontiguousarray(E[np.lexsort((E[:, 1], E[:, 0])), 1])
    pools = {(0, 0): np.zeros(pool_size * 2, np.uint32)}
    s.fill(pools, pool_size, "biased_reject", T, walk_length=L, walk_batch=nb, augmentation_step=aug, shuffle_base=1)
    want = [np.zeros(pool_size * 2, np.uint32)]
    work = (pool_size + T - 1) // T
    for t in range(T):
        rnd = oracle.host_uniforms(8, t, 0, 3000000)
        used = oracle.sample_walks(2, E, s.edge_prob, s.edge_alias, fo, nbp, nba, None, part, local, P, want, pool_size,
                                   work * t, min(work * (t + 1), pool_size), L, nb, aug, 1, rnd, sorted_nb=sorted_nb,
                                   p=p, q=q)
        assert used == s.stream_position(t)
    assert (pools[(0, 0)] == want[0]).all()
    # walks of length 2 emit (c0,c1) (c0,c2) (c1,c2): transition frequencies from the most frequent (u -> v)
    inv = np.argsort(local)
    rec = pools[(0, 0)].reshape(-1, 3, 2)
    c0, c1, c2 = inv[rec[:, 0, 1]], inv[rec[:, 0, 0]], inv[rec[:, 1, 0]]
    ok = (inv[rec[:, 1, 1]] == c0) & (inv[rec[:, 2, 1]] == c1) & (inv[rec[:, 2, 0]] == c2)
    assert ok.mean() > 0.99  # thread-slice boundaries may cut a walk
    key, counts = np.unique(c0[ok].astype(np.int64) << 32 | c1[ok], return_counts=True)
    u, v = int(key[np.argmax(counts)] >> 32), int(key[np.argmax(counts)] & 0xffffffff)
    adj = {}
    for (a, b), x in zip(E.tolist(), g.edge_weights.tolist()):
        adj.setdefault(a, {})
        adj[a][b] = adj[a].get(b, 0) + x
    want_p = {x: wt * (1 / p if x == u else (1.0 if u in adj.get(x, {}) else 1 / q)) for x, wt in adj[v].items()}
    total = sum(want_p.values())
    sel = ok & (c0 == u) & (c1 == v)
    got = np.bincount(c2[sel], minlength=g.num_vertex) / sel.sum()
    for x, wt in want_p.items():
        assert abs(got[x] - wt / total) < 4 * np.sqrt(wt / total / sel.sum()) + 0.01


def test_edge_sampler_column_mode_bit_exact(oracle):
    """One GPU's column: EDGE mode draws from an alias table over exactly the edges that end in that partition."""
    g = small_graph(seed=9)
    P, T, pool_size, r = 3, 2, 1500, 1
    part, local, _ = hostlib.partition(g.vertex_weights, P)
    s = hostlib.Sampler(g, part, local, P, seed=21)
    column = {(h, r): np.zeros(pool_size * 2, np.uint32) for h in range(P)}
    s.fill(column, pool_size, "edge", T, sample_batch_size=200, tail_partition=r)
    ids, prob, alias = s.column(r)
    want_ids = np.nonzero(part[g.edges[:, 1]] == r)[0]
    assert (ids == want_ids).all()
    oprob, oalias = oracle.alias_build(g.edge_weights[want_ids], 8)
    assert (prob == oprob).all() and (alias == oalias).all()
    want = [np.zeros(pool_size * 2, np.uint32) for _ in range(P * P)]
    work = (pool_size + T - 1) // T
    for t in range(T):
        rnd = oracle.host_uniforms(21, t, 0, 200000)
        used = oracle.sample_edges(g.edges[want_ids], oprob, oalias, part, local, P, want, work * t,
                                   min(work * (t + 1), pool_size), 200, rnd, tail_filter=r)
        assert used == s.stream_position(t)
    for h in range(P):
        assert (column[(h, r)] == want[h * P + r]).all()
    # nothing is dropped: every draw lands in the column, so exactly 2 uniforms per pool slot (+ round slack)
    assert sum(s.stream_position(t) for t in range(T)) < 2 * (P * pool_size) * 3


def test_edge_sampler_distribution():
    """Positive pairs follow the edge weights; ids land in the right block with the right local ids."""
    g = small_graph(seed=7, n=60, e=400)
    P = 2
    part, local, _ = hostlib.partition(g.vertex_weights, P)
    s = hostlib.Sampler(g, part, local, P, seed=1)
    pool_size = 200000
    pools = {(h, t): np.zeros(pool_size * 2, np.uint32) for h in range(P) for t in range(P)}
    s.fill(pools, pool_size, "edge", 4, sample_batch_size=4000)
    inv = {(int(part[v]), int(local[v])): v for v in range(g.num_vertex)}
    w = {}
    for (u, v), x in zip(g.edges.tolist(), g.edge_weights.tolist()):
        w[(u, v)] = w.get((u, v), 0) + x
    for (hp, tp), pool in pools.items():
        rec = pool.reshape(-1, 2)
        heads = np.array([inv[(hp, int(h))] for h in rec[:2000, 1]])
        tails = np.array([inv[(tp, int(t))] for t in rec[:2000, 0]])
        assert all((int(h), int(t)) in w for h, t in zip(heads, tails))  # every record is a real directed edge
    # block (0, 1): empirical frequency of the heaviest edges ~ weight share inside the block
    rec = pools[(0, 1)].reshape(-1, 2)
    keys, counts = np.unique(rec[:, 1].astype(np.int64) << 32 | rec[:, 0], return_counts=True)
    block_w = {(u, v): x for (u, v), x in w.items() if part[u] == 0 and part[v] == 1}
    total = sum(block_w.values())
    for key, cnt in sorted(zip(keys, counts), key=lambda kc: -kc[1])[:10]:
        u, v = inv[(0, int(key >> 32))], inv[(1, int(key & 0xffffffff))]
        expect = block_w[(u, v)] / total
        assert abs(cnt / pool_size - expect) < 5 * np.sqrt(expect / pool_size) + 1e-4


def test_sampler_argument_errors():
    g = small_graph(seed=2, n=50, e=200)
    part, local, _ = hostlib.partition(g.vertex_weights, 1)
    s = hostlib.Sampler(g, part, local, 1, seed=0)
    pools = {(0, 0): np.zeros(2000, np.uint32)}
    with pytest.raises(ValueError, match="prepare"):
        s.fill(pools, 1000, "walk", 1, augmentation_step=2, shuffle_base=1)
    s.prepare("walk", num_thread=2)
    with pytest.raises(ValueError, match="shuffle"):
        s.fill(pools, 1000, "walk", 1, augmentation_step=2, shuffle_base=3)
    with pytest.raises(ValueError, match="augmentation_step"):
        s.fill(pools, 1000, "walk", 1, walk_length=3, augmentation_step=5, shuffle_base=1)
    with pytest.raises(ValueError):
        s.fill({(0, 0): np.zeros(10, np.uint32)}, 1000, "edge", 1)
    with pytest.raises(ValueError):
        hostlib.Sampler(g, part[:-1], local[:-1], 1, seed=0)


def test_samplers_bit_exact_with_thin_tables():
    """Below 2^27 table entries the samplers read 'fat' slots (draw outcome inline); above, the plain tables.  The
    sampler tests above ran the fat form; run them again with the thin form forced."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", __file__, "-k",
                          "edge_sampler_bit_exact or walk_samplers_bit_exact or node2vec_rejection or column_mode"],
                         cwd=root, capture_output=True, text=True, env=dict(os.environ, GVS_FAT_SLOT_LIMIT="0"))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]


def test_graph_neighbor_tables_match_the_sampler_tables(oracle):
    """gvs_graph_neighbor_tables (what the device walk sampler uploads) = the walk sampler's per-vertex tables."""
    from graphvite_amd import _lib
    g = small_graph(seed=7, as_undirected=False)
    D = g.num_directed_edge
    packed = np.zeros(D, np.dtype([("prob", np.float32), ("alias", np.uint32)]))
    _lib.check(_lib.lib().gvs_graph_neighbor_tables(g._handle, 3, packed.ctypes.data), "gvs_graph_neighbor_tables")
    part, local, _ = hostlib.partition(g.vertex_weights, 1)
    s = hostlib.Sampler(g, part, local, 1, seed=1)
    s.prepare("walk", num_thread=2)
    prob, alias = s.neighbor_tables(D)
    assert (packed["prob"] == prob).all() and (packed["alias"] == alias).all()
    fo = g.flat_offsets
    for u in range(0, g.num_vertex, 17):
        if fo[u + 1] > fo[u]:
            oprob, oalias = oracle.alias_build(g.edge_weights[fo[u]:fo[u + 1]], 4)
            assert (oprob == prob[fo[u]:fo[u + 1]]).all() and (oalias == alias[fo[u]:fo[u + 1]]).all()
    assert _lib.lib().gvs_graph_neighbor_tables(g._handle, 1, None) != 0


@pytest.mark.parametrize("seed", range(6))
def test_alias_table_reconstructs_the_distribution(seed):
    """Size-independent property of Vose's construction: slot i keeps prob_i of its 1/n mass and hands the rest to
    alias_i, so the masses must add back up to w / sum(w) (float32 construction: 1e-5 absolute on n * p)."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 5000))
    w = (rng.pareto(1.1, n) + (rng.random(n) < 0.1) * 50).astype(np.float32)
    w[rng.random(n) < 0.05] = 0  # zero-weight entries must never be drawn
    if not w.any():
        w[0] = 1
    prob, alias, _ = K.alias_build(w)
    mass = prob.astype(np.float64).copy()
    np.add.at(mass, alias.astype(np.int64), 1.0 - prob.astype(np.float64))
    want = w.astype(np.float64) / w.astype(np.float64).sum() * n
    np.testing.assert_allclose(mass, want, atol=2e-4 * max(1.0, want.max()))
    assert (mass[w == 0] < 1e-4).all()
    assert ((prob >= 0) & (prob <= 1.0 + 1e-3)).all() and (alias < n).all()  # residual rounding may exceed 1 slightly


def _corpus_oracle(lines, window, min_count, delimiters=" \t\r\n", comment="#"):
    """Plain restatement of WordGraph::load_file_compact (include/instance/word_graph.cuh:73-181)."""
    import re
    split = re.compile("[%s]+" % re.escape(delimiters))
    sentences = []
    for line in lines:
        if comment and comment in line:
            line = line[:line.index(comment)]
        sentences.append([t for t in split.split(line) if t])
    frequency, order = {}, []
    for words in sentences:
        for word in words:
            if word not in frequency:
                order.append(word)
            frequency[word] = frequency.get(word, 0) + 1
    names = [word for word in order if frequency[word] >= min_count]
    ids = {word: i for i, word in enumerate(names)}
    weight, vertex_weight = {}, [0.0] * len(names)
    for words in sentences:
        s = [ids[word] for word in words if word in ids]
        for i in range(len(s)):
            for j in range(1, window + 1):
                if i + j >= len(s):
                    break
                u, v = s[i], s[i + j]
                weight[(u, v)] = weight.get((u, v), 0) + 1
                weight[(v, u)] = weight.get((v, u), 0) + 1
                vertex_weight[u] += 1
                vertex_weight[v] += 1
    return names, weight, vertex_weight


@pytest.mark.parametrize("window,min_count", [(5, 1), (2, 3), (1, 2), (0, 1)])
def test_word_graph_from_corpus(tmp_path, window, min_count):
    rng = np.random.default_rng(window * 10 + min_count)
    vocab = ["w%d" % i for i in range(40)]
    lines = []
    for _ in range(60):
        n = int(rng.integers(0, 25))
        words = [vocab[int(min(rng.zipf(1.4), 40)) - 1] for _ in range(n)]
        lines.append(" ".join(words) + ("  # trailing comment w0 w0" if rng.random() < 0.2 else ""))
    lines.append("the the the the")  # a word next to itself: both directions land on the same entry
    path = tmp_path / "corpus.txt"
    path.write_text("\n".join(lines) + "\n")
    names, weight, vertex_weight = _corpus_oracle(lines, window, min_count)

    g = gv.graph.WordGraph()
    g.load(str(path), window=window, min_count=min_count)
    assert repr(g).startswith("WordGraph<uint32>") and g.as_undirected and not g.normalization
    assert g.num_vertex == len(names) and [g.id2name[i] for i in range(g.num_vertex)] == names
    assert g.num_edge == len(weight) == g.num_directed_edge  # both directions are counted (word_graph.cuh:156-160)
    got = {(int(u), int(v)): float(w) for (u, v), w in zip(g.edges.tolist(), g.edge_weights.tolist())}
    assert got == {k: float(v) for k, v in weight.items()}
    np.testing.assert_array_equal(g.vertex_weights, np.array(vertex_weight, np.float32))
    fo = g.flat_offsets
    assert (np.diff(g.edges[:, 0].astype(np.int64)) >= 0).all() and fo[-1] == len(weight)  # CSR by source

    gn = gv.graph.WordGraph()
    gn.load(str(path), window=window, min_count=min_count, normalization=True)
    if len(weight):
        out_w = np.array(vertex_weight)
        in_w = np.zeros(len(names))
        for (u, v), w in weight.items():
            in_w[v] += w
        want = {k: w / np.sqrt(out_w[k[0]] * in_w[k[1]]) for k, w in weight.items()}  # graph.cuh:103-121
        gotn = {(int(u), int(v)): float(w) for (u, v), w in zip(gn.edges.tolist(), gn.edge_weights.tolist())}
        assert gotn.keys() == want.keys()
        for k in want:
            assert gotn[k] == pytest.approx(want[k], rel=1e-5)
    with pytest.raises(ValueError):
        g.load(str(tmp_path / "missing.txt"))
    # a word graph trains through the same solver (WordGraphApplication)
    assert isinstance(gv.application.Application("word graph", dim=32), gv.application.WordGraphApplication)
    with pytest.raises(ValueError):
        gv.application.Application("knowledge graph", dim=32)
