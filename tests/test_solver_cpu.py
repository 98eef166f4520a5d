"""GraphSolver / GraphApplication host logic without a GPU: the kernels are replaced — explicitly, in the test —
by the oracle (tests/fake_kernels.py), so that partitioning, sampling, pool handling, batch-id / learning-rate
accounting, the multi-process exchange (gloo, world_size 2) and write-back can be checked here.  The product
itself has no CPU path: constructing a GraphSolver without a GPU raises."""
import os
import pickle
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import graphvite_amd as gv
from fake_kernels import OracleKernels
from graphvite_amd import synthetic
from oracle_lib import link_prediction_auc


def make_graph(n=300, e=3000, seed=1):
    g = gv.graph.Graph()
    g.load(synthetic.power_law_edges(n, e, seed=seed))
    return g


def test_no_cpu_training_path():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="No GPU"):
        gv.solver.GraphSolver(128)
    with pytest.raises(AttributeError):
        gv.solver.GraphSolver(100, kernels=OracleKernels())
    with pytest.raises(AttributeError):
        gv.solver.GraphSolver(128, float_type=gv.float64, kernels=OracleKernels())


def test_build_defaults_follow_the_reference():
    g = make_graph()
    s = gv.solver.GraphSolver(128, kernels=OracleKernels(), num_sampler_per_worker=2)
    s.build(g)
    assert s.optimizer.type == "SGD" and s.optimizer.init_lr == 0.025 and s.optimizer.weight_decay == 0.005
    assert s.optimizer.schedule.type == "linear"
    assert s.num_partition == 1 and s.num_negative == 1 and s.batch_size == 100000
    assert s.episode_size == 200  # max(300 * 175 / 1 / 1e5, 1) -> 1, single partition -> 2e7 / 1e5 (solver.h:426-436)
    assert s.vertex_embeddings.shape == (300 if g.num_vertex == 300 else g.num_vertex, 128)
    s.build(g, optimizer=0.1, batch_size=500, episode_size=3)  # bare learning rate keeps the default optimizer
    assert s.optimizer.type == "SGD" and s.optimizer.init_lr == pytest.approx(0.1)
    s.build(g, optimizer=gv.optimizer.Adam(1e-3), batch_size=500, episode_size=3)
    assert s.optimizer.type == "Adam" and s.num_moment == 2
    with pytest.raises(ValueError):
        s.build(g, num_partition=0 - 1)
    with pytest.raises(ValueError):
        s.train(model="TransE")


def test_train_accounting_and_determinism():
    g = make_graph()
    runs = []
    for _ in range(2):
        k = OracleKernels()
        s = gv.solver.GraphSolver(64, kernels=k, num_sampler_per_worker=2, seed=3)
        s.build(g, batch_size=500, episode_size=6)
        views = (s.vertex_embeddings, s.context_embeddings)
        s.train("LINE", num_epoch=5, log_frequency=7)
        assert s.vertex_embeddings is views[0] and s.context_embeddings is views[1]  # stable host buffers
        runs.append((s.vertex_embeddings.copy(), s.context_embeddings.copy(), list(k.launches), s))
    v0, c0, launches, s = runs[0]
    assert (v0 == runs[1][0]).all() and (c0 == runs[1][1]).all()  # same seed -> same pools, negatives, result
    # num_batch = num_epoch * |E| / batch (solver.h:611), overshoot to whole episodes (solver.h:629)
    assert s.num_batch == 5 * 3000 // 500 and s.augmentation_step == 3 and s.shuffle_base == 3
    ids = [b for b, _ in launches]
    assert ids == list(range(len(ids))) and len(ids) == 30 and s.batch_id == 30
    for b, lr in launches:  # lr = init_lr * max(1 - b / num_batch, 1e-4) (optimizer.h:77-79)
        assert lr == pytest.approx(0.025 * max(1 - b / 30.0, 1e-4), rel=1e-6)
    assert np.abs(c0).max() > 0 and np.isfinite(v0).all()
    # resume continues the batch counter and does not re-initialise
    before = s.vertex_embeddings.copy()
    s.train("LINE", num_epoch=1, resume=True, log_frequency=1000)
    assert s.num_batch == 30 + 6 and s.batch_id == 36 and not (s.vertex_embeddings == before).all()


def test_grouped_pair_order_permutes_inside_batches_only():
    """pair_order="grouped": every batch trains the same multiset of pairs as with "sampled", heads adjacent."""
    g = make_graph()

    class Recording(OracleKernels):
        def __init__(self):
            OracleKernels.__init__(self)
            self.batches = []

        def train(self, vertex, context, pairs, *args, **kwargs):
            self.batches.append(self._np(pairs).copy())
            return OracleKernels.train(self, vertex, context, pairs, *args, **kwargs)

    seen = {}
    for order in ("sampled", "grouped"):
        k = Recording()
        s = gv.solver.GraphSolver(32, kernels=k, num_sampler_per_worker=2, seed=3, pair_order=order)
        s.build(g, batch_size=500, episode_size=3)
        s.train("LINE", num_epoch=2, augmentation_step=1, log_frequency=1 << 30)
        seen[order] = k.batches
    assert len(seen["sampled"]) == len(seen["grouped"]) > 0
    for a, b in zip(seen["sampled"], seen["grouped"]):
        a, b = a.reshape(-1, 2), b.reshape(-1, 2)
        assert (np.diff(b[:, 1].astype(np.int64)) >= 0).all() and not (a == b).all()
        key = lambda x: np.sort(x[:, 1].astype(np.int64) << 32 | x[:, 0].astype(np.int64))
        assert (key(a) == key(b)).all()
    with pytest.raises(ValueError):
        gv.solver.GraphSolver(32, kernels=OracleKernels(), pair_order="sorted")


@pytest.mark.parametrize("model,aug", [("LINE", 1), ("DeepWalk", 3), ("node2vec", 2)])
def test_models_and_samplers_run(model, aug):
    g = make_graph(200, 1500, seed=2)
    k = OracleKernels()
    s = gv.solver.GraphSolver(32, kernels=k, num_sampler_per_worker=2)
    s.build(g, batch_size=300, episode_size=4, num_negative=2)
    s.train(model, num_epoch=2, augmentation_step=aug, random_walk_length=8, random_walk_batch_size=5, p=0.5, q=2.0,
            positive_reuse=2)
    assert s.batch_id == 2 * 1500 // 300 - (2 * 1500 // 300) % 8 + 8 or s.batch_id % 8 == 0  # episodes of 4 x reuse 2
    assert np.abs(s.context_embeddings).max() > 0
    if model != "LINE":
        assert s.shuffle_base == 1


def test_device_sampling_mode_host_logic(oracle):
    """Edge positives drawn by gvk_sample_pairs (oracle-backed here): block tables cover exactly the block's edges,
    every pair trained on is a real edge of the right block, training learns, walk models are refused."""
    g = make_graph(200, 2000, seed=8)
    k = OracleKernels()
    s = gv.solver.GraphSolver(32, kernels=k, num_sampler_per_worker=1, device_sampling=True, seed=4)
    s.build(g, batch_size=400, episode_size=3)
    s.train("LINE", num_epoch=3, augmentation_step=1)
    assert s.batch_id == 15 and np.abs(s.context_embeddings).max() > 0
    drawn = s._positive_index
    assert drawn >= 15 * 400
    s.train("LINE", num_epoch=3, augmentation_step=1, resume=True)  # the positive stream goes on, it does not start over
    assert s.batch_id == 30 and s._positive_index >= drawn + 15 * 400
    # random-walk models sample on the device too when there is a single partition
    for model, aug in (("DeepWalk", 3), ("node2vec", 2), ("LINE", 2)):
        s.build(g, batch_size=300, episode_size=4)
        s.train(model, num_epoch=2, augmentation_step=aug, random_walk_length=8, p=0.5, q=2.0)
        assert np.abs(s.context_embeddings).max() > 0 and s._sampler is None  # no CPU sampler was ever built
    s.build(g, batch_size=400, episode_size=3)
    # the pairs the kernel produced for an episode are edges of the graph (local ids of the single partition)
    s._configure_training("LINE", 1, False, 1, 40, 100, 0, 1, 1, 1, 0.75, 5.0, 1000)
    state = s._upload_state()
    s._upload_block_tables(state)
    table = state["block_tables"][(0, 0)]  # gvk_edge_entry per directed edge
    assert table.shape == (g.num_directed_edge, 2)
    pool = torch.zeros(2 * 5000, dtype=torch.int32)
    k.sample_edges(table, 123, 0, pool, 5000)
    inv = np.argsort(s._local)  # local id -> global id (one partition)
    rec = pool.numpy().view(np.uint32).reshape(-1, 2)
    real = set(map(tuple, g.edges.tolist()))
    assert all((int(inv[h]), int(inv[t])) in real for t, h in rec[:500])
    # uniform weights -> every directed edge equally likely: heads follow the degree distribution
    deg = np.bincount(g.edges[:, 0], minlength=g.num_vertex)
    got = np.bincount(inv[rec[:, 1]], minlength=g.num_vertex)
    top = np.argsort(-deg)[:5]
    assert np.allclose(got[top] / 5000.0, deg[top] / deg.sum(), atol=0.02)


def test_device_walk_sampler_semantics(oracle):
    """gvk_sample_walks as restated by the oracle: every emitted pair is a walk pair (distance <= augmentation_step
    along real edges), each thread fills exactly its quota, node2vec transition frequencies follow the p / q weights
    of the reference's per-edge tables (graph.cuh:656-677)."""
    from graphvite_amd import hostlib
    g = gv.graph.Graph()
    edges = synthetic.community_edges(60, 500, num_community=3, seed=1)
    w = np.random.default_rng(0).uniform(0.5, 2.0, len(edges)).astype(np.float32)
    g.load([(str(a), str(b), float(c)) for (a, b), c in zip(edges, w)])
    part, local, _ = hostlib.partition(g.vertex_weights, 1)
    s = hostlib.Sampler(g, part, local, 1, seed=0)
    s.prepare("walk", num_thread=2)
    D = g.num_directed_edge
    nb_prob, nb_alias = s.neighbor_tables(D)
    edge_prob, edge_alias = oracle.alias_build(g.edge_weights)
    flat, E = g.flat_offsets, g.edges
    order = np.lexsort((E[:, 1], E[:, 0]))
    sorted_nb = np.ascontiguousarray(E[order, 1])
    adj = {}
    for (u, v), x in zip(E.tolist(), g.edge_weights.tolist()):
        adj.setdefault(u, {})
        adj[u][v] = adj[u].get(v, 0) + x
    inv = np.argsort(local)
    L, aug, sb = 6, 2, 2
    per_walk = aug * L - aug * (aug - 1) // 2
    pool_pairs = per_walk * 4000
    pool = oracle.sample_walks_device(flat, E, edge_prob, edge_alias, np.ascontiguousarray(nb_prob),
                                      np.ascontiguousarray(nb_alias), sorted_nb, local, False, 1.0, 1.0, 5, 0,
                                      pool_pairs, L, aug, sb)
    # undo the pseudo shuffle, then thread w's pairs are offsets [w * per_walk, (w + 1) * per_walk)
    offsets = np.arange(pool_pairs)
    slots = offsets % sb * (pool_pairs // sb) + offsets // sb
    rec = pool[slots]
    heads, tails = inv[rec[:, 1]], inv[rec[:, 0]]
    first = rec.reshape(4000, per_walk, 2)
    for wlk in range(0, 4000, 97):  # pair order: (c0,c1) (c1,c2) (c0,c2) (c2,c3) (c1,c3) ...
        h, t = inv[first[wlk, :, 1]], inv[first[wlk, :, 0]]
        chain = [h[0], t[0]]
        i = 1
        while i < per_walk:
            chain.append(t[i])
            assert h[i] == chain[-2] and h[i + 1] == chain[-3] and t[i + 1] == chain[-1]
            i += 2
        assert all(chain[j + 1] in adj[chain[j]] for j in range(L))
    # first step of unbiased walks follows the out-edge weights of the start edge's head
    # node2vec: from a fixed (u -> v), next-node frequencies ~ w(v, x) * f(x)
    p, q = 0.25, 4.0
    pool = oracle.sample_walks_device(flat, E, edge_prob, edge_alias, np.ascontiguousarray(nb_prob),
                                      np.ascontiguousarray(nb_alias), sorted_nb, local, True, p, q, 9, 0,
                                      3 * 60000, 2, 2, 1)
    rec = pool.reshape(60000, 3, 2)  # (c0,c1) (c1,c2) (c0,c2)
    c0, c1, c2 = inv[rec[:, 0, 1]], inv[rec[:, 0, 0]], inv[rec[:, 1, 0]]
    key, counts = np.unique(c0.astype(np.int64) << 32 | c1, return_counts=True)
    u, v = int(key[np.argmax(counts)] >> 32), int(key[np.argmax(counts)] & 0xffffffff)
    sel = (c0 == u) & (c1 == v)
    want = {x: wt * (1 / p if x == u else (1.0 if u in adj.get(x, {}) else 1 / q)) for x, wt in adj[v].items()}
    total = sum(want.values())
    got = np.bincount(c2[sel], minlength=60) / sel.sum()
    for x, wt in want.items():
        assert abs(got[x] - wt / total) < 4 * np.sqrt(wt / total / sel.sum()) + 0.01


def test_node2vec_switches_to_rejection_past_the_table_limit():
    g = make_graph(200, 1500, seed=2)
    s = gv.solver.GraphSolver(32, kernels=OracleKernels(), num_sampler_per_worker=2)
    s.build(g, batch_size=300, episode_size=4)
    s.node2vec_table_limit = 10  # force the O(|E|)-memory sampler
    s.train("node2vec", num_epoch=2, augmentation_step=2, random_walk_length=8, random_walk_batch_size=5, p=0.5, q=2.0)
    assert s._mode == "biased_reject" and np.abs(s.context_embeddings).max() > 0


def test_schedule_interleaves_head_groups_for_overlap():
    from graphvite_amd import hostlib
    for P, W in ((4, 2), (8, 4), (16, 8), (12, 4)):
        ref = hostlib.schedule(P, W)
        new = gv.solver.GraphSolver._overlap_order(ref, P, W)
        assert sorted(map(tuple, new.reshape(-1, 2).tolist())) == sorted(map(tuple, ref.reshape(-1, 2).tolist()))
        assert len({tuple(b) for b in new.reshape(-1, 2).tolist()}) == P * P       # every block once per episode
        for step in new:
            assert len(set(step[:, 0])) == W and len(set(step[:, 1])) == W           # orthogonal within a step
            assert len({h // W for h in step[:, 0]}) == 1                             # one head group per step
        groups = [int(step[0, 0]) // W for step in new]
        assert all(a != b for a, b in zip(groups, groups[1:]))  # consecutive steps never touch the same head group
    same = hostlib.schedule(4, 4)
    assert (gv.solver.GraphSolver._overlap_order(same, 4, 4) == same).all()


def test_training_session_steps_equal_train():
    """solver.session(): the public step-by-step form of train() produces the same tables as train() itself."""
    g = make_graph(250, 2500, seed=3)
    kw = dict(model="LINE", num_epoch=2, augmentation_step=1, log_frequency=100000)
    a = gv.solver.GraphSolver(32, kernels=OracleKernels(), num_sampler_per_worker=2, seed=5)
    a.build(g, batch_size=500, episode_size=5)
    a.train(**kw)
    b = gv.solver.GraphSolver(32, kernels=OracleKernels(), num_sampler_per_worker=2, seed=5)
    b.build(g, batch_size=500, episode_size=5)
    session = b.session(**kw)
    assert session.blocks == [(0, 0)]
    while b.batch_id < b.num_batch:
        pools = session.new_host_pools()
        session.fill(pools)
        resident = session.upload(pools)
        for step, (hp, tp) in enumerate(session.blocks):
            session.train_block(hp, tp, resident[(hp, tp)])
            session.exchange(step)
    assert session.loss.numel() == 500
    session.finish()
    assert a.batch_id == b.batch_id
    assert (a.vertex_embeddings == b.vertex_embeddings).all() and (a.context_embeddings == b.context_embeddings).all()
    with pytest.raises(TypeError):
        b.session(modle="LINE")


def test_custom_schedule_and_optimizers():
    g = make_graph(150, 900, seed=4)
    k = OracleKernels()
    s = gv.solver.GraphSolver(32, kernels=k, num_sampler_per_worker=1)
    s.build(g, optimizer=gv.optimizer.SGD(0.1, 0, lambda b, n: 0.5), batch_size=300, episode_size=2)
    s.train("LINE", num_epoch=1, augmentation_step=1)
    assert all(lr == pytest.approx(0.05) for _, lr in k.launches)
    for opt in (gv.optimizer.Momentum(0.01), gv.optimizer.AdaGrad(0.01), gv.optimizer.RMSprop(0.01),
                gv.optimizer.Adam(0.01)):
        s.build(g, optimizer=opt, batch_size=300, episode_size=2)
        s.train("LINE", num_epoch=1, augmentation_step=1)
        assert np.isfinite(s.vertex_embeddings).all() and np.abs(s.context_embeddings).max() > 0


def test_predict_and_link_prediction_pipeline(tmp_path):
    edges = synthetic.power_law_edges(400, 6000, seed=5)
    train, (valid, test) = synthetic.link_prediction_split(edges, (100, 5, 5), seed=1024)
    assert len(train) + (valid[2] == 1).sum() + (test[2] == 1).sum() == len(edges)
    assert (test[2] == 1).sum() == (test[2] == 0).sum()
    app = gv.application.GraphApplication(dim=32)
    app.get_solver = lambda **kw: gv.solver.GraphSolver(32, kernels=OracleKernels(), num_sampler_per_worker=2)
    app.load(edge_list=train)
    app.build(batch_size=1000, episode_size=10)
    app.train(model="LINE", num_epoch=60, augmentation_step=1, log_frequency=100000)
    H, T, Y = test
    result = app.evaluate("link prediction", H=[str(h) for h in H], T=[str(t) for t in T], Y=Y.tolist(),
                          filter_H=[str(h) for h in train[:, 0]], filter_T=[str(t) for t in train[:, 1]])
    # the same number from the numpy restatement of the reference's AUC (application.py:433-449)
    n2i = app.graph.name2id
    in_train = {(n2i[str(h)], n2i[str(t)]) for h, t in train}
    keep = [(n2i[str(h)], n2i[str(t)], y) for h, t, y in zip(H, T, Y) if str(h) in n2i and str(t) in n2i]
    keep = [k for k in keep if (k[0], k[1]) not in in_train]  # filter_H / filter_T drop pairs seen in training
    want = link_prediction_auc(app.solver.vertex_embeddings, app.solver.context_embeddings, [k[0] for k in keep],
                               [k[1] for k in keep], [k[2] for k in keep])
    assert result["AUC"] == pytest.approx(want, abs=1e-9) and result["AUC"] > 0.6
    # predict takes (v, c) pairs in global ids and returns dot products
    pairs = np.array([[1, 2], [3, 4], [5, 5]])
    got = app.solver.predict(pairs)
    want = np.einsum("ij,ij->i", app.solver.vertex_embeddings[pairs[:, 0]], app.solver.context_embeddings[pairs[:, 1]])
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
    with pytest.raises(ValueError):
        app.solver.predict(np.zeros((3, 3), np.int64))
    # save / load round trip maps nodes by name
    path = str(tmp_path / "model.pkl")
    app.save_model(path)
    saved = pickle.load(open(path, "rb"))
    assert saved["solver"]["vertex_embeddings"].shape == app.solver.vertex_embeddings.shape
    old = app.solver.vertex_embeddings.copy()
    app.solver.vertex_embeddings[:] = 0
    app.load_model(path)
    assert (app.solver.vertex_embeddings == old).all()
    with pytest.raises(ValueError):
        app.evaluate("node clustering")
    # the file has the reference's layout (application.py:145-187): attribute access all the way down, the class the
    # reference pickles (easydict.EasyDict), and with save_hyperparameter its key set including solver.optimizer
    assert type(saved).__module__ == "easydict" and type(saved).__name__ == "EasyDict"
    assert saved.graph.name2id["%d" % train[0, 0]] == app.graph.name2id["%d" % train[0, 0]]
    assert saved.solver.vertex_embeddings is saved["solver"]["vertex_embeddings"]
    app.save_model(path, save_hyperparameter=True)
    full = pickle.load(open(path, "rb"))
    assert full.solver.optimizer.type == "SGD" and full.solver.optimizer.schedule == "linear"
    assert full.solver.optimizer.lr == pytest.approx(0.025) and full.solver.num_negative == 1
    assert full.graph.num_vertex == app.graph.num_vertex and full.solver.model == "LINE"
    assert full.solver.batch_size == 1000 and full.solver.random_walk_batch_size == 100
    # what the reference's load_model does with such a file (application.py:131-142, 288-291): attribute access only
    mapping = [full.graph.name2id[name] for name in app.graph.id2name]
    assert (full.solver.vertex_embeddings[mapping] == old).all()
    # and a file the way the reference writes it — object attributes gathered into nested EasyDicts, the name map an
    # EasyDict too — or as older versions of this package wrote it (plain dicts) loads here
    from graphvite_amd.application.application import easy_dict_class
    EasyDict = easy_dict_class()
    theirs = EasyDict()
    theirs.graph = EasyDict()
    theirs.graph["name2id"] = dict(app.graph.name2id)
    theirs.graph["id2name"] = list(app.graph.id2name)
    theirs.solver = EasyDict(vertex_embeddings=old * 2, context_embeddings=np.array(app.solver.context_embeddings))
    for record in (theirs, {"graph": dict(theirs.graph), "solver": dict(theirs.solver)}):
        with open(path, "wb") as fout:
            pickle.dump(record, fout, protocol=pickle.HIGHEST_PROTOCOL)
        app.solver.vertex_embeddings[:] = 0
        app.load_model(path)
        assert (app.solver.vertex_embeddings == old * 2).all()


def test_node_classification_cli_and_embedding_file(tmp_path):
    """The "next" rows around the path: node classification, `run config.yaml`, word2vec-format embeddings."""
    import yaml
    from graphvite_amd import cmd
    edges = synthetic.community_edges(300, 6000, num_community=3, seed=2)
    graph_file = tmp_path / "graph.txt"
    np.savetxt(graph_file, edges, fmt="%d")
    label_file = tmp_path / "label.txt"
    with open(label_file, "w") as f:
        for i in range(300):
            f.write("%d\tc%d\n" % (i, i // 100))
    config = {"application": "graph", "resource": {"dim": 32}, "format": {"delimiters": " \t\r\n", "comment": "#"},
              "graph": {"file_name": str(graph_file), "as_undirected": True},
              "build": {"optimizer": {"type": "SGD", "lr": 0.025, "weight_decay": 0.005}, "num_partition": "auto",
                        "num_negative": 1, "batch_size": 1000, "episode_size": 10},
              "train": {"model": "LINE", "num_epoch": 150, "augmentation_step": 1, "log_frequency": 100000},
              "evaluate": [{"task": "node classification", "file_name": str(label_file), "portions": [0.2],
                            "times": 1}],
              "save": {"file_name": str(tmp_path / "model.pkl")}}
    config_file = tmp_path / "config.yaml"
    config_file.write_text(yaml.safe_dump(config))
    real = gv.application.GraphApplication.get_solver
    gv.application.GraphApplication.get_solver = lambda self, **kw: gv.solver.GraphSolver(
        self.dim, kernels=OracleKernels(), num_sampler_per_worker=2)
    try:
        app = cmd.run_main(cmd.main.__globals__["argparse"].Namespace(config=str(config_file), gpu=None, cpu=None,
                                                                      eval=True))
    finally:
        gv.application.GraphApplication.get_solver = real
    assert app.solver.num_partition == 1 and app.solver.optimizer.type == "SGD" and (tmp_path / "model.pkl").exists()
    result = app.node_classification(file_name=str(label_file), portions=(0.2,), times=2)
    assert result["micro-F1@20%"] > 0.9 and result["macro-F1@20%"] > 0.9  # three planted communities
    # word2vec-style embedding file
    out = tmp_path / "emb.bin"
    app.solver.save_embeddings(str(out))
    data = open(out, "rb").read()
    header, rest = data.split(b"\n", 1)
    assert header == b"300 32"
    name0 = app.graph.id2name[0].encode()
    assert rest.startswith(name0 + b" ")
    first = np.frombuffer(rest[len(name0) + 1:len(name0) + 1 + 32 * 4], np.float32)
    assert (first == app.solver.vertex_embeddings[0]).all()
    with pytest.raises(ValueError):
        cmd.load_config.__call__  # placeholder datasets are rejected
        bad = tmp_path / "bad.yaml"
        bad.write_text("graph:\n  file_name: <blogcatalog.train>\n")
        cmd.load_config(str(bad))


# ---- world_size 2 over gloo -----------------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, model, aug, num_partition=0, pair_order="sampled", device_sampling=False):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import logging
        gv.init_logging(logging.ERROR)
        g = make_graph(240, 2400, seed=6)
        k = OracleKernels()
        s = gv.solver.GraphSolver(32, kernels=k, num_sampler_per_worker=2, seed=9, pair_order=pair_order,
                                  device_sampling=device_sampling)
        s.build(g, batch_size=400, episode_size=3, num_partition=num_partition)
        assert s.num_worker == world and s.num_partition == (num_partition or world)
        # every pair a worker trains on must be a real (walk) pair of the graph that lives in the block being trained
        E = g.edges
        nbrs = [set() for _ in range(g.num_vertex)]
        for u, v in E.tolist():
            nbrs[u].add(v)
        inv = {(int(p), int(l)): v for v, (p, l) in enumerate(zip(s._part, s._local))}
        original = s._train_block

        def checked(state, hp, tp, pool):
            rec = pool.numpy().view(np.uint32).reshape(-1, 2)[:s.episode_size * s.batch_size]
            if pair_order == "grouped":  # every batch arrives in ascending head-row order
                assert (np.diff(rec[:, 1].astype(np.int64).reshape(-1, s.batch_size), axis=1) >= 0).all()
            rec = rec[::37]
            for t_local, h_local in rec.tolist():
                h, t = inv[(hp, h_local)], inv[(tp, t_local)]        # KeyError = a pair routed to the wrong block
                reach = nbrs[h] if aug == 1 else nbrs[h] | set().union(*[nbrs[x] for x in nbrs[h]])
                assert t in reach, "pair (%d, %d) is not within %d steps" % (h, t, aug)
            return original(state, hp, tp, pool)

        s._train_block = checked
        s.train(model, num_epoch=4, augmentation_step=aug, random_walk_length=6, random_walk_batch_size=4,
                p=0.25, q=0.25, log_frequency=100000)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), v=s.vertex_embeddings, c=s.context_embeddings,
                 ids=np.array([b for b, _ in k.launches]), lrs=np.array([lr for _, lr in k.launches]),
                 batch_id=s.batch_id, num_batch=s.num_batch, tails=np.array(s._my_tails))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model,aug", [("LINE", 1), ("DeepWalk", 2), ("node2vec", 2)])
def test_two_process_training_over_gloo(tmp_path, model, aug):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), model, aug), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % i)) for i in range(world)]
    # after write-back every process holds the same, complete tables
    assert (r[0]["v"] == r[1]["v"]).all() and (r[0]["c"] == r[1]["c"]).all()
    assert np.abs(r[0]["c"]).max() > 0 and np.isfinite(r[0]["v"]).all()
    assert r[0]["tails"].tolist() == [0] and r[1]["tails"].tolist() == [1]  # context shard pinned per worker
    # the two workers share one batch counter: ids interleave, every id exactly once, whole episodes
    ids = np.sort(np.concatenate([r[0]["ids"], r[1]["ids"]]))
    assert (ids == np.arange(len(ids))).all()
    assert (r[0]["ids"] % 2 == 0).all() and (r[1]["ids"] % 2 == 1).all()
    assert len(ids) % (2 * 2 * 3) == 0 and int(r[0]["batch_id"]) == len(ids) >= int(r[0]["num_batch"])
    for i in range(world):
        want = 0.025 * np.maximum(1 - r[i]["ids"] / float(r[i]["num_batch"]), 1e-4)
        np.testing.assert_allclose(r[i]["lrs"], want, rtol=1e-6)


@pytest.mark.parametrize("model,aug", [("LINE", 1), ("DeepWalk", 2)])
def test_more_partitions_than_workers_over_gloo(tmp_path, model, aug):
    """P = 4 partitions on 2 workers (solver.h:562-574 with x, y groups): each worker owns two context shards; the
    steps interleave the two head groups and the exchange is asynchronous; walk models route their pairs."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), model, aug, 4), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % i)) for i in range(world)]
    assert (r[0]["v"] == r[1]["v"]).all() and (r[0]["c"] == r[1]["c"]).all()
    assert r[0]["tails"].tolist() == [0, 2] and r[1]["tails"].tolist() == [1, 3]
    ids = np.sort(np.concatenate([r[0]["ids"], r[1]["ids"]]))
    assert (ids == np.arange(len(ids))).all() and len(ids) % (4 * 4 * 3) == 0  # whole episodes of P^2 blocks
    assert np.abs(r[0]["c"]).max() > 0


@pytest.mark.parametrize("model,aug", [("LINE", 1), ("DeepWalk", 2)])
def test_grouped_pair_order_over_gloo(tmp_path, model, aug):
    """pair_order="grouped" on 2 workers / 4 partitions: uploaded (LINE) and routed (walk) pools are regrouped batch by
    batch before they are trained; everything else as above."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), model, aug, 4, "grouped"), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % i)) for i in range(world)]
    assert (r[0]["v"] == r[1]["v"]).all() and (r[0]["c"] == r[1]["c"]).all()
    ids = np.sort(np.concatenate([r[0]["ids"], r[1]["ids"]]))
    assert (ids == np.arange(len(ids))).all() and np.abs(r[0]["c"]).max() > 0


def test_word_graph_application_trains_a_corpus(tmp_path):
    """WordGraphApplication (application.py:536-573): corpus -> co-occurrence graph -> the same GraphSolver path."""
    rng = np.random.default_rng(0)
    topics = [["cat", "dog", "pet", "vet", "fur"], ["gpu", "hbm", "wave", "lane", "simd"]]
    lines = [" ".join(rng.choice(topics[i % 2], 12)) for i in range(400)]
    path = tmp_path / "corpus.txt"
    path.write_text("\n".join(lines) + "\n")
    real = gv.application.GraphApplication.get_solver
    gv.application.GraphApplication.get_solver = lambda self, **kw: gv.solver.GraphSolver(
        self.dim, kernels=OracleKernels(), num_sampler_per_worker=2)
    try:
        app = gv.application.Application("word graph", dim=32)
        app.load(file_name=str(path), window=3, min_count=5)
        app.build(batch_size=200, episode_size=5)
        app.train(model="LINE", num_epoch=300, augmentation_step=1, log_frequency=1 << 30)
    finally:
        gv.application.GraphApplication.get_solver = real
    assert isinstance(app.graph, gv.graph.WordGraph) and app.graph.num_vertex == 10
    v, c = app.solver.vertex_embeddings, app.solver.context_embeddings
    score = v @ c.T
    ids = [[app.graph.name2id[w] for w in topic] for topic in topics]
    inside = np.mean([score[np.ix_(t, t)].mean() for t in ids])
    across = np.mean([score[np.ix_(ids[0], ids[1])].mean(), score[np.ix_(ids[1], ids[0])].mean()])
    assert inside > across + 0.5  # words of a topic co-occur, words of different topics never do


def test_four_workers_eight_partitions_over_gloo(tmp_path):
    """4 workers, 8 partitions (two head groups): every schedule step moves each rank's head partition into its own slot
    of the group's slab and one in-place all-gather rebuilds the group on every rank (`_claim_slot`, `_exchange`) while
    the other group trains.  After write-back all four ranks must hold the same, complete tables, every batch id exactly
    once, and every trained pair in the block it was trained in (checked inside the workers)."""
    world, port = 4, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), "LINE", 1, 8, "grouped"), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % i)) for i in range(world)]
    for other in r[1:]:
        assert (r[0]["v"] == other["v"]).all() and (r[0]["c"] == other["c"]).all()
    assert np.abs(r[0]["c"]).max() > 0 and np.isfinite(r[0]["v"]).all()
    assert [x["tails"].tolist() for x in r] == [[0, 4], [1, 5], [2, 6], [3, 7]]  # two pinned context shards per worker
    ids = np.sort(np.concatenate([x["ids"] for x in r]))
    assert (ids == np.arange(len(ids))).all() and len(ids) % (8 * 8 * 3) == 0


def test_device_sampling_over_gloo(tmp_path):
    """device_sampling=True on 2 workers / 4 partitions (LINE): every worker draws the pools of its own blocks one block
    ahead (gvk_sample_pairs from the block's alias table), regroups them, trains, exchanges."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), "LINE", 1, 4, "grouped", True), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % i)) for i in range(world)]
    assert (r[0]["v"] == r[1]["v"]).all() and (r[0]["c"] == r[1]["c"]).all()
    ids = np.sort(np.concatenate([r[0]["ids"], r[1]["ids"]]))
    assert (ids == np.arange(len(ids))).all() and len(ids) % (4 * 4 * 3) == 0 and np.abs(r[0]["c"]).max() > 0


@pytest.mark.parametrize("model,aug,partitions", [("DeepWalk", 2, 2), ("node2vec", 2, 4)])
def test_device_sampled_walks_over_gloo(tmp_path, model, aug, partitions):
    """device_sampling=True for the random-walk models on 2 workers: every worker draws its half of EVERY block's pool
    on the device (gvk_sample_walks_blocks: walks binned per (head, tail) block, restated by the stand-in), one
    all_to_all routes the halves to the worker that trains the block — no CPU sampler exists on either rank.  The
    worker-side checks make sure every trained pair is a walk pair of the block being trained."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), model, aug, partitions, "sampled", True), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % i)) for i in range(world)]
    assert (r[0]["v"] == r[1]["v"]).all() and (r[0]["c"] == r[1]["c"]).all()
    ids = np.sort(np.concatenate([r[0]["ids"], r[1]["ids"]]))
    assert (ids == np.arange(len(ids))).all() and np.abs(r[0]["c"]).max() > 0


def test_device_sampled_walks_on_several_partitions_of_one_gpu():
    """One worker, 3 partitions, walks drawn on the device: the same routed path with nothing to route."""
    g = make_graph(240, 2400, seed=6)
    k = OracleKernels()
    s = gv.solver.GraphSolver(32, kernels=k, num_sampler_per_worker=1, device_sampling=True, seed=3)
    s.build(g, batch_size=300, episode_size=2, num_partition=3)
    seen = []
    original = s._train_block
    s._train_block = lambda state, hp, tp, pool: (seen.append((hp, tp)), original(state, hp, tp, pool))[1]
    s.train("DeepWalk", num_epoch=3, augmentation_step=2, random_walk_length=6, random_walk_batch_size=4)
    assert s._sampler is None and set(seen) == {(hp, tp) for hp in range(3) for tp in range(3)}
    assert s.batch_id % (9 * 2) == 0 and np.abs(s.context_embeddings).max() > 0


def test_auto_build_rules_match_the_reference_solver():
    """num_partition = auto and episode_size = auto as SolverMixin::build of the reference resolved them
    (solver.h:365-434; tests/golden/reference_solver.npz, one worker)."""
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_solver.npz"))
    args, info = G["cfg_auto_1_args"], G["cfg_auto_1_info"]
    g = gv.graph.Graph()
    g.load(G["edges"].astype(np.int64), as_undirected=bool(args[1]))
    s = gv.solver.GraphSolver(128, kernels=OracleKernels(), num_sampler_per_worker=1)
    s.build(g, batch_size=int(args[5]))
    assert (s.num_vertex, s.num_edge) == (int(info[0]), int(info[1]))
    assert s.num_partition == int(info[3]) and s.episode_size == int(info[4])


def _auc_worker(rank, world, port, out_path, num_partition):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import logging
        gv.init_logging(logging.ERROR)
        G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_solver.npz"))
        n, e, communities, graph_seed, batch, episode, epochs = [int(x) for x in G["train_small_args"]]
        edges = synthetic.community_edges(n, e, num_community=communities, seed=graph_seed)
        train, (valid, test) = synthetic.link_prediction_split(edges, (100, 3, 3))
        g = gv.graph.Graph()
        g.load(train)
        s = gv.solver.GraphSolver(128, kernels=OracleKernels(), num_sampler_per_worker=2, seed=3)
        s.build(g, batch_size=batch, episode_size=episode, num_partition=num_partition)
        s.train(model="LINE", num_epoch=epochs, augmentation_step=1, log_frequency=1 << 30)
        if rank == 0:
            n2i = g.name2id
            keep = [(n2i[str(h)], n2i[str(t)], y) for h, t, y in zip(*test) if str(h) in n2i and str(t) in n2i]
            auc = link_prediction_auc(s.vertex_embeddings, s.context_embeddings, [k[0] for k in keep],
                                      [k[1] for k in keep], [k[2] for k in keep])
            np.save(out_path, np.array([auc]))
    finally:
        dist.destroy_process_group()


def test_two_workers_learn_what_the_reference_two_workers_learn(tmp_path):
    """Learning quality of the multi-GPU data path.  The reference's own training loop with 2 worker threads and 4
    partitions (partition loads and write-backs through host memory, solver.h:1349-1504; run on the host by
    oracle/ref_solver_harness.cpp) reached the link-prediction AUC stored in tests/golden/reference_solver.npz; two
    gloo workers of this repo (context shards pinned per worker, asynchronous all-gather of head shards, head groups
    interleaved) must reach it too.  A stale or misplaced shard costs far more than the +-0.002 of seed noise."""
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_solver.npz"))
    world, port, out = 2, _free_port(), str(tmp_path / "auc.npy")
    mp.spawn(_auc_worker, args=(world, port, out, 4), nprocs=world, join=True)
    auc = float(np.load(out)[0])
    ref2, ref1 = float(G["train_small_w2_p4_auc"]), float(G["train_small_w1_p1_auc"])
    print("2 workers / 4 partitions: AUC %.6f | reference loop: %.6f (2 workers / 4 partitions), %.6f (1 worker)"
          % (auc, ref2, ref1))
    assert abs(auc - ref2) <= 0.006 and abs(auc - ref1) <= 0.006
