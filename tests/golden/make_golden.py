"""Generates tests/golden/reference_arithmetic.npz from the REFERENCE's own model / optimizer code compiled for
the host (oracle/ref_harness.cpp -> oracle/_ref/libgvref.so, built from /root/reference/include by oracle/Makefile),
and tests/golden/reference_alias.npz from the reference's own AliasTable (oracle/ref_alias_harness.cpp), and
tests/golden/reference_solver.npz from its solver front end and CPU samplers (oracle/ref_solver_harness.cpp).

Run here (the container that has /root/reference):   python tests/golden/make_golden.py
The fixture travels to the GPU box; /root/reference does not.  Every case stores the inputs and what the
reference arithmetic produced from them, so the oracle (and through it the HIP kernels) can be checked against
the reference without the reference being present.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import (ADAGRAD, ADAM, MOMENTUM, RMSPROP, SGD, Oracle, Reference, ReferenceSolver,  # noqa: E402
                        link_prediction_auc, reference_load, reference_train)

HP = {SGD: (0, 0, 0), MOMENTUM: (0.9, 0, 0), ADAGRAD: (0, 0, 1e-10), RMSPROP: (0.99, 0, 1e-8),
      ADAM: (0.9, 0.99, 1e-8)}


def main():
    ref = Reference()
    rng = np.random.default_rng(20260923)
    out = {}
    cases = [(dim, SGD) for dim in (32, 64, 96, 128, 256, 512)] + \
            [(dim, opt) for dim in (32, 128) for opt in (MOMENTUM, ADAGRAD, RMSPROP, ADAM)]
    for dim, opt in cases:
        N, B, k = 40, 60, 2
        v = rng.uniform(-0.5, 0.5, (N, dim)).astype(np.float32)
        c = rng.uniform(-0.5, 0.5, (N, dim)).astype(np.float32)
        pairs = rng.integers(0, N, (B, 2)).astype(np.uint32)  # {tail, head}, conflicts included (sequential order)
        negs = rng.integers(0, N, (B, k)).astype(np.uint32)
        nm = 0 if opt == SGD else (2 if opt == ADAM else 1)
        moments = [rng.uniform(0, 1e-2, (N, dim)).astype(np.float32) if i < 2 * nm else None for i in range(4)]
        key = "d%d_o%d" % (dim, opt)
        out[key + "_v_in"], out[key + "_c_in"], out[key + "_pairs"], out[key + "_negs"] = v.copy(), c.copy(), pairs, negs
        for i, m in enumerate(moments):
            if m is not None:
                out[key + "_m%d_in" % i] = m.copy()
        loss = ref.train(v, c, pairs, negs, 0.025, 0.005, 5.0, opt, moments, HP[opt])
        out[key + "_v_out"], out[key + "_c_out"], out[key + "_loss"] = v, c, loss
        for i, m in enumerate(moments):
            if m is not None:
                out[key + "_m%d_out" % i] = m
        out[key + "_logits"] = ref.predict(v, c, pairs)
    xs = np.concatenate([np.linspace(-30, 30, 121), [-100, -88.5, 0, 1e-8, 88.5, 100]]).astype(np.float32)
    out["sigmoid_x"] = xs
    out["sigmoid_y"] = np.array([ref.sigmoid(float(x)) for x in xs], np.float32)
    ids = np.array([0, 1, 499, 500, 999, 1000, 1500, 99999], np.int32)
    out["lr_batch_id"] = ids
    out["lr_linear"] = np.array([ref.lr(0.025, True, int(i), 1000) for i in ids], np.float32)
    out["lr_constant"] = np.array([ref.lr(0.025, False, int(i), 1000) for i in ids], np.float32)
    path = os.path.join(HERE, "reference_arithmetic.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%d arrays, %.1f KiB)" % (path, len(out), os.path.getsize(path) / 1024))

    # the reference's own AliasTable (oracle/ref_alias_harness.cpp -> oracle/_ref/libgvref_alias.so): build + sample
    alias = {}
    rng = np.random.default_rng(20260924)
    cases = {"one": np.array([3.0], np.float32), "uniform": np.ones(17, np.float32),
             "two_to_one": np.array([2, 1], np.float32), "tiny": np.full(5, 1e-30, np.float32),
             "with_zeros": np.array([0, 0, 1, 0, 4, 0.5, 0], np.float32),
             "pareto": (rng.pareto(1.2, 1000) + 1e-3).astype(np.float32),
             "degree_075": np.floor(rng.pareto(1.5, 4097) + 1).astype(np.float32) ** np.float32(0.75),
             "small_ints": rng.integers(1, 4, 333).astype(np.float32),
             "sparse": np.where(rng.random(2500) < 0.3, rng.random(2500), 0).astype(np.float32)}
    for name, w in cases.items():
        alias[name + "_w"] = w
        alias[name + "_prob"], alias[name + "_alias"] = ref.alias_build(w, 4)
        prob64, alias64 = ref.alias_build(w, 8)
        assert (prob64 == alias[name + "_prob"]).all() and (alias64 == alias[name + "_alias"]).all()
        rand = rng.random((257, 2))
        rand[0] = (0.0, 0.0)
        rand[1] = (np.nextafter(1.0, 0.0), np.nextafter(1.0, 0.0))
        alias[name + "_rand"] = rand
        alias[name + "_draws"] = ref.alias_sample(w, rand)
    path = os.path.join(HERE, "reference_alias.npz")
    np.savez_compressed(path, **alias)
    print("wrote %s (%d arrays, %.1f KiB)" % (path, len(alias), os.path.getsize(path) / 1024))

    # the reference's own solver front end (oracle/ref_solver_harness.cpp): graph store, partition, schedule, auto
    # episode size, alias tables, and the pools its three CPU samplers fill from the oracle's uniform streams
    oracle = Oracle()
    solver = {}
    rng = np.random.default_rng(20260925)
    n_vertex, n_edge = 300, 3000
    edges = np.stack([rng.integers(0, n_vertex, n_edge), rng.zipf(1.6, n_edge) % n_vertex], 1).astype(np.uint32)
    weights = (rng.pareto(2.0, n_edge) + 0.1).astype(np.float32)  # distinct weights: no ties in the partition
    solver["edges"], solver["weights"] = edges, weights
    seed = 20260925
    solver["seed"] = np.int64(seed)
    configs = {  # name: (weighted, undirected, workers, samplers per worker, partitions, batch, episode)
        "w_p4": (True, True, 2, 2, 4, 200, 3), "u_p4": (False, True, 2, 2, 4, 200, 3),
        "w_p1": (True, True, 1, 3, 1, 500, 2), "w_dir_p2": (True, False, 1, 2, 2, 300, 2),
        "auto_p": (True, True, 4, 1, 0, 250, 0), "auto_1": (False, True, 1, 1, 0, 100000, 0)}
    for name, (weighted, undirected, W, spw, P, B, episode) in configs.items():
        rs = ReferenceSolver(oracle, seed, edges, weights if weighted else None, undirected, W, spw, P, 1, B, episode)
        key = "cfg_" + name
        solver[key + "_args"] = np.array([weighted, undirected, W, spw, P, B, episode], np.int64)
        solver[key + "_info"] = np.array([rs.num_vertex, rs.num_edge, rs.num_directed_edge, rs.num_partition,
                                          rs.episode_size, rs.partition_size, rs.num_sampler, rs.num_worker], np.int64)
        labels, part, local, vw = rs.partition()
        uv, ew = rs.edges()
        solver[key + "_labels"], solver[key + "_part"], solver[key + "_local"] = labels, part, local
        solver[key + "_vertex_weights"], solver[key + "_uv"], solver[key + "_edge_weights"] = vw, uv, ew
        solver[key + "_schedule"] = rs.schedule()
        if name.startswith("auto"):
            continue
        # WorkerMixin::build_negative_sampler for every tail partition (degree ^ 0.75 in local order)
        for tp in range(rs.num_partition):
            solver[key + "_negative_prob_%d" % tp], solver[key + "_negative_alias_%d" % tp] = rs.negative_table(0, tp)
        solver[key + "_edge_pools"] = rs.sample("LINE", 1)
        prob, alias = rs.table(0)
        solver[key + "_edge_prob"], solver[key + "_edge_alias"] = prob, alias
    # walk modes on one partition: tables bit for bit, pools for distribution tests
    for name, (model, aug, length, batch, shuffle) in {"line2": ("LINE", 2, 5, 10, 2), "deepwalk": ("DeepWalk", 3, 8, 10, 1),
                                                       "node2vec": ("node2vec", 2, 6, 10, 1)}.items():
        rs = ReferenceSolver(oracle, seed, edges, weights, True, 1, 3, 1, 1, 200, 300)
        pools = rs.sample(model, aug, length, batch, shuffle, 0.5, 2.0)
        key = "walk_" + name
        solver[key + "_args"] = np.array([aug, length, batch, shuffle], np.int64)
        solver[key + "_pool"] = pools[0, 0]
        which = 2 if model == "node2vec" else 1
        count = rs.num_directed_edge if which == 2 else rs.num_vertex
        probs, aliases, sizes = [], [], []
        for i in range(count):
            prob, alias = rs.table(which, i)
            probs.append(prob), aliases.append(alias.astype(np.uint32)), sizes.append(len(prob))
        solver[key + "_table_prob"], solver[key + "_table_alias"] = np.concatenate(probs), np.concatenate(aliases)
        solver[key + "_table_sizes"] = np.array(sizes, np.int64)
    # the reference's text loaders: Graph::load_file and WordGraph::load_file_compact on committed inputs
    import tempfile
    rng = np.random.default_rng(20260926)
    names = ["n%d" % i for i in range(25)] + ["7", "007", "a-b", "x_y"]
    lines = ["# an edge list with comments, blank lines, weights and repeats", ""]
    for _ in range(120):
        u, v = rng.choice(names, 2)
        kind = rng.random()
        line = "%s %s" % (u, v) if kind < 0.4 else ("%s\t%s\t%.3f" % (u, v, rng.random() * 3 + 0.1))
        lines.append(line + ("   # trailing comment" if rng.random() < 0.1 else ""))
    lines += ["n1 n1 2.5", "", "n2   n3", "#n4 n5"]
    edge_text = "\n".join(lines) + "\n"
    vocab = ["w%d" % i for i in range(30)]
    corpus_text = "\n".join(" ".join(rng.choice(vocab, int(rng.integers(0, 20)))) +
                            ("  # note w0 w0" if rng.random() < 0.2 else "") for _ in range(80)) + "\nthe the the the\n"
    solver["loader_edge_text"] = np.frombuffer(edge_text.encode(), np.uint8)
    solver["loader_corpus_text"] = np.frombuffer(corpus_text.encode(), np.uint8)
    with tempfile.TemporaryDirectory() as tmp:
        edge_path, corpus_path = os.path.join(tmp, "edges.txt"), os.path.join(tmp, "corpus.txt")
        open(edge_path, "w").write(edge_text)
        open(corpus_path, "w").write(corpus_text)
        cases = {"file_und": (0, edge_path, 1, 0, False), "file_dir": (0, edge_path, 0, 0, False),
                 "file_und_norm": (0, edge_path, 1, 0, True), "file_dir_norm": (0, edge_path, 0, 0, True),
                 "corpus_w5_c1": (1, corpus_path, 5, 1, False), "corpus_w2_c3": (1, corpus_path, 2, 3, False),
                 "corpus_w3_c2_norm": (1, corpus_path, 3, 2, True)}
        for name, (kind, path, a, b, norm) in cases.items():
            got_names, uv, ew, vw, num_edge = reference_load(kind, path, a, b, norm)
            key = "loader_" + name
            solver[key + "_args"] = np.array([kind, a, b, norm], np.int64)
            solver[key + "_names"] = np.frombuffer("\n".join(got_names).encode(), np.uint8)
            solver[key + "_uv"], solver[key + "_edge_weights"], solver[key + "_vertex_weights"] = uv, ew, vw
            solver[key + "_num_edge"] = np.int64(num_edge)
    # The reference's WHOLE training loop (GraphSolver::train as written: sampler threads, schedule, partition loads,
    # negative sampler, lr schedule; the worker's kernel emulated by a sequential host loop over its own model code) on
    # the graph of the T3 parity test, three uniform seeds: link-prediction AUC.
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from graphvite_amd import synthetic  # the graph generator only; nothing of the product trains here
    community = synthetic.community_edges(20000, 400000, num_community=100, seed=3)
    train_edges, (valid, test) = synthetic.link_prediction_split(community, (100, 3, 3))
    aucs = []
    for train_seed in (17, 18, 19):
        rs = ReferenceSolver(oracle, train_seed, train_edges.astype(np.uint32), None, True, 1, 4, 1, 1, 500, 200)
        vertex, context, batch_id = reference_train(rs, "LINE", 50, 1)
        labels = rs.partition()[0]
        name2id = {int(label): i for i, label in enumerate(labels)}
        keep = [(name2id[int(h)], name2id[int(t)], y) for h, t, y in zip(*test) if int(h) in name2id and int(t) in name2id]
        aucs.append(link_prediction_auc(vertex, context, [k[0] for k in keep], [k[1] for k in keep], [k[2] for k in keep]))
        print("reference training loop, seed %d: %d batches, AUC %.6f" % (train_seed, batch_id, aucs[-1]), flush=True)
    solver["train_line_community_auc"] = np.array(aucs, np.float64)
    solver["train_line_community_args"] = np.array([20000, 400000, 100, 3, 500, 200, 50], np.int64)
    # ... and with several workers on a smaller graph: the reference's partition loads / write-backs through host memory
    # between its worker threads (what this repo replaces by pinned context shards + an all-gather of head shards)
    small = synthetic.community_edges(4000, 80000, num_community=40, seed=5)
    small_train, (valid, small_test) = synthetic.link_prediction_split(small, (100, 3, 3))
    for W, P in ((1, 1), (2, 2), (2, 4)):
        rs = ReferenceSolver(oracle, 3, small_train.astype(np.uint32), None, True, W, 2, P, 1, 200, 40)
        vertex, context, batch_id = reference_train(rs, "LINE", 150, 1)
        labels = rs.partition()[0]
        name2id = {int(label): i for i, label in enumerate(labels)}
        keep = [(name2id[int(h)], name2id[int(t)], y) for h, t, y in zip(*small_test)
                if int(h) in name2id and int(t) in name2id]
        auc = link_prediction_auc(vertex, context, [k[0] for k in keep], [k[1] for k in keep], [k[2] for k in keep])
        print("reference training loop, %d workers / %d partitions: AUC %.6f" % (W, P, auc), flush=True)
        solver["train_small_w%d_p%d_auc" % (W, P)] = np.float64(auc)
    solver["train_small_args"] = np.array([4000, 80000, 40, 5, 200, 40, 150], np.int64)
    path = os.path.join(HERE, "reference_solver.npz")
    np.savez_compressed(path, **solver)
    print("wrote %s (%d arrays, %.1f KiB)" % (path, len(solver), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
