"""Synthetic power-law graphs (the reference's datasets need the network, python/graphvite/dataset.py:182-211).

Chung-Lu style: both endpoints of every edge are drawn independently from a Zipf-like node distribution
(expected degree of node i proportional to (i + 1)^(-1 / (gamma - 1))); self loops are redrawn, duplicate edges
are kept (the reference's Graph keeps multi-edges too, include/instance/graph.cuh:124-153)."""
import numpy as np


def power_law_edges(num_vertex, num_edge, gamma=2.3, seed=1024):
    """uint32 [num_edge, 2] undirected edge lines over vertices 0 .. num_vertex-1."""
    rng = np.random.default_rng(seed)
    w = np.arange(1, num_vertex + 1, dtype=np.float64) ** (-1.0 / (gamma - 1.0))
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    # a random relabeling so that vertex id carries no degree information
    label = rng.permutation(num_vertex).astype(np.uint32)
    out = np.empty((num_edge, 2), np.uint32)
    filled = 0
    while filled < num_edge:
        n = num_edge - filled
        u = np.searchsorted(cdf, rng.random(n), side="right")
        v = np.searchsorted(cdf, rng.random(n), side="right")
        keep = u != v
        m = int(keep.sum())
        out[filled:filled + m, 0] = label[np.minimum(u[keep], num_vertex - 1)]
        out[filled:filled + m, 1] = label[np.minimum(v[keep], num_vertex - 1)]
        filled += m
    return out


def degrees(edges, num_vertex):
    """Weighted degree of the as-undirected graph with unit weights (Graph::add_edge, graph.cuh:146-151)."""
    return (np.bincount(edges[:, 0], minlength=num_vertex) + np.bincount(edges[:, 1], minlength=num_vertex)).astype(
        np.float32)


def link_prediction_split(edges, portions=(100, 1, 1), seed=1024):
    """The reference's Dataset.link_prediction_split (python/graphvite/dataset.py:318-361) on an in-memory edge
    array: every edge line goes to split searchsorted(cumsum(portions) / sum, rand()); every non-train split gets
    as many random false edges (u != v, neither (u, v) nor (v, u) an edge) as it has true ones.
    Returns (train_edges [n, 2], [(H, T, Y) for each further split]).  Node order for the false edges is the
    sorted label order (the reference iterates a Python set of strings, whose order is not reproducible)."""
    rs = np.random.RandomState(seed)
    cum = np.cumsum(portions, dtype=np.float32) / np.sum(portions)
    which = np.searchsorted(cum, rs.rand(len(edges)))
    nodes = np.unique(edges)
    keys = set((edges[:, 0].astype(np.int64) << 32 | edges[:, 1].astype(np.int64)).tolist())
    splits = []
    for i in range(1, len(portions)):
        true = edges[which == i]
        H, T = [], []
        while len(H) < len(true):
            u = int(nodes[int(rs.rand() * len(nodes))])
            v = int(nodes[int(rs.rand() * len(nodes))])
            if u != v and (u << 32 | v) not in keys and (v << 32 | u) not in keys:
                H.append(u)
                T.append(v)
        splits.append((np.concatenate([true[:, 0], np.asarray(H, edges.dtype)]),
                       np.concatenate([true[:, 1], np.asarray(T, edges.dtype)]),
                       np.concatenate([np.ones(len(true), np.int64), np.zeros(len(H), np.int64)])))
    return edges[which == 0], splits


def community_edges(num_vertex, num_edge, num_community=50, p_in=0.9, seed=7):
    """Planted-partition graph with near-uniform degrees: u uniform, v in u's community with probability p_in.
    Unlike the power-law graphs it has no hubs, so concurrent (Hogwild) and sequential training see almost no
    same-row conflicts — the regime in which the two must agree closely."""
    rng = np.random.default_rng(seed)
    size = (num_vertex + num_community - 1) // num_community
    out = np.empty((num_edge, 2), np.uint32)
    filled = 0
    while filled < num_edge:
        n = num_edge - filled
        u = rng.integers(0, num_vertex, n)
        inside = rng.random(n) < p_in
        base = (u // size) * size
        width = np.minimum(size, num_vertex - base)
        v = np.where(inside, base + (rng.random(n) * width).astype(np.int64), rng.integers(0, num_vertex, n))
        keep = u != v
        m = int(keep.sum())
        out[filled:filled + m, 0] = u[keep]
        out[filled:filled + m, 1] = v[keep]
        filled += m
    return out


def hub_community_edges(num_vertex, num_edge, gamma=2.3, num_community=40, p_in=0.7, seed=7):
    """Degree-corrected planted partition: hub-heavy like `power_law_edges` (u drawn from the same Zipf-like node
    distribution) AND learnable like `community_edges` (with probability p_in, v is drawn from the same distribution
    restricted to u's community; communities are the residue classes of the degree rank, so every community has its
    share of hubs).  The stand-in for the reference's social-network datasets (BlogCatalog: 10 312 nodes, 333 983 edges,
    39 groups, maximum degree 3 992), which need the network (python/graphvite/dataset.py:182-211)."""
    rng = np.random.default_rng(seed)
    K = int(num_community)
    w = np.arange(1, num_vertex + 1, dtype=np.float64) ** (-1.0 / (gamma - 1.0))
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    members = [np.arange(k, num_vertex, K) for k in range(K)]          # ranks of community k, ascending
    member_cdf = []
    for m in members:
        c = np.cumsum(w[m])
        member_cdf.append(c / c[-1])
    label = rng.permutation(num_vertex).astype(np.uint32)
    out = np.empty((num_edge, 2), np.uint32)
    filled = 0
    while filled < num_edge:
        n = num_edge - filled
        u = np.minimum(np.searchsorted(cdf, rng.random(n), side="right"), num_vertex - 1)
        v = np.minimum(np.searchsorted(cdf, rng.random(n), side="right"), num_vertex - 1)
        inside = rng.random(n) < p_in
        r = rng.random(n)
        community = u % K
        for k in range(K):
            pick = np.nonzero(inside & (community == k))[0]
            if pick.size:
                j = np.minimum(np.searchsorted(member_cdf[k], r[pick], side="right"), len(members[k]) - 1)
                v[pick] = members[k][j]
        keep = u != v
        m = int(keep.sum())
        out[filled:filled + m, 0] = label[u[keep]]
        out[filled:filled + m, 1] = label[v[keep]]
        filled += m
    return out
