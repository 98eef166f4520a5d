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
