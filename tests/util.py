"""Shared generators for the tests (seeded; numpy only)."""
import numpy as np


def init_tables(rng, n_vertex, n_context, dim, zero_context=False):
    """vertex ~ U(-0.5/dim, 0.5/dim) like GraphSolver::init_embeddings (include/instance/graph.cuh:724-731);
    context random of the same scale unless zero_context (the reference's init) is asked for."""
    v = rng.uniform(-0.5 / dim, 0.5 / dim, (n_vertex, dim)).astype(np.float32)
    if zero_context:
        c = np.zeros((n_context, dim), np.float32)
    else:
        c = rng.uniform(-0.5 / dim, 0.5 / dim, (n_context, dim)).astype(np.float32)
    return v, c


def conflict_free_batch(rng, n_vertex, n_context, batch_size, k):
    """All heads distinct, all tails and negatives distinct: the parallel kernel and the sequential oracle
    then compute the same thing. Records are {tail, head}."""
    assert batch_size <= n_vertex and batch_size * (k + 1) <= n_context
    heads = rng.permutation(n_vertex)[:batch_size]
    ctx = rng.permutation(n_context)[:batch_size * (k + 1)]
    tails = ctx[:batch_size]
    negatives = ctx[batch_size:].reshape(batch_size, k)
    pairs = np.stack([tails, heads], 1).astype(np.uint32)
    return pairs, np.ascontiguousarray(negatives.astype(np.uint32))


def random_batch(rng, n_vertex, n_context, batch_size, k):
    pairs = np.stack([rng.integers(0, n_context, batch_size), rng.integers(0, n_vertex, batch_size)], 1)
    negatives = rng.integers(0, n_context, (batch_size, k))
    return pairs.astype(np.uint32), negatives.astype(np.uint32)


def power_law_weights(rng, n, exponent=0.75):
    """Degree-like weights: Zipf-ish degrees raised to the negative-sampling exponent."""
    deg = np.floor(rng.pareto(1.5, n) + 1).astype(np.float32)
    return deg ** np.float32(exponent)
