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


def compare_auc(label, here, reference, tolerance=0.002):
    """T3: mean AUC here against the mean of the reference's own training loop, with the standard error of the
    difference (seeds are independent random streams on both sides).  Asserts |difference| <= tolerance and says on its line
    whether the verdict survives two standard errors ("clear") or not ("MARGINAL": |difference| + 2 SE > tolerance)."""
    here, reference = np.asarray(here, np.float64), np.asarray(reference, np.float64)
    reference = reference[~np.isnan(reference)]
    difference = float(here.mean() - reference.mean())
    se = float(np.sqrt((here.std(ddof=1) ** 2 / len(here) if len(here) > 1 else 0.0) +
                       (reference.std(ddof=1) ** 2 / len(reference) if len(reference) > 1 else 0.0)))
    verdict = "clear" if abs(difference) + 2 * se <= tolerance else "MARGINAL" if abs(difference) <= tolerance else "OUTSIDE"
    print("%s: AUC here %s (mean %.6f, %d seeds) | reference training loop %s (mean %.6f, %d seeds) | difference %+.6f, SE %.6f: %s" % (
        label, " ".join("%.6f" % a for a in here), here.mean(), len(here), " ".join("%.6f" % a for a in reference), reference.mean(),
        len(reference), difference, se, verdict))
    assert abs(difference) <= tolerance, (label, difference, se)
    return difference, se
