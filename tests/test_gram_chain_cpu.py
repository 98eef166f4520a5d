"""Long chains by Gram matrices (`long_chain_gram`, graphvite_amd/csrc/gvk_kernels.hip; GVK_TUNE_HOT_GRAM, off by default) —
a lane-level twin of the device code on the CPU, checked against the oracle.

The device function could not be run when it was written (no GPU minutes were left in round 4), and what can go wrong in it is
index arithmetic: which lane holds which chunk of which row, where `v_mfma_f32_16x16x4_f32` puts its results, which lane a DPP
row broadcast or a wavefront shuffle reads.  The twin below is the same algorithm on arrays of 64 lanes, one tile per quarter of a wavefront per round (the device function
runs two rounds' tiles in one pass, their recurrences side by side: tests/test_gram_simt_cpu.py runs its own source)
(four of them: the wavefronts of a workgroup), with the matrix instruction, the broadcasts and the shuffles in the lane maps of
/opt/skills/guides/cdna_hip_programming.md ("A[l&15][k=l>>4] / B[k=l>>4][l&15]", "col=lane&15, row=(lane>>4)*4+reg_idx"); it is
compared with the oracle's tasks-of-16 form (`gvo_hot_unit_chains`, cap 16, max_tasks 64: oracle/gv_oracle.c), which is what
the device function computes in exact arithmetic.  The GPU test of the real kernel against the same oracle call is
tests/test_hub_chains_gpu.py::test_long_chains_by_gram_matrices (runs when GVK_TEST_GRAM=1)."""
import ctypes as C

import numpy as np
import pytest

from oracle_lib import Oracle

F = np.float32
LANES = np.arange(64)
R, Q = LANES & 15, LANES >> 4
K_GRAM_TILES = 64


# ---- the wavefront's cross-lane primitives --------------------------------------------------------------------------

def shfl(x, source):
    return x[source]


def shfl_xor(x, mask):
    return x[LANES ^ mask]


def row_bcast(x, j):  # DPP row_newbcast:j — lane j of every 16-lane row to the whole row
    return x[(LANES & ~15) | j]


def group_sum16(x):  # the DPP butterfly over a 16-lane row: every lane of the row gets the row's sum
    return np.repeat(x.reshape(4, 16).sum(axis=1, dtype=F), 16).astype(F)


def mfma_16x16x4(a, b, acc):
    """v_mfma_f32_16x16x4_f32: lane l holds A[l & 15][k = l >> 4] and B[k = l >> 4][l & 15]; component v of lane l of the
    result is D[4 (l >> 4) + v][l & 15]."""
    A = np.zeros((16, 4), F)
    B = np.zeros((4, 16), F)
    A[R, Q] = a
    B[Q, R] = b
    D = (A.astype(np.float64) @ B.astype(np.float64)).astype(F)
    out = acc.copy()
    for v in range(4):
        out[:, v] += D[4 * Q + v, R]
    return out


def ballot(pred):
    return sum(1 << int(l) for l in LANES[pred])


# ---- the device function, one statement at a time ------------------------------------------------------------------------

def long_chain_gram(dim, a, h, chain, first, n):
    """`a`: vertex, context, hot_vertex, hot_context, wd, neg_weight; `h`: entries, mirror (`from`), lr, log2 decays.
    Returns the row the workgroup stores to the mirror `to`."""
    nch = dim // 16
    gram = np.full((4, 4, 256), np.nan, F)  # LDS, uninitialised
    part = np.full((4, dim), np.nan, F)
    positives = np.full(K_GRAM_TILES, np.nan, F)
    last, tiles = first + n, (n + 15) // 16
    rounds = (tiles + 15) // 16
    is_vertex = chain < a["hot_vertex"]
    partner_table = a["context"] if is_vertex else a["vertex"]
    partner_hot = a["hot_context"] if is_vertex else a["hot_vertex"]
    mirror = h["mirror"]  # [hot_vertex + hot_context][dim]
    partner_mirror = mirror[a["hot_vertex"]:] if is_vertex else mirror
    row0 = mirror[chain]
    own_row = row0.copy()  # LDS
    entries = h["entries"]

    def chunks(row_per_lane):  # lane (r, q): float4 chunks q, q + 4, ... of its row -> [64][nch][4]
        out = np.zeros((64, nch, 4), F)
        for m in range(nch):
            for l in LANES:
                f = 4 * m + Q[l]
                out[l, m] = row_per_lane[l][4 * f:4 * f + 4]
        return out

    def load_tile(t, e_own, qq):
        e = shfl(e_own, R + 16 * qq)
        ids = e & 0x7fffffff
        rows = []
        for l in LANES:
            row = partner_mirror[ids[l]] if ids[l] < partner_hot else partner_table[ids[l]]
            rows.append(row if first + 16 * t + R[l] < last else row0)
        return chunks(rows)

    own_chunks = chunks([own_row] * 64)

    # positives of every tile; this lane's entry of round 0 (per wavefront)
    e_first = np.zeros((4, 64), np.uint32)
    for wave in range(4):
        for p in range(rounds):
            t = wave + 4 * Q + 16 * p
            at = first + 16 * t + R
            e = np.where(at < last, entries[np.minimum(at, len(entries) - 1)], 0).astype(np.uint32)
            if p == 0:
                e_first[wave] = e
            mask = ballot((e >> 31) != 0)
            for l in LANES[R == 0]:
                positives[t[l]] = bin((mask >> (16 * int(Q[l]))) & 0xffff).count("1")
    # __syncthreads()
    all_ = F(0)
    for t in range(tiles):
        all_ = F(all_ + positives[t])
    lp, ln = F(h["log2_decay_positive"]), F(h["log2_decay_negative"])
    total = np.exp2(F(all_ * lp + (F(n) - all_) * ln), dtype=F)

    for p in range(rounds):
        for wave in range(4):
            t_wave = wave + 16 * p
            t_own = t_wave + 4 * Q
            exists = t_own < tiles
            at = first + 16 * t_own + R
            e_own = e_first[wave] if p == 0 else np.where(at < last, entries[np.minimum(at, len(entries) - 1)], 0).astype(np.uint32)
            logit = np.zeros(64, F)
            # 1. rows, Gram matrix, start logits
            if t_wave < tiles:
                for qq in range(4):
                    c = load_tile(t_wave + 4 * qq, e_own, qq)
                    g0, g1 = np.zeros((64, 4), F), np.zeros((64, 4), F)
                    dot = np.zeros(64, F)
                    for m in range(nch):
                        v = own_chunks[:, m]
                        g0 = mfma_16x16x4(c[:, m, 0], c[:, m, 0], g0)
                        g1 = mfma_16x16x4(c[:, m, 1], c[:, m, 1], g1)
                        g0 = mfma_16x16x4(c[:, m, 2], c[:, m, 2], g0)
                        g1 = mfma_16x16x4(c[:, m, 3], c[:, m, 3], g1)
                        dot = (dot + (c[:, m, 0] * v[:, 0] + c[:, m, 1] * v[:, 1] + c[:, m, 2] * v[:, 2] + c[:, m, 3] * v[:, 3])).astype(F)
                    for v in range(4):
                        gram[wave, qq, (4 * Q + v) * 16 + R] = g0[:, v] + g1[:, v]
                    dot = dot + shfl_xor(dot, 16)
                    dot = dot + shfl_xor(dot, 32)
                    logit = np.where(Q == qq, dot, logit).astype(F)
            # decay before / after the lane's own tile
            pb = np.zeros(64, F)
            for t in range(tiles):
                pb = (pb + np.where(t < t_own, positives[t], 0)).astype(F)
            pi = np.where(exists, positives[np.where(exists, t_own, 0)], 0).astype(F)
            nb = (16 * t_own).astype(F)
            ni = np.where(exists, np.minimum(n - 16 * np.minimum(t_own, tiles), 16), 0).astype(F)
            pa, na = (all_ - pb - pi).astype(F), (F(n) - nb - ni).astype(F)
            with np.errstate(over="ignore", invalid="ignore"):
                before_ = np.where(exists, np.exp2((pb * lp + (nb - pb) * ln).astype(F)), 0).astype(F)
                after_ = np.where(exists, np.exp2((pa * lp + (na - pa) * ln).astype(F)), 0).astype(F)
            # __syncthreads(); 2. the recurrence
            gr = np.zeros((64, 16), F)
            for l in LANES:
                gr[l] = gram[wave, Q[l], R[l] * 16:R[l] * 16 + 16] if exists[l] else 0
            valid, positive = at < last, (e_own >> 31) != 0
            signed_weight = np.where(valid, np.where(positive, F(1), F(-a["neg_weight"])), F(0)).astype(F)
            logit = np.where(exists, logit * before_, 0).astype(F)
            kappa = np.zeros(64, F)
            lr, wd = F(h["lr"]), F(a["wd"])
            for j in range(16):
                s, sw = row_bcast(logit, j), row_bcast(signed_weight, j)
                w = np.abs(sw)
                aj = (F(1) - lr * w * wd).astype(F)
                ex = np.exp(-np.abs(s), dtype=F)
                b = (lr * w * (np.where(s > 0, F(1), ex) / (F(1) + ex) - np.where(sw > 0, F(1), F(0)))).astype(F)
                with np.errstate(invalid="ignore"):
                    logit = (aj * logit - b * gr[:, j]).astype(F)
                kappa = np.where(R == j, b, kappa * aj).astype(F)
            kappa = np.where(exists & valid, kappa * after_, 0).astype(F)
            # 3. the rows again, weighted; rows of a quarter by DPP, rounds in LDS
            acc = np.zeros((64, nch, 4), F)
            if t_wave < tiles:
                for qq in range(4):
                    c = load_tile(t_wave + 4 * qq, e_own, qq)
                    cf = shfl(kappa, R + 16 * qq)
                    acc = (acc + cf[:, None, None] * c).astype(F)
            for m in range(nch):
                for x in range(4):
                    acc[:, m, x] = group_sum16(acc[:, m, x])
                for l in LANES[R == 0]:
                    f = 4 * m + Q[l]
                    if p == 0:
                        part[wave, 4 * f:4 * f + 4] = acc[l, m]
                    else:
                        part[wave, 4 * f:4 * f + 4] += acc[l, m]
        # __syncthreads()
    out = (total * own_row).astype(F)
    for wave in range(4):
        out = (out - part[wave]).astype(F)
    assert np.isfinite(out).all()
    return out


# ---- against the oracle ---------------------------------------------------------------------------------------------------------

def oracle_chain(oracle, dim, vertex, context, lr, wd, nw, kv, kc, chain_start, entries, cap, max_tasks):
    fn = oracle.lib.gvo_hot_unit_chains
    fp, up = np.ctypeslib.ndpointer(np.float32, flags="C"), np.ctypeslib.ndpointer(np.uint32, flags="C")
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, up, up, C.c_uint32, C.c_uint32, C.c_int]
    v, c = vertex.copy(), context.copy()
    assert fn(dim, v, c, lr, wd, nw, kv, kc, chain_start, entries, cap, max_tasks, 1) == 0
    return v, c


@pytest.mark.parametrize("dim,n", [(128, n) for n in (8, 16, 17, 100, 250, 256, 257, 600, 1024)] +
                         [(dim, n) for dim in (32, 64, 96) for n in (17, 257)])  # the other dims on two lengths
def test_gram_form_of_a_long_chain_matches_the_oracle(dim, n):
    rng = np.random.default_rng(1000 * dim + n)
    oracle = Oracle()
    kv, kc, rows = 6, 5, 400
    vertex = (rng.standard_normal((rows, dim)) * 0.35).astype(F)
    context = (rng.standard_normal((rows, dim)) * 0.35).astype(F)
    lr, wd, nw = F(0.025), F(0.005), F(5.0)
    for chain in (2, kv + 3):  # a head row's chain and a context row's
        partners = rng.integers(0, rows, n).astype(np.uint32)
        hub = rng.random(n) < 0.25  # a quarter of the entries name hub rows (read from the mirror)
        partners[hub] = rng.integers(0, kc if chain < kv else kv, hub.sum())
        labels = (rng.random(n) < 0.4).astype(np.uint32)
        # the chain before this one owns the first 37 entries of the list: the chain under test starts at an odd offset
        junk = rng.integers(kv + kc, rows, 37).astype(np.uint32)
        entries = np.concatenate([junk, partners | labels << 31]).astype(np.uint32)
        chain_start = np.zeros(kv + kc + 1, np.uint32)
        chain_start[chain:] = 37
        chain_start[chain + 1:] = 37 + n
        ov, oc = oracle_chain(oracle, dim, vertex, context, lr, wd, nw, kv, kc, chain_start, entries, 16, 64)
        want = ov[chain] if chain < kv else oc[chain - kv]
        a = dict(vertex=vertex, context=context, hot_vertex=kv, hot_context=kc, wd=wd, neg_weight=nw)
        h = dict(entries=entries, mirror=np.concatenate([vertex[:kv], context[:kc]]), lr=lr,
                 log2_decay_positive=F(np.log2(1.0 - float(lr) * float(wd))),
                 log2_decay_negative=F(np.log2(1.0 - float(lr) * float(nw) * float(wd))))
        got = long_chain_gram(dim, a, h, chain, 37, n)
        start = vertex[chain] if chain < kv else context[chain - kv]
        assert np.abs(want - start).max() > 1e-3  # the chain moved the row
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)  # rows of magnitude 1: fp32 sums in another order
