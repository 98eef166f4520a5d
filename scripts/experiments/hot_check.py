"""Experiment / bring-up (GPU): the hub-row chains (gvk_hot_build, gvk_train_episode_hot) against the oracle.
(1) work lists: per chain the same multiset of entries as the oracle's lists from the same negatives;
(2) serialized launches = the oracle's gvo_train_hot on the GPU's own lists (hub rows and the rows of conflict-free samples);
(3) the fused launch next to the serialized one."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from graphvite_amd import kernels as K  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

hip, oracle = K.HipKernels(), Oracle()
DEV = "cuda:0"


def case(dim, k, cap, seed=0, N=1 << 16, B=4000, KV=24, KC=40, batches=2, scale=0.05):
    rng = np.random.default_rng(seed)
    hip.set_tuning(8, cap)  # GVK_TUNE_CHAIN_CAP
    v = rng.uniform(-0.5, 0.5, (N, dim)).astype(np.float32) * scale
    c = rng.uniform(-0.5, 0.5, (N, dim)).astype(np.float32) * scale
    # heads / tails: 40 % hub rows (skewed), the rest distinct cold rows
    def column(hot, lo):
        ids = lo + rng.permutation(N // 4)[:batches * B]
        pick = rng.random(batches * B) < 0.4
        ids[pick] = np.minimum((rng.pareto(1.0, pick.sum()) * 2).astype(np.int64), hot - 1)
        return ids
    heads, tails = column(KV, N // 4), column(KC, N // 2)
    pool = np.stack([tails, heads], 1).astype(np.uint32)
    w = np.ones(N, np.float32)
    w[:KC] = N * 0.3 / KC  # 30 % of the negatives are hub rows
    w[KC:3 * N // 4] = 1e-3
    prob, alias, packed = K.alias_build(w)
    table = K.packed_to_device(packed, DEV)
    opt = K.OptimizerSpec("SGD", 0.025, 0.005)
    dpool = torch.from_numpy(pool.view(np.int32)).to(DEV)
    nbytes = hip.hot_plan(B, k, KV, KC, batches)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    first_id, total = 7, 100
    hip.hot_build(ws, dpool, B, batches, k, table, 5, first_id, KV, KC)
    torch.cuda.synchronize()
    # layout as gvk_hot_plan lays it out (hot_layout in gvk_kernels.hip)
    chains = KV + KC
    cap_e = (max(cap, 1) if cap else 256)
    cap_e = (cap_e + k) // (k + 1) * (k + 1)
    entry_capacity = 2 * (k + 1) * B
    align = lambda x: (x + 255) // 256 * 256
    off_entries = align(batches * (chains + 1) * 4)
    raw = ws.cpu().numpy()
    starts = raw[:batches * (chains + 1) * 4].view(np.uint32).reshape(batches, chains + 1)
    entries = raw[off_entries:off_entries + batches * entry_capacity * 4].view(np.uint32).reshape(batches, entry_capacity)
    negs = torch.zeros(batches, B * k, dtype=torch.int32, device=DEV)
    ov, oc = v.copy(), c.copy()
    worst = 0
    for b in range(batches):
        hip.negative_draw(table, 5, first_id + b, negs[b], B, k)
        nb = negs[b].cpu().numpy().view(np.uint32).reshape(B, k)
        pb = pool[b * B:(b + 1) * B]
        st, en = oracle.hot_lists(pb, nb, KV, KC)
        assert (st == starts[b]).all(), "chain_start differs in batch %d" % b
        for ch in range(chains):
            assert (np.sort(en[st[ch]:st[ch + 1]]) == np.sort(entries[b, st[ch]:st[ch + 1]])).all(), "entries of chain %d differ" % ch
        worst = max(worst, int(np.diff(st).max()))
        lr = oracle.lr(0.025, True, first_id + b, total)
        oracle.train_hot(ov, oc, pb, nb, lr, 0.005, 5.0, KV, KC, starts[b], entries[b, :st[-1]], cap_e)
    out = {}
    for name, serialized in (("serialized", True), ("fused", False)):
        tv, tc = torch.from_numpy(v).to(DEV), torch.from_numpy(c).to(DEV)
        loss = torch.zeros(B, device=DEV)
        hip.train_episode_hot(tv, tc, dpool, loss, opt, k, 5.0, table, 5, first_id, total, batches, B, ws, KV, KC,
                              serialized=serialized)
        torch.cuda.synchronize()
        out[name] = (tv.cpu().numpy(), tc.cpu().numpy())
    # rows of samples that share a cold row with another sample are Hogwild in the pair phase: left out
    allneg = negs.cpu().numpy().view(np.uint32).reshape(batches * B, k)
    cold_ctx = np.concatenate([pool[:, 0][pool[:, 0] >= KC], allneg[allneg >= KC]])
    ids, counts = np.unique(cold_ctx, return_counts=True)
    dirty_ctx = set(ids[counts > 1].tolist())
    hid, hcount = np.unique(pool[:, 1][pool[:, 1] >= KV], return_counts=True)
    dirty_head = set(hid[hcount > 1].tolist())
    bad = np.array([(int(pool[s, 0]) in dirty_ctx) or (int(pool[s, 1]) in dirty_head) or any(int(x) in dirty_ctx for x in allneg[s])
                    for s in range(batches * B)])
    keep_v = np.ones(N, bool)
    keep_c = np.ones(N, bool)
    keep_v[pool[bad, 1]] = False
    keep_c[pool[bad, 0]] = False
    keep_c[allneg[bad].reshape(-1)] = False
    keep_v[:KV] = True
    keep_c[:KC] = True
    sv, sc = out["serialized"]
    fv, fc = out["fused"]
    print("dim %3d k %d cap %4d: longest chain %5d entries, %d of %d samples left out | serialized vs oracle: hub rows %.3g / %.3g, "
          "others %.3g / %.3g | fused vs serialized: hub rows %.3g / %.3g (hub rows moved %.3g / %.3g), other rows %.3g / %.3g "
          "(moved %.3g / %.3g)" % (
              dim, k, cap_e, worst, bad.sum(), len(bad), np.abs(sv[:KV] - ov[:KV]).max(), np.abs(sc[:KC] - oc[:KC]).max(),
              np.abs(sv[keep_v] - ov[keep_v]).max(), np.abs(sc[keep_c] - oc[keep_c]).max(), np.abs(fv[:KV] - sv[:KV]).max(),
              np.abs(fc[:KC] - sc[:KC]).max(), np.abs(sv[:KV] - v[:KV]).max(), np.abs(sc[:KC] - c[:KC]).max(),
              np.abs(fv[keep_v] - sv[keep_v]).max(), np.abs(fc[keep_c] - sc[keep_c]).max(), np.abs(sv[keep_v] - v[keep_v]).max(),
              np.abs(sc[keep_c] - c[keep_c]).max()), flush=True)
    bad_v = np.abs(sv[keep_v] - ov[keep_v]) > 2e-4 * np.abs(ov[keep_v]) + 2e-6
    bad_c = np.abs(sc[keep_c] - oc[keep_c]) > 2e-4 * np.abs(oc[keep_c]) + 2e-6
    if bad_v.any() or bad_c.any():
        print('    MISMATCH: vertex rows', np.nonzero(keep_v)[0][np.unique(np.nonzero(bad_v)[0])][:10], 'context rows', np.nonzero(keep_c)[0][np.unique(np.nonzero(bad_c)[0])][:10], flush=True)


for dim, k, cap in ((128, 1, 0), (128, 1, 16), (128, 3, 0), (128, 3, 10), (32, 1, 0), (64, 1, 8), (96, 1, 0), (96, 2, 12), (256, 1, 0),
                    (512, 1, 32)):
    case(dim, k, cap)
hip.set_tuning(8, 0)
print("hub-row chains: ok")
