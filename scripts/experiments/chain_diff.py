import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from graphvite_amd import kernels as K
from oracle_lib import Oracle
import test_hub_chains_gpu as T
hip, oracle = K.HipKernels(), Oracle()
DEV = "cuda:0"
for cap in (0, 8192, 64):
    dim, k = 128, 1
    rng = np.random.default_rng(dim * 10 + k)
    N, B, batches, kv, kc = 1 << 15, 1500, 1, 24, 40
    hip.set_tuning(8, cap)
    v = (rng.uniform(-0.5, 0.5, (N, dim)) * 0.05).astype(np.float32)
    c = (rng.uniform(-0.5, 0.5, (N, dim)) * 0.05).astype(np.float32)
    pool, w = T.hub_case(rng, N, B, batches, kv, kc)
    table = T.negative_table(w, False)
    opt = K.OptimizerSpec("SGD", 0.025, 0.005)
    dpool = torch.from_numpy(pool.view(np.int32)).to(DEV)
    ws = torch.zeros(hip.hot_plan(B, k, kv, kc, batches), dtype=torch.uint8, device=DEV)
    hip.hot_build(ws, dpool, B, batches, k, table, 5, 7, kv, kc)
    torch.cuda.synchronize()
    chains = kv + kc
    cap_entries, entry_capacity, off = T.layout(B, k, chains, batches, cap)
    raw = ws.cpu().numpy()
    starts = raw[:batches * (chains + 1) * 4].view(np.uint32).reshape(batches, chains + 1)
    entries = raw[off:off + batches * entry_capacity * 4].view(np.uint32).reshape(batches, entry_capacity)
    negs = torch.zeros(B * k, dtype=torch.int32, device=DEV)
    hip.negative_draw(table, 5, 7, negs, B, k)
    nb = negs.cpu().numpy().view(np.uint32).reshape(B, k)
    ov, oc = v.copy(), c.copy()
    oracle.train_hot(ov, oc, pool, nb, oracle.lr(0.025, True, 7, 100), 0.005, 5.0, kv, kc, starts[0], entries[0, :starts[0][-1]], cap_entries)
    tv, tc = torch.from_numpy(v).to(DEV), torch.from_numpy(c).to(DEV)
    loss = torch.zeros(B, device=DEV)
    hip.train_episode_hot(tv, tc, dpool, loss, opt, k, 5.0, table, 5, 7, 100, batches, B, ws, kv, kc, serialized=True)
    torch.cuda.synchronize()
    sv, sc = tv.cpu().numpy(), tc.cpu().numpy()
    lens = np.diff(starts[0].astype(np.int64))
    dv = np.abs(sv[:kv] - ov[:kv]).max(1); dc = np.abs(sc[:kc] - oc[:kc]).max(1)
    print("cap", cap_entries, "vertex chains (len, diff, moved):", [(int(lens[i]), float("%.2g" % dv[i]), float("%.2g" % np.abs(sv[i]-v[i]).max())) for i in range(6)])
    print("   context chains:", [(int(lens[kv + i]), float("%.2g" % dc[i]), float("%.2g" % np.abs(sc[i]-c[i]).max())) for i in range(6)])
hip.set_tuning(8, 0)
