"""Cost of grouping a block pool's batches by head row on the device with stock torch ops (experiment)."""
import time, torch
dev = "cuda:0"
nb, B = 217, 100000
pool = torch.randint(0, 1_000_000, (nb, B, 2), dtype=torch.int32, device=dev)
def timed(f, n=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def full_sort():
    idx = torch.argsort(pool[:, :, 1], dim=1)
    return torch.gather(pool, 1, idx.unsqueeze(-1).expand(-1, -1, 2))
def low16_sort():
    idx = torch.argsort((pool[:, :, 1] & 0xFFFF).to(torch.int16), dim=1)
    return torch.gather(pool, 1, idx.unsqueeze(-1).expand(-1, -1, 2))
def flat_sort():
    # one flat sort of (batch << 32 | head) keys
    keys = (torch.arange(nb, device=dev, dtype=torch.int64).unsqueeze(1) << 32) | (pool[:, :, 1].to(torch.int64) & 0xFFFFFFFF)
    idx = torch.argsort(keys.view(-1))
    return pool.view(-1, 2)[idx]
for name, f in (("argsort dim=1 int32", full_sort), ("argsort low 16 bits", low16_sort), ("flat int64 sort", flat_sort)):
    print("%-24s %.2f ms per block pool (%d x %d)" % (name, timed(f), nb, B), flush=True)
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from graphvite_amd import kernels as K
hip = K.HipKernels()
flat, out = pool.view(-1), torch.empty_like(pool).view(-1)
for rows in (1 << 20, 1 << 16):
    print("gvk_group_pairs %2d row bits  %.2f ms per block pool" % (rows.bit_length() - 1,
          timed(lambda: hip.group_pairs(flat, out, B, nb, rows))), flush=True)
