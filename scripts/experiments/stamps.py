"""Where the time of a train_hot_kernel launch goes: summary of the stamp files the timestamp library writes
(make -C graphvite_amd/csrc ts; GVK_LIBRARY=graphvite_amd/csrc/build/ts/libgvk_ts.so GVK_ALLOW_TEST_LIBRARY=1
GVK_STAMP_FILE=<prefix> python bench.py ...).  A record is eight 100 MHz stamps of thread 0 of a workgroup: role (1 long
chain, 2 short chains, 3 pairs, 0 nothing to do), start, record here, rows here, own steps done, all steps done, stored, entries.

    python scripts/experiments/stamps.py <prefix>.<n> ...
"""
import sys

import numpy as np


def us(x):
    return x * 0.01


for name in sys.argv[1:]:
    raw = np.fromfile(name, dtype=np.uint8)
    header = raw[:32].view(np.int32)
    grid = int(header[0])
    r = raw[32:32 + grid * 64].view(np.uint64).reshape(grid, 8).astype(np.int64)
    t0 = r[:, 1][r[:, 1] > 0].min()
    end = r[:, 1:7].max()
    print("%s: grid %d blocks (long %d, pairs %d, short %d, copy %d, order %d); first start to last stamp %.2f us" % (
        name, grid, header[1], header[2], header[3], header[4], header[5], us(end - t0)))
    start = r[:, 1] - t0
    print("  workgroup starts: median %.2f us, 90%% %.2f, last %.2f" % (us(np.median(start)), us(np.percentile(start, 90)), us(start.max())))
    L = r[r[:, 0] == 1]
    if len(L):
        print("  long chains (%d workgroups, entries median %d max %d):" % (len(L), np.median(L[:, 7]), L[:, 7].max()))
        for label, a, b in (("start", None, 1), ("record", 1, 2), ("own row + entries", 2, 3), ("task 0's steps", 3, 4), ("every task's steps", 3, 5), ("composed + stored", 5, 6)):
            d = (L[:, b] - t0) if a is None else (L[:, b] - L[:, a])
            print("    %-20s mean %.2f us, max %.2f" % (label, us(d.mean()), us(d.max())))
        print("    %-20s mean %.2f us, max %.2f" % ("done at", us((L[:, 6] - t0).mean()), us((L[:, 6] - t0).max())))
        top = L[np.argsort(-L[:, 7])[:5]]
        for row in top:
            print("      %5d entries: start %.2f record +%.2f rows +%.2f steps +%.2f (all %.2f) stored +%.2f = done at %.2f us" % (
                row[7], us(row[1] - t0), us(row[2] - row[1]), us(row[3] - row[2]), us(row[4] - row[3]), us(row[5] - row[3]), us(row[6] - row[5]), us(row[6] - t0)))
    S = r[r[:, 0] == 2]
    if len(S):
        print("  short chains (%d workgroups with work): start mean %.2f max %.2f | record %.2f (max %.2f) | rows + steps %.2f (max %.2f) | done at mean %.2f max %.2f us" % (
            len(S), us((S[:, 1] - t0).mean()), us((S[:, 1] - t0).max()), us((S[:, 2] - S[:, 1]).mean()), us((S[:, 2] - S[:, 1]).max()),
            us((S[:, 5] - S[:, 2]).mean()), us((S[:, 5] - S[:, 2]).max()), us((S[:, 5] - t0).mean()), us((S[:, 5] - t0).max())))
    P = r[r[:, 0] == 3]
    if len(P):
        print("  pairs (%d workgroups): start mean %.2f max %.2f | a sample %.2f us (max %.2f) | done at mean %.2f max %.2f us" % (
            len(P), us((P[:, 1] - t0).mean()), us((P[:, 1] - t0).max()), us((P[:, 5] - P[:, 1]).mean()), us((P[:, 5] - P[:, 1]).max()),
            us((P[:, 5] - t0).mean()), us((P[:, 5] - t0).max())))
