"""Generates tests/golden/reference_arithmetic.npz from the REFERENCE's own model / optimizer code compiled for
the host (oracle/ref_harness.cpp -> oracle/_ref/libgvref.so, built from /root/reference/include by oracle/Makefile),
and tests/golden/reference_alias.npz from the reference's own AliasTable (oracle/ref_alias_harness.cpp).

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
from oracle_lib import ADAGRAD, ADAM, MOMENTUM, RMSPROP, SGD, Reference  # noqa: E402

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


if __name__ == "__main__":
    main()
