"""TEST INFRASTRUCTURE: bench.py's own loop on the CPU — the host build of the engine (tests/hostdev: the engine's sources over a
host stand-in for HIP, its kernels the oracle) instead of the HIP library, gloo instead of RCCL.  A logic test of the part of
the multi-GPU run that cannot be exercised on a 1-GPU box (block walk, staging, the in-place all-gather, bytes per collective);
its JSON line says "DRY RUN" in `data`.  Run as bench.py is run (directly, or one process per "GPU" under torch.distributed.run):

    python tests/bench_dry_run.py --gpus 1 --vertices 400 ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GVK_LIBRARY"] = os.path.join(ROOT, "tests", "hostdev", "build", "libgvk_host.so")
os.environ["GVK_ALLOW_TEST_LIBRARY"] = "1"
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    bench.DRY_RUN = True
    from graphvite_amd import _lib
    assert _lib.lib().gvh_is_host_build(), "the dry run needs the host build of the engine"
    bench.main()
