"""Launcher of bench.py's loop for the CPU suite (tests/test_api_cpu.py::test_bench_loop_dry_run): the same block walk,
staging, asynchronous exchange and rank reduction, with the oracle stand-in injected for the HIP kernels and gloo for
RCCL.  A logic test of the multi-rank path that a 1-GPU box cannot host; the JSON line it prints says so."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
from fake_kernels import OracleKernels  # noqa: E402

if __name__ == "__main__":
    bench.main(sys.argv[1:], stand_in_kernels=OracleKernels())
