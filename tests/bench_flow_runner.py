"""Drives bench.py's real control flow on CPU tensors through the kernel emulator (launched by
tests/test_bench_flow.py, single-rank and under torch.distributed.run with gloo).  Test infrastructure only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from virtex_amd import _lib, build  # noqa: E402

_lib.use_library(build.build_emu())
import bench  # noqa: E402

bench.main(sys.argv[1:], device=torch.device("cpu"), backend="gloo")
