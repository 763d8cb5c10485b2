"""The lockstep loop of bench.lockstep_sequences (quick set) under the kernel trace: which kernels a time step of <= 11
pairs launches and how long the host takes to enqueue a step.   rocprofv3 --kernel-trace --stats ... -- python tools/lockstep_trace.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from pnec_amd import capi  # noqa: E402
from pnec_amd import tracks as tk  # noqa: E402

t0 = time.perf_counter()
r = bench.lockstep_sequences(torch.device("cuda:0"), capi, tk, quick=True, check_steps=0)
print(json.dumps({"steps": r["steps"], "pairs": r["pairs"], "wall_s": r["wall_s"], "ms_per_step": 1e3 * r["wall_s"] / r["steps"],
                  "whole_call_s": time.perf_counter() - t0}))
