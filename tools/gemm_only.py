#!/usr/bin/env python3
"""A few launches of ONE GEMM shape (for rocprofv3 --pmc passes): python tools/gemm_only.py M N K [epilogue] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import ops  # noqa: E402

M, N, K = (int(a) for a in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
a = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
w = torch.empty((N, K), dtype=torch.bfloat16, device=dev)
ops.fill_hash_(a, 1, "a", 1.0)
ops.fill_hash_(w, 1, "w", 0.05)
bias = torch.zeros((N,), dtype=torch.float32, device=dev)
out = torch.zeros((M, N), dtype=torch.float32, device=dev) if epi == 2 else None
ws = ops.gemm_fix_workspace(dev)
for i in range(iters):
    ops.gemm_nt(a, w, None if epi == 3 else bias, epilogue=epi, out=out, use_mfma=1, fix_ws=ws, fix_epoch=i + 1)
torch.cuda.synchronize()
print("done", M, N, K, epi)
