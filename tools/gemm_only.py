#!/usr/bin/env python3
"""A few launches of the product GEMM on fixed shapes (target for rocprofv3 PMC passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import ops
dev = torch.device("cuda:0")
def rand(shape, dtype=torch.bfloat16, scale=1.0):
    t = torch.empty(shape, dtype=dtype, device=dev); ops.fill_hash_(t, 1, f"g{shape}", scale); return t
for name, M, N, K, epi in [("fc1", 16384, 10240, 2560, 1), ("qkv", 16384, 7680, 2560, 0), ("fc2", 16384, 2560, 10240, 2), ("sq8k", 8192, 8192, 8192, 0)]:
    a, w, b = rand((M, K)), rand((N, K), scale=0.05), rand((N,), torch.float32, 0.1)
    out = torch.zeros((M, N), dtype=torch.float32, device=dev) if epi == 2 else None
    for _ in range(12):
        ops.gemm_nt(a, w, b, epilogue=epi, out=out, use_mfma=1)
    torch.cuda.synchronize()
