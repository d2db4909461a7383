"""A/B timing of the attention kernels at the ESM2-3B shape, alternating rounds (the chip's clock wanders by +-10 % with load history:
compare medians of interleaved rounds, never two numbers from different processes).   python3 tools/attn64_ab.py [B] [T] [rounds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
nh, d = 40, 64
qkv = torch.empty((B * T, 3 * nh * d), dtype=torch.bfloat16, device=dev)
ops.fill_hash_(qkv, 1, "attn_only", 1.0)
inv = torch.ones((d // 2,), dtype=torch.float32, device=dev)
km, kv, _ = ops.mask_prepare(torch.ones((B, T), dtype=torch.int64, device=dev))
q, k, v = ops.qkv_post(qkv, inv, B, T, nh, nh, d, d ** -0.5 * 1.4426950408889634)
res = {3: [], 2: []}
for r in range(rounds):
    for mode in (3, 2):
        for _ in range(10):
            ops.attention(q, k, v, km, kv, d, 1.0, False, use_mfma=mode, log2_scores=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 100
        for _ in range(n):
            ops.attention(q, k, v, km, kv, d, 1.0, False, use_mfma=mode, log2_scores=True)
        e1.record()
        torch.cuda.synchronize()
        res[mode].append(e0.elapsed_time(e1) / n * 1e3)
fl = 4.0 * B * nh * T * T * d
for mode, name in ((3, "hand-placed"), (2, "general")):
    a = np.array(res[mode])
    print(f"{name:12s} B={B} T={T}: median {np.median(a):.1f} us ({fl / np.median(a) / 1e6:.0f} TFLOP/s)  rounds " + " ".join(f"{x:.1f}" for x in a), flush=True)
