"""A few launches of the ESM2-3B attention shape (16 x 1024 tokens, 40 heads, head_dim 64) in the towers' log2-scores form, for
rocprofv3 --pmc passes:  rocprofv3 --kernel-trace --pmc <counters> -d <dir> --output-format csv -- python3 tools/attn_only.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
B, T, nh, d = 16, 1024, 40, 64
qkv = torch.empty((B * T, 3 * nh * d), dtype=torch.bfloat16, device=dev)
ops.fill_hash_(qkv, 1, "attn_only", 1.0)
inv = torch.ones((d // 2,), dtype=torch.float32, device=dev)
km, kv, _ = ops.mask_prepare(torch.ones((B, T), dtype=torch.int64, device=dev))
q, k, v = ops.qkv_post(qkv, inv, B, T, nh, nh, d, d ** -0.5 * 1.4426950408889634)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ops.attention(q, k, v, km, kv, d, 1.0, False, use_mfma=1, log2_scores=True)
torch.cuda.synchronize()
