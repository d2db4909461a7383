"""The decode step's attention alone (p2t_attention_decode) at Llama-3.1-8B sizes: B rows x 8 kv heads x 4 query heads x head_dim 128 over
a prompt of T keys + S generated ones, a ring of cache copies so that nothing is served from a cache.  python tools/decode_attn_bench.py [B] [T]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import _lib, ops  # noqa: E402
from p2t_hip.ops import ptr, stream  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1088
nh, nkv, d, S = 32, 8, 128, 32
Tp, G = ops.round_up(T, 64), 64
copies = 24
mk = lambda *s: torch.randn(s, device=dev, dtype=torch.float32).mul_(0.5).to(torch.bfloat16)
kp, vtp = [mk(B, nkv, Tp, d) for _ in range(copies)], [mk(B, nkv, d, Tp) for _ in range(copies)]
kg, vtg = mk(B, nkv, G, d), mk(B, nkv, d, G)
q = mk(B, nh, d)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
step = torch.tensor([S], dtype=torch.int32, device=dev)
out = torch.zeros((B, nh * d), dtype=torch.bfloat16, device=dev)
for use_mfma in (1, 0):
    i = [0]

    def run():
        c = i[0] % copies
        i[0] += 1
        _lib.call("p2t_attention_decode", ptr(q), ptr(kp[c]), ptr(vtp[c]), ptr(kg), ptr(vtg), ptr(lens), ptr(step), B, 1, nh, nkv, d, Tp, G, 1.0, 1,
                  _lib.BF16, use_mfma, ptr(out), nh * d, stream())
    for _ in range(copies):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    n = 96
    for _ in range(n):
        run()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    nbytes = 2 * 2 * B * nkv * d * (T + S + 1)
    print(f"decode attention B={B} T={T}+{S + 1} {'mfma' if use_mfma else 'lane-per-key'}: {ms * 1e3:.1f} us, {nbytes / ms / 1e6:.0f} GB/s of keys + values", flush=True)
