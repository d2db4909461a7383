"""Development check of the hand-placed attention kernel (csrc/attn_fwd64.hip) against the exact fp32-softmax kernel and the general
MFMA kernel on the same bf16 q, k, v; then timing at the ESM2-3B shape.   python3 tools/attn64_check.py [quick]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
L2E = 1.4426950408889634


def run(B, T, nh, nkv, d, causal, lens, seed, amp, non_prefix=False):
    qkv = torch.empty((B * T, (nh + 2 * nkv) * d), dtype=torch.bfloat16, device=dev)
    ops.fill_hash_(qkv, seed, "a64", amp)
    inv = torch.ones((d // 2,), dtype=torch.float32, device=dev)
    mask = np.zeros((B, T), dtype=np.int64)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    if non_prefix:
        rng = np.random.default_rng(seed)
        mask = (rng.random((B, T)) < 0.6).astype(np.int64)
        mask[:, 0] = 0
        for b in range(B):
            if mask[b].sum() == 0:
                mask[b, T // 2] = 1
    km, kv, _ = ops.mask_prepare(torch.from_numpy(mask).to(dev))
    q, k, v = ops.qkv_post(qkv, inv, B, T, nh, nkv, d, d ** -0.5 * L2E)
    lse_a = torch.zeros((B, nh, T), dtype=torch.float32, device=dev)
    lse_r = torch.zeros((B, nh, T), dtype=torch.float32, device=dev)
    a = ops.attention(q, k, v, km, kv, d, 1.0, causal, use_mfma=3, log2_scores=True, lse=lse_a).float().cpu().numpy().reshape(B, T, -1)
    r = ops.attention(q, k, v, km, kv, d, 1.0, causal, use_mfma=0, log2_scores=True, lse=lse_r).float().cpu().numpy().reshape(B, T, -1)
    g = ops.attention(q, k, v, km, kv, d, 1.0, causal, use_mfma=2, log2_scores=True).float().cpu().numpy().reshape(B, T, -1)
    la, lr = lse_a.cpu().numpy(), lse_r.cpu().numpy()
    worst, worst_g, worst_l = 0.0, 0.0, 0.0
    ok = np.isfinite(a).all()
    for b in range(B):
        rows = np.nonzero(mask[b])[0] if non_prefix else np.arange(lens[b])
        ra, rr, rg = a[b, rows, :nh * d], r[b, rows, :nh * d], g[b, rows, :nh * d]
        den = np.abs(rr).max() + 1e-30
        worst = max(worst, float(np.abs(ra - rr).max() / den))
        worst_g = max(worst_g, float(np.abs(rg - rr).max() / den))
        fl = np.isfinite(lr[b][:, rows])
        worst_l = max(worst_l, float(np.abs(la[b][:, rows][fl] - lr[b][:, rows][fl]).max()) if fl.any() else 0.0)
    tag = "ok " if (ok and worst < 1e-2 and worst_l < 2e-2) else "BAD"
    print(f"{tag} B={B} T={T} nh={nh}/{nkv} d={d} causal={int(causal)} lens={lens[:4]} np={int(non_prefix)} amp={amp}: "
          f"hand {worst:.2e} general {worst_g:.2e} lse {worst_l:.2e} finite={ok}", flush=True)
    if tag == "BAD":
        b = 0
        n = lens[0]
        err = np.abs(a[b, :n, :nh * d] - r[b, :n, :nh * d])
        rows = np.nonzero(err.max(axis=1) > 1e-2 * np.abs(r[b]).max())[0]
        print("   bad query rows (batch 0):", rows[:40], "... count", len(rows), flush=True)
        cols = np.nonzero(err.max(axis=0) > 1e-2 * np.abs(r[b]).max())[0]
        print("   bad columns:", cols[:40], "... count", len(cols), flush=True)
    return tag == "ok "


def main():
    good = True
    good &= run(1, 256, 1, 1, 64, False, [256], 1, 1.0)
    good &= run(1, 64, 1, 1, 64, False, [64], 2, 1.0)
    good &= run(1, 512, 2, 2, 64, False, [512], 3, 1.0)
    good &= run(2, 300, 4, 2, 64, False, [300, 77], 4, 2.0)
    good &= run(2, 1024, 3, 3, 64, False, [1024, 411], 5, 1.5)
    good &= run(2, 700, 4, 1, 64, True, [700, 123], 6, 1.0)
    good &= run(3, 130, 8, 8, 40, True, [130, 1, 64], 7, 4.0)
    good &= run(2, 333, 2, 2, 64, False, [333, 200], 8, 1.0, non_prefix=True)
    good &= run(1, 1024, 2, 2, 64, False, [1024], 9, 6.0)
    print("ALL OK" if good else "SOME BAD", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        return
    # timing at the ESM2-3B shape
    B, T, nh, d = 16, 1024, 40, 64
    qkv = torch.empty((B * T, 3 * nh * d), dtype=torch.bfloat16, device=dev)
    ops.fill_hash_(qkv, 1, "attn_only", 1.0)
    inv = torch.ones((d // 2,), dtype=torch.float32, device=dev)
    km, kv, _ = ops.mask_prepare(torch.ones((B, T), dtype=torch.int64, device=dev))
    q, k, v = ops.qkv_post(qkv, inv, B, T, nh, nh, d, d ** -0.5 * L2E)
    for mode in (3, 2):
        for _ in range(5):
            ops.attention(q, k, v, km, kv, d, 1.0, False, use_mfma=mode, log2_scores=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 50
        for _ in range(n):
            ops.attention(q, k, v, km, kv, d, 1.0, False, use_mfma=mode, log2_scores=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"mode {mode}: {us:.1f} us  {4 * B * nh * T * T * d / us / 1e6:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
