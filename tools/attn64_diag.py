"""Per-phase cycle anatomy of the hand-placed attention kernel (lab build: P2T_HIP_LIB=tools/build/libp2t_lab.so).
Phases (tools/gen_attn_fwd64.py stamp()): 0 prologue, 1 barrier wait, 2 slot setup + decide A, 3 segment 1, 4 mask B + decide B,
5 segment 2, 6 mask A + loop control, 7 tail; [8] = whole kernel incl. compiler prologue / epilogue.   python3 tools/attn64_diag.py [T]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("P2T_HIP_LIB", os.path.join(ROOT, "tools", "build", "libp2t_lab.so"))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1]
B, nh, d = 16, 40, 64
qkv = torch.empty((B * T, 3 * nh * d), dtype=torch.bfloat16, device=dev)
ops.fill_hash_(qkv, 1, "attn_only", 1.0)
inv = torch.ones((d // 2,), dtype=torch.float32, device=dev)
km, kv, _ = ops.mask_prepare(torch.ones((B, T), dtype=torch.int64, device=dev))
q, k, v = ops.qkv_post(qkv, inv, B, T, nh, nh, d, d ** -0.5 * 1.4426950408889634)
grid = ((T + 255) // 256) * nh * B
buf = torch.zeros((max(grid * 4 * 16, B * nh * T),), dtype=torch.float32, device=dev)
for _ in range(20):          # warm: clocks and caches in the steady state of back-to-back launches
    ops.attention(q, k, v, km, kv, d, 1.0, False, use_mfma=3, log2_scores=True)
names = ["prologue", "barrier", "setup+decideA", "seg1", "maskB+decideB", "seg2", "maskA+loop", "tail", "kernel", "n_it"]
for var in variants:
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.attention(q, k, v, km, kv, d, 1.0, 1 + var, use_mfma=3, log2_scores=True, lse=buf.view(B, nh, -1))
    e0.record()
    for _ in range(10):
        ops.attention(q, k, v, km, kv, d, 1.0, 1 + var, use_mfma=3, log2_scores=True, lse=buf.view(B, nh, -1))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    st = buf[:grid * 4 * 16].view(grid, 4, 16).cpu().numpy()
    n_it = st[0, 0, 9]
    print(f"variant {var} T={T}: {us:.1f} us per launch; mean cycles per wave: " + " ".join(f"{n}={st[:, :, i].mean():.0f}" for i, n in enumerate(names[:9])))
    # blocks of one CU in time order (wave 0 of each block): busy time, gaps between blocks, clock held inside a block
    w0 = st[:, 0, :]
    cu = (w0[:, 12].astype(np.int64) >> 8) & 0xFFFF            # HW_ID bits 8..: cu / sh / se
    key = w0[:, 13].astype(np.int64) * 65536 + cu
    durs, gaps, clk = [], [], []
    for kk in np.unique(key):
        rows = w0[key == kk]
        rows = rows[np.argsort((rows[:, 10] - rows[:, 10].min()) % 16777216)]
        for i in range(len(rows)):
            dur = (rows[i, 11] - rows[i, 10]) % 16777216
            durs.append(dur)
            clk.append(rows[i, 8] / max(dur, 1.0) / 10.0)       # cycles / (ticks * 10 ns) = GHz
            if i:
                gaps.append((rows[i, 10] - rows[i - 1, 11]) % 16777216)
    print(f"   {len(np.unique(key))} CUs; block {np.mean(durs) / 100:.2f} us, gap between blocks of a CU {np.mean(gaps) / 100:.2f} us (median {np.median(gaps) / 100:.2f}), "
          f"in-block clock {np.mean(clk):.3f} GHz, blocks per CU {len(w0) / len(np.unique(key)):.1f}")
    print(f"   epilogue (normalise, stage through LDS, stores): {st[:, :, 14].mean():.0f} cycles per block")
    print("   per iteration: barrier %.0f setupA %.0f seg1 %.0f maskB %.0f seg2 %.0f maskA %.0f  total %.0f" % (tuple(st[:, :, i].mean() / n_it for i in (1, 2, 3, 4, 5, 6)) + (st[:, :, 1:7].sum(axis=2).mean() / n_it,)), flush=True)
