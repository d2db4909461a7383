"""Where a K step of the four-wave fp8 GEMM goes (lab build: P2T_HIP_LIB=tools/build/libp2t_lab.so; plain bf16 store epilogue).
tile 2001: s_memtime stamps around the two barriers of every step, summed per workgroup; tile 2002: no DMA in the steady state
(garbage results, timing only).  A step is 64 MFMAs of 32 cycles = 2 048 matrix-pipe cycles.   python3 tools/fp8w4_diag.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("P2T_HIP_LIB", os.path.join(ROOT, "tools", "build", "libp2t_lab.so"))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import ops  # noqa: E402

dev = torch.device("cuda:0")


def rand(shape, scale=1.0):
    t = torch.empty(shape, dtype=torch.bfloat16, device=dev)
    ops.fill_hash_(t, 7, "fp8diag" + str(shape), scale)
    return t


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


shapes = [("esm qkv b64", 65536, 7680, 2560), ("esm fc2 b64", 65536, 2560, 10240), ("esm qkv b16", 16384, 7680, 2560), ("llama down", 8192, 4096, 14336)]
for name, M, N, K in shapes:
    a8, sa = ops.quant_rows_fp8(rand((M, K)))
    w8, sw = ops.quant_rows_fp8(rand((N, K), 0.05))
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    dbg = torch.zeros((256 * 8,), dtype=torch.int64, device=dev)
    fl = 2.0 * M * N * K / 1e12
    res = {}
    for tile in (4, 2002, 2011, 2001):
        res[tile] = min(res.get(tile, 1e9), timeit(lambda: ops.gemm_nt_fp8(a8, sa, w8, sw, None, n=N, k=K, epilogue=0, out=out, z=dbg, tile=tile)))
    def anatomy(tag):
        d = dbg.view(-1, 8).cpu().numpy().astype(np.float64)
        d = d[d[:, 3] > 0]
        steps = d[:, 3]
        per = d[:, 0] / steps
        print(f"   {tag}, per step (median over {len(d)} workgroups): loop {np.median(per):6.0f} cycles (2048 = matrix pipe), barrier 1 wait "
              f"{np.median(d[:, 1] / steps):5.0f}, barrier 2 (vmcnt + barrier) wait {np.median(d[:, 2] / steps):5.0f}; in-loop clock "
              f"{np.median(d[:, 0] / np.maximum(d[:, 4], 1)) * 0.1:5.2f} GHz; per tile ({K // 128} steps): epilogue "
              f"{np.median(d[:, 5] / steps) * K / 128:6.0f} cycles, epilogue end -> next K loop {np.median(d[:, 6] / steps) * K / 128:5.0f}, first step of a tile "
              f"{np.median(d[:, 7] / steps) * K / 128:5.0f}", flush=True)

    print(f"{name:12s} M={M} N={N} K={K}: product {res[4] * 1e6:7.1f} us {fl / res[4]:6.0f} TF/s | stamped {res[2001] * 1e6:7.1f} us | no DMA {res[2002] * 1e6:7.1f} us "
          f"{fl / res[2002]:6.0f} TF/s", flush=True)
    anatomy("stamped build, 256 workgroups")
    dbg.zero_()
    ops.gemm_nt_fp8(a8, sa, w8, sw, None, n=N, k=K, epilogue=0, out=out, z=dbg, tile=2011)
    anatomy("stamped build, 64 workgroups (a quarter of the chip: the same per-CU work, a quarter of the chip-wide traffic)")
