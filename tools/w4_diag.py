"""Tile anatomy of the four-wave bf16 GEMM (lab build: P2T_HIP_LIB=tools/build/libp2t_lab.so): cycles in the K loop, in the
epilogue and between the end of the epilogue and the next K loop, per tile (s_memtime stamps, median over the workgroups).
A 64-deep stage is 128 MFMAs of 16 cycles = 2 048 matrix-pipe cycles.   python3 tools/w4_diag.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("P2T_HIP_LIB", os.path.join(ROOT, "tools", "build", "libp2t_lab.so"))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def rand(shape, scale=1.0, dtype=torch.bfloat16):
    t = torch.empty(shape, dtype=dtype, device=dev)
    ops.fill_hash_(t, 7, "w4diag" + str(shape), scale)
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


# (z is the stamp buffer: only epilogues that never touch z -- plain store and fp32 residual read-modify-write)
def report(name, M, N, K, t_plain, t_diag, dbg, fl):
    d = dbg.view(-1, 8).cpu().numpy().astype(np.float64)
    d = d[d[:, 3] > 0]
    tiles = d[:, 3]
    if len(d) == 0:
        print(f"{name}: {t_plain * 1e6:7.1f} us {fl / t_plain:6.0f} TF/s (no stamps: another kernel form ran)")
        return
    print(f"{name:26s} M={M} N={N} K={K}: {t_plain * 1e6:7.1f} us {fl / t_plain:6.0f} TF/s (stamped {t_diag * 1e6:7.1f} us); per tile, median over {len(d)} "
          f"workgroups x {np.median(tiles):.0f} tiles: K loop {np.median(d[:, 0] / tiles):7.0f} cycles ({K // 64} stages x {np.median(d[:, 0] / tiles) / (K // 64):5.0f}; 2048 = "
          f"matrix pipe), epilogue {np.median(d[:, 1] / tiles):6.0f}, epilogue end -> next K loop {np.median(d[:, 2] / np.maximum(tiles - 1, 1)):6.0f}; in-loop clock "
          f"{np.median(d[:, 0] / np.maximum(d[:, 4], 1)) * 0.1:4.2f} GHz", flush=True)


shapes = [("esm qkv (store)", 16384, 7680, 2560, _lib.EPI_STORE), ("esm fc1 (store)", 16384, 10240, 2560, _lib.EPI_STORE),
          ("esm o (resid)", 16384, 2560, 2560, _lib.EPI_RESID), ("esm fc2 (resid)", 16384, 2560, 10240, _lib.EPI_RESID),
          ("esm o b64 (resid)", 65536, 2560, 2560, _lib.EPI_RESID)]
for name, M, N, K, epi in shapes:
    assert epi in (_lib.EPI_STORE, _lib.EPI_RESID)
    a, w = rand((M, K)), rand((N, K), 0.05)
    bias = rand((N,), 0.1, torch.float32)
    out = torch.zeros((M, N), dtype=torch.float32 if epi == _lib.EPI_RESID else torch.bfloat16, device=dev)
    dbg = torch.zeros((256 * 8,), dtype=torch.int64, device=dev)
    fl = 2.0 * M * N * K / 1e12
    t_plain = timeit(lambda: ops.gemm_nt(a, w, bias, epilogue=epi, out=out, use_mfma=1))
    t_diag = timeit(lambda: ops.gemm_nt(a, w, bias, epilogue=epi, out=out, use_mfma=1, z=dbg))
    report(name, M, N, K, t_plain, t_diag, dbg, fl)
    _lib.call("p2t_set_gemm_policy", 1064)       # a quarter of the chip: the same per-CU work, a quarter of the chip-wide traffic
    dbg.zero_()
    t_q = timeit(lambda: ops.gemm_nt(a, w, bias, epilogue=epi, out=out, use_mfma=1, z=dbg), iters=5, warm=2)
    report(name + " 64 CUs", M, N, K, t_q, t_q, dbg, fl)
    _lib.call("p2t_set_gemm_policy", 1000)

