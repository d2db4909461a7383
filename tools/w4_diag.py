"""Tile anatomy of the four-wave bf16 GEMM (lab build: P2T_HIP_LIB=tools/build/libp2t_lab.so): cycles in the K loop, in the
epilogue and between the end of the epilogue and the next K loop, per tile (s_memtime stamps, median over the workgroups), for every
epilogue type of the step (QKV + RoPE, GELU, SwiGLU through the lab build's global stamp pointer; store and residual through ep.z).
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
def stamped_any(name, fn, K, M, N):
    """Epilogues that use (or have no) z: stamps through the lab build's global stamp pointer."""
    import ctypes
    _lib.lib.p2t_lab_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    _lib.lib.p2t_lab_set_stamp_buffer.restype = ctypes.c_int
    dbg = torch.zeros((256 * 8,), dtype=torch.int64, device=dev)
    t_plain = timeit(fn)
    assert _lib.lib.p2t_lab_set_stamp_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
    t_diag = timeit(fn)
    assert _lib.lib.p2t_lab_set_stamp_buffer(ctypes.c_void_p(0)) == 0
    report(name, M, N, K, t_plain, t_diag, dbg, 2.0 * M * N * K / 1e12)


M_, H_, F_ = 16384, 2560, 10240
a_ = rand((M_, H_))
wq_, bq_ = rand((3 * H_, H_), 0.05), rand((3 * H_,), 0.1, torch.float32)
inv_ = torch.ones((32,), dtype=torch.float32, device=dev)
stamped_any("esm qkv + rope", lambda: ops.gemm_qkv_rope(a_, wq_, bq_, inv_, 1024, 40, 40, 64, 0.125), H_, M_, 3 * H_)
w1_, b1_ = rand((F_, H_), 0.05), rand((F_,), 0.1, torch.float32)
o1_ = torch.empty((M_, F_), dtype=torch.bfloat16, device=dev)
stamped_any("esm fc1 (gelu)", lambda: ops.gemm_nt(a_, w1_, b1_, epilogue=_lib.EPI_GELU, out=o1_, use_mfma=1), H_, M_, F_)
at_, wg_ = rand((2048, 4096)), rand((28672, 4096), 0.05)
stamped_any("llama gate/up (swiglu)", lambda: ops.gemm_nt(at_, wg_, None, epilogue=_lib.EPI_SWIGLU, use_mfma=1), 4096, 2048, 28672)
wq2_ = rand((6144, 4096), 0.05)
inv2_ = torch.ones((64,), dtype=torch.float32, device=dev)
del a_, w1_, o1_, wg_
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

