#!/usr/bin/env python3
"""Driver for tools/gemm_lab.hip (kernel-structure A/B tests on the GPU box; not product code).
Variants are timed in interleaved rounds inside one process (min and median reported)."""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import ops  # noqa: E402

lab = C.CDLL(os.path.join(ROOT, "tools", "build", "libgemm_lab.so"))
lab.lab_gemm.restype = C.c_float
lab.lab_gemm.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")


def rand(shape, dtype=torch.bfloat16, scale=1.0, name="x"):
    t = torch.empty(shape, dtype=dtype, device=dev)
    ops.fill_hash_(t, 1, f"lab{name}{shape}", scale)
    return t


def main():
    variants = [(0, 0), (2, 0), (3, 0), (3, 8), (3, 1), (3, 9)]
    args = [a for a in sys.argv[1:] if "," in a]
    if args:
        variants = [tuple(int(x) for x in v.split(",")) for v in args]
    rounds = 5
    shapes = [("esm qkv", 16384, 7680, 2560), ("esm o", 16384, 2560, 2560), ("esm fc1", 16384, 10240, 2560),
              ("esm fc2", 16384, 2560, 10240), ("llama gu", 2048, 28672, 4096), ("sq 8k", 8192, 8192, 8192)]
    s = torch.cuda.current_stream().cuda_stream
    print("variant vVaA: ABL bits 1=no epilogue 2=no global loads 4=no MFMA 8=interleaved schedule; TF/s = min-time (median)", flush=True)
    for name, M, N, K in shapes:
        a, w = rand((M, K), name="a"), rand((N, K), scale=0.05, name="w")
        bias = rand((N,), torch.float32, 0.1, "b")
        ref = ops.gemm_nt(a, w, bias, use_mfma=1)
        times = {v: [] for v in variants}
        bad = {}
        for r in range(rounds):
            for var, abl in variants:
                c = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
                ms = lab.lab_gemm(var, abl, a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N, M, N, K, 5, s)
                torch.cuda.synchronize()
                times[(var, abl)].append(ms)
                if r == 0 and (abl & 7) == 0:
                    err = (c.float() - ref[:, :N].float()).abs().max().item()
                    if not err < 0.02 * ref.float().abs().max().item():
                        bad[(var, abl)] = err
        line = f"{name:9s} M={M} N={N} K={K}:"
        for v in variants:
            fl = 2.0 * M * N * K / 1e9
            line += f"  v{v[0]}a{v[1]} {fl / min(times[v]):5.0f} ({fl / statistics.median(times[v]):5.0f})" + (f" WRONG {bad[v]:.3g}" if v in bad else "")
        print(line, flush=True)


if __name__ == "__main__":
    main()
