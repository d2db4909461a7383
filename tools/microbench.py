#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box (HIP-event timed, random data): GEMM shapes of the
cfg-3 towers, attention, norms.  Usage: python tools/microbench.py [gemm] [attn] [norm]

The `gemm` / `cold` / `ksweep` modes force launch forms through p2t_set_gemm_policy: run them against the LAB build
(P2T_HIP_LIB=tools/build/libp2t_lab.so, `make -C prot2text-v2-esm3_amd/csrc lab`); the product library only knows 0 and 9."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rand(shape, dtype=torch.bfloat16, scale=1.0):
    t = torch.empty(shape, dtype=dtype, device=dev)
    ops.fill_hash_(t, 1, f"mb{shape}", scale)
    return t


def bench_gemm():
    shapes = [  # name, M, N, K, epilogue
        ("esm qkv", 16384, 7680, 2560, 0), ("esm o", 16384, 2560, 2560, 2), ("esm fc1", 16384, 10240, 2560, 1),
        ("esm fc2", 16384, 2560, 10240, 2), ("llama qkv", 2048, 6144, 4096, 0), ("llama o", 2048, 4096, 4096, 2),
        ("llama gu", 2048, 28672, 4096, 3), ("llama down", 2048, 4096, 14336, 2), ("adapter fc1", 16384, 2048, 2560, 1),
        ("adapter fc2", 16384, 4096, 2048, 1), ("square 4k", 4096, 4096, 4096, 0), ("square 8k", 8192, 8192, 8192, 0),
    ]
    for name, M, N, K, epi in shapes:
        a, w = rand((M, K)), rand((N, K), scale=0.05)
        bias = None if epi == 3 else rand((N,), torch.float32, 0.1)
        out = torch.zeros((M, N), dtype=torch.float32, device=dev) if epi == 2 else None
        ws, ep = ops.gemm_fix_workspace(dev), [0]

        def with_tail():
            ep[0] += 1
            ops.gemm_nt(a, w, bias, epilogue=epi, out=out, use_mfma=1, fix_ws=ws, fix_epoch=ep[0])
        res = []
        for env, fn in (("2", lambda: ops.gemm_nt(a, w, bias, epilogue=epi, out=out, use_mfma=1)), ("2", with_tail),
                        ("4", with_tail), ("3", with_tail), (None, with_tail), ("7", with_tail), ("8", with_tail)):
            _lib.call("p2t_set_gemm_policy", int(env or 0))
            res.append(timeit(fn))
        _lib.call("p2t_set_gemm_policy", 0)
        fl = 2.0 * M * N * K / 1e9
        print(f"gemm {name:12s} M={M:6d} N={N:6d} K={K:6d} epi={epi}: TF/s per-tile {fl / res[0]:7.1f} | per-tile+splitK {fl / res[1]:7.1f} | "
              f"persistent {fl / res[2]:7.1f} | persistent+splitK {fl / res[3]:7.1f} | default {fl / res[4]:7.1f} ({res[4]:.3f} ms) | four-wave per-tile {fl / res[5]:7.1f} | four-wave persistent {fl / res[6]:7.1f}", flush=True)


def bench_cold():
    """Encoder GEMMs with their operands / residual stream COLD (a 768 MB fill between launches evicts the 256 MB Infinity
    Cache), as they are inside a step; each launch timed on its own with events."""
    shapes = [("esm qkv", 16384, 7680, 2560, 0), ("esm o", 16384, 2560, 2560, 2), ("esm fc1", 16384, 10240, 2560, 1),
              ("esm fc2", 16384, 2560, 10240, 2)]
    flush = torch.empty((768 << 20,), dtype=torch.uint8, device=dev)
    for name, M, N, K, epi in shapes:
        a, w = rand((M, K)), rand((N, K), scale=0.05)
        bias = rand((N,), torch.float32, 0.1)
        out = torch.zeros((M, N), dtype=torch.float32, device=dev) if epi == 2 else None
        ws, ep = ops.gemm_fix_workspace(dev), [0]
        fl = 2.0 * M * N * K / 1e9
        res = []
        for env in ("2", "4", "3", "5", "9", "7", "8", "10", "12"):
            _lib.call("p2t_set_gemm_policy", int(env or 0))
            tot = 0.0
            for i in range(6):
                flush.fill_(i)
                s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ep[0] += 1
                s0.record()
                ops.gemm_nt(a, w, bias, epilogue=epi, out=out, use_mfma=1, fix_ws=ws, fix_epoch=ep[0])
                e0.record()
                torch.cuda.synchronize()
                if i > 0:
                    tot += s0.elapsed_time(e0)
            res.append(tot / 5)
        _lib.call("p2t_set_gemm_policy", 0)
        print(f"cold {name:8s}: per-tile(+splitK) {res[0] * 1e3:7.1f} us {fl / res[0]:7.1f} TF/s | persistent {res[1] * 1e3:7.1f} us {fl / res[1]:7.1f} | "
              f"persistent+splitK {res[2] * 1e3:7.1f} us {fl / res[2]:7.1f} | persistent+half-tiles {res[3] * 1e3:7.1f} us {fl / res[3]:7.1f} | "
              f"eight-wave default {res[4] * 1e3:7.1f} us {fl / res[4]:7.1f} | four-wave per-tile {res[5] * 1e3:7.1f} us {fl / res[5]:7.1f} | four-wave persistent + split-K tail {res[6] * 1e3:7.1f} us {fl / res[6]:7.1f} | "
              f"four-wave persistent, whole tiles only {res[7] * 1e3:7.1f} us {fl / res[7]:7.1f} | same, other order {res[8] * 1e3:7.1f} us {fl / res[8]:7.1f}", flush=True)


def bench_ksweep():
    """Per-tile fixed cost vs steady-state rate: time = rounds * (a + b*K)."""
    for epi in (0, 1, 2):
        for M, N in ((16384, 10240), (16384, 2560)):
            for K in (256, 512, 1024, 2560, 5120, 10240):
                a, w = rand((M, K)), rand((N, K), scale=0.05)
                bias = rand((N,), torch.float32, 0.1)
                out = torch.zeros((M, N), dtype=torch.float32, device=dev) if epi == 2 else None
                ms = timeit(lambda: ops.gemm_nt(a, w, bias, epilogue=epi, out=out, use_mfma=1))
                rounds = (M // 256) * (N // 256) / 256.0
                print(f"ksweep epi={epi} M={M} N={N} K={K:6d}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s  "
                      f"per-round {ms * 1e3 / rounds:7.2f} us", flush=True)


def bench_attn():
    for name, B, T, nh, nkv, d, causal in [("esm3b", 16, 1024, 40, 40, 64, False), ("llama8b", 16, 128, 32, 8, 128, True),
                                            ("esm35m", 32, 512, 20, 20, 24, False)]:
        qkv = rand((B * T, (nh + 2 * nkv) * d))
        inv = torch.ones((d // 2,), dtype=torch.float32, device=dev)
        mask = torch.ones((B, T), dtype=torch.int64, device=dev)
        km, kv, _ = ops.mask_prepare(mask)
        q, k, v = ops.qkv_post(qkv, inv, B, T, nh, nkv, d, 1.0)
        ms_post = timeit(lambda: ops.qkv_post(qkv, inv, B, T, nh, nkv, d, 1.0))
        ms = timeit(lambda: ops.attention(q, k, v, km, kv, d, d ** -0.5, causal, use_mfma=1))
        ms2 = timeit(lambda: ops.attention(q, k, v, km, kv, d, 1.0, causal, use_mfma=1, log2_scores=True))     # timing only: q not re-scaled
        fl = 4.0 * B * nh * T * T * d * (0.5 if causal else 1.0)
        print(f"attn {name:8s} B={B} T={T} nh={nh}/{nkv} d={d}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF/s | log2-scores form {ms2:7.3f} ms "
              f"{fl / ms2 / 1e9:7.1f} TF/s | qkv_post {ms_post:7.3f} ms", flush=True)


def bench_norm():
    for rows, cols in ((16384, 2560), (2048, 4096)):
        x = rand((rows, cols), torch.float32)
        w, b = rand((cols,), torch.float32), rand((cols,), torch.float32)
        ms = timeit(lambda: ops.layernorm(x, w, b, 1e-5, torch.bfloat16))
        print(f"layernorm {rows}x{cols}: {ms * 1e3:7.1f} us  {rows * cols * 6 / ms / 1e6:7.1f} GB/s", flush=True)


def bench_blas():
    """Vendor-library reference point (NOT used by the product): torch.matmul (hipBLASLt / rocBLAS) on the encoder GEMM
    shapes, plain bf16 store, same cold-cache protocol as `cold`."""
    shapes = [("esm qkv", 16384, 7680, 2560), ("esm o", 16384, 2560, 2560), ("esm fc1", 16384, 10240, 2560),
              ("esm fc2", 16384, 2560, 10240), ("square", 8192, 8192, 8192)]
    flush = torch.empty((768 << 20,), dtype=torch.uint8, device=dev)
    for name, M, N, K in shapes:
        a, w = rand((M, K)), rand((N, K), scale=0.05)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        fl = 2.0 * M * N * K / 1e9
        res = []
        for which in ("blas", "ours"):
            tot = 0.0
            for i in range(6):
                flush.fill_(i)
                s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                if which == "blas":
                    torch.matmul(a, w.t(), out=out)
                else:
                    ops.gemm_nt(a, w, None, epilogue=0, out=out, use_mfma=1)
                e0.record()
                torch.cuda.synchronize()
                if i > 0:
                    tot += s0.elapsed_time(e0)
            res.append(tot / 5)
        print(f"blas {name:8s} M={M} N={N} K={K}: torch.matmul {res[0] * 1e3:7.1f} us {fl / res[0]:7.1f} TF/s | p2t gemm_nt (store epilogue) "
              f"{res[1] * 1e3:7.1f} us {fl / res[1]:7.1f} TF/s", flush=True)


def bench_fp8():
    """fp8 MFMA GEMM (cfg5) on the tower shapes, operands cold (a 768 MB fill between launches), per tile height; beside it
    the bf16 kernel's default policy on the same shape, and the cost of the quantise pass that feeds FFN-down / o-proj."""
    shapes = [("esm qkv", 16384, 7680, 2560, 0), ("esm o", 16384, 2560, 2560, 2), ("esm fc1", 16384, 10240, 2560, 1),
              ("esm fc2", 16384, 2560, 10240, 2), ("esm qkv b64", 65536, 7680, 2560, 0), ("esm fc2 b64", 65536, 2560, 10240, 2),
              ("llama qkv", 8192, 6144, 4096, 0), ("llama gu", 8192, 28672, 4096, 3), ("llama down", 8192, 4096, 14336, 2)]
    flush = torch.empty((768 << 20,), dtype=torch.uint8, device=dev)
    for name, M, N, K, epi in shapes:
        a, w = rand((M, K)), rand((N, K), scale=0.05)
        a8, sa = ops.quant_rows_fp8(a)
        w8, sw = ops.quant_rows_fp8(w)
        bias = None if epi == 3 else rand((N,), torch.float32, 0.1)
        out = torch.zeros((M, N), dtype=torch.float32, device=dev) if epi == 2 else None
        fl = 2.0 * M * N * K / 1e9
        res = []
        for which in (256, 128, "bf16", "quant", 0):
            if which == 0 and epi == 1:
                res.append(float("nan"))                     # plain GELU has no four-wave form (the towers use GELU -> e4m3)
                continue
            tot = 0.0
            for i in range(5):
                flush.fill_(i)
                s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                if which == "bf16":
                    ops.gemm_nt(a, w, bias, epilogue=epi, out=out, use_mfma=1)
                elif which == "quant":
                    ops.quant_rows_fp8(a)
                else:
                    ops.gemm_nt_fp8(a8, sa, w8, sw, bias, n=N, k=K, epilogue=epi, out=out, tile=which)
                e0.record()
                torch.cuda.synchronize()
                if i > 0:
                    tot += s0.elapsed_time(e0)
            res.append(tot / 4)
        print(f"fp8 {name:12s} M={M:6d} N={N:6d} K={K:6d} epi={epi}: 256-row {res[0] * 1e3:7.1f} us {fl / res[0]:7.1f} TF/s | 128-row "
              f"{res[1] * 1e3:7.1f} us {fl / res[1]:7.1f} | bf16 kernel {res[2] * 1e3:7.1f} us {fl / res[2]:7.1f} | quantise A [{M}x{K}] "
              f"{res[3] * 1e3:6.1f} us {M * K * 3 / res[3] / 1e6:6.0f} GB/s | four-wave persistent {res[4] * 1e3:7.1f} us {fl / res[4]:7.1f}", flush=True)


def bench_skinny():
    """The decode step's weight-streaming GEMM (p2t_gemm_nt_skinny) on the Llama-3.1-8B projections: every call reads another copy
    of the weights (a ring of copies > 1 GB, so neither the L2s nor the 256 MB memory-side cache serve them) -> GB/s of W."""
    from p2t_hip.ops import ptr, stream
    shapes = [("o-proj", 4096, 4096, _lib.EPI_RESID), ("qkv", 6144, 4096, _lib.EPI_STORE_F32), ("gate/up", 28672, 4096, _lib.EPI_SWIGLU),
              ("down", 4096, 14336, _lib.EPI_RESID), ("lm head", 128256, 4096, _lib.EPI_STORE)]
    Ms = [int(a[1:]) for a in sys.argv if a.startswith("m") and a[1:].isdigit()] or [8, 32]
    for M in Ms:
        for name, N, K, epi in shapes:
            copies = max(2, min(40, int(1.3e9 // (N * K * 2)) + 1))
            ws = [rand((N, K), scale=0.02) for _ in range(copies)]
            x = rand((M, K))
            if epi == _lib.EPI_SWIGLU:
                out, ldc, odt = torch.zeros((M, N // 2), dtype=torch.bfloat16, device=dev), N // 2, _lib.BF16
            elif epi == _lib.EPI_STORE:
                out, ldc, odt = torch.zeros((M, N), dtype=torch.bfloat16, device=dev), N, _lib.BF16
            else:
                out, ldc, odt = torch.zeros((M, N), dtype=torch.float32, device=dev), N, _lib.F32
            from p2t_hip.generation import preshuffle
            res = []
            for pre in (0, 1):
                if pre:
                    ws = [preshuffle(w, N) for w in ws]
                i = [0]

                def run():
                    w = ws[i[0] % copies]
                    i[0] += 1
                    _lib.call("p2t_gemm_nt_skinny", ptr(x), K, ptr(w), K, pre, ptr(out), ldc, M, N, K, odt, epi, stream())
                ms = timeit(run, iters=max(20, copies), warm=copies)
                res.append(f"{'stream copy' if pre else 'row-major W'} {ms * 1e3:7.1f} us {N * K * 2 / ms / 1e6:6.0f} GB/s")
            print(f"skinny M={M:2d} {name:8s} N={N:6d} K={K:5d}: " + " | ".join(res), flush=True)
            del ws


def bench_fp8abl():
    """Lab ablations of the fp8 K loop (per-tile kernel, garbage results): full / no DMA / no fragment reloads / neither."""
    for name, M, N, K in (("esm qkv b64", 65536, 7680, 2560), ("esm fc2 b64", 65536, 2560, 10240), ("square 8k", 8192, 8192, 8192)):
        a, w = rand((M, K)), rand((N, K), scale=0.05)
        a8, sa = ops.quant_rows_fp8(a)
        w8, sw = ops.quant_rows_fp8(w)
        fl = 2.0 * M * N * K / 1e9
        res = [timeit(lambda t=t: ops.gemm_nt_fp8(a8, sa, w8, sw, None, n=N, k=K, epilogue=0, tile=t), iters=5, warm=2) for t in (256, 1001, 1002, 1003)]
        print(f"fp8abl {name:12s}: full {fl / res[0]:7.1f} TF/s | no DMA {fl / res[1]:7.1f} | no LDS reads {fl / res[2]:7.1f} | neither {fl / res[3]:7.1f}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "norm"]
    print(torch.cuda.get_device_name(0), flush=True)
    if "gemm" in which:
        bench_gemm()
    if "cold" in which:
        bench_cold()
    if "ksweep" in which:
        bench_ksweep()
    if "attn" in which:
        bench_attn()
    if "norm" in which:
        bench_norm()
    if "blas" in which:
        bench_blas()
    if "fp8" in which:
        bench_fp8()
    if "skinny" in which:
        bench_skinny()
    if "fp8abl" in which:
        bench_fp8abl()
