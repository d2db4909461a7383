"""Forced launch forms of the bf16 MFMA GEMM -- cases that need the LAB build of the library (csrc/Makefile `make lab`,
-DP2T_LAB -> tools/build/libp2t_lab.so; tools/lab/gemm_forms_lab.h).  NOT collected by the plain `pytest tests/` run (the file name
does not match test_*.py): tests/test_gpu_lab_forms.py runs it in a child process with P2T_HIP_LIB pointing at the lab build, so
the product library in the parent stays the product library.

Each case forces one kernel form (p2t_set_gemm_policy: 2 per-tile, 3 / 4 / 5 eight-wave persistent forms, 7 four-wave per tile,
8 / 10 / 12 four-wave persistent forms, 6 the 64-deep skeleton) on shapes the default policy would hand to another form, and
compares with the exact fp32-FMA kernel on the same bf16 operands."""
import numpy as np
import pytest
import torch

from gpu_util import bf16r, dev, rel, rnd, to_dev, to_np
from test_gpu_kernels import EPI_GELU, EPI_RESID, EPI_STORE, _gemm_ref, _no_timeout, gemm_policy, ops  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def test_this_is_the_lab_build():
    from p2t_hip import _lib
    assert _lib.call("p2t_is_lab_build") == 1, f"{_lib.LIB_PATH} is not a lab build"


@pytest.mark.parametrize("shape", [(16384, 2560, 4096), (4096, 5376, 4096), (2048, 4096, 8192)])
@pytest.mark.parametrize("path", ["persistent", "per_tile"])
@pytest.mark.parametrize("epi", [EPI_STORE, EPI_RESID, EPI_GELU])
def test_gemm_mfma_splitk_tail(ops, epi, path, shape, gemm_policy):
    """640 tiles = 2.5 rounds of the 256 CUs: with a fix-up workspace the last 128 tiles run as two concurrent K
    halves (producer slab -> consumer epilogue).  Must equal the plain kernel bit for bit in structure-independent
    terms (same fp32 sums up to the order of the two K halves) and the oracle; repeated launches reuse the flags.
    Both kernels that implement it: the persistent one (default for this shape) and the one-block-per-tile one."""
    # 2: per-tile kernels only; 3: persistent kernel with the fix-up whenever possible (the default policy only uses it
    # from K = 6144 up, where it pays)
    gemm_policy(2 if path == "per_tile" else 3)
    M, N, K = shape                 # 640 tiles = 2.5 rounds / 336 tiles = 1 round + 80 / 128 tiles, all split (long K)
    n_tail = ((M // 256) * (N // 256)) % 256
    a, w = bf16r(rnd(12, "s.a", (M, K), 1.0)), bf16r(rnd(12, "s.w", (N, K), 0.3))
    bias = rnd(12, "s.b", (N,), 0.3)
    resid = rnd(12, "s.r", (M, N), 1.0)
    ad, wd, bd = to_dev(a, torch.bfloat16), to_dev(w, torch.bfloat16), to_dev(bias)
    ref = _gemm_ref(a, w, bias, epi, resid, None)
    ws = ops.gemm_fix_workspace(dev())
    for epoch in (1, 2, 3):
        out = to_dev(resid) if epi == EPI_RESID else None
        got = to_np(ops.gemm_nt(ad, wd, bd, epilogue=epi, out=out, use_mfma=1, fix_ws=ws, fix_epoch=epoch))
        assert rel(got[:, :N], ref) < (3e-6 if epi == EPI_RESID else 3e-3), epoch
    flags = ws[:2048].view(torch.int32).cpu().numpy()
    assert (flags[:n_tail] == 3).all() and not flags[n_tail:128].any() and _no_timeout()   # every tail tile published; no time-out
    gemm_policy(2)                                                # plain per-tile kernel, no workspace
    plain = to_np(ops.gemm_nt(ad, wd, bd, epilogue=epi, out=to_dev(resid) if epi == EPI_RESID else None, use_mfma=1))
    assert rel(got[:, :N], plain[:, :N]) < (1e-6 if epi == EPI_RESID else 2e-3)


@pytest.mark.parametrize("tile", ["3", "4", "5", "6", "7", "8", "10", "12"])
def test_gemm_persistent_forms_fuzz(ops, tile, gemm_policy):
    """Random whole-tile shapes through the persistent kernel (default policy / split-K fix-up forced / no fix-up /
    128-row halves for the partial round) against
    the exact fp32-FMA kernel on the same bf16 operands: any stale LDS read or mis-counted wait shows up as a wrong tile."""
    gemm_policy(int(tile))
    rng = np.random.default_rng(int(tile) + 7)
    ws = ops.gemm_fix_workspace(dev())
    epoch = 0
    for _ in range(10):
        tiles = int(rng.integers(256, 700))
        tm = int(rng.choice([d for d in range(4, 65) if tiles // d >= 4]))
        tn = max(4, tiles // tm)
        M, N, K = 256 * tm, 256 * tn, 128 * int(rng.integers(3, 21))
        a = torch.empty((M, K), dtype=torch.bfloat16, device=dev())
        w = torch.empty((N, K), dtype=torch.bfloat16, device=dev())
        ops.fill_hash_(a, 3, f"fz.a{M}x{K}", 1.0)
        ops.fill_hash_(w, 3, f"fz.w{N}x{K}", 0.5)
        b = to_dev(rnd(3, "fz.b", (N,), 0.3))
        for epi in (EPI_STORE, EPI_RESID):
            out0 = torch.ones((M, N), dtype=torch.float32, device=dev()) if epi == EPI_RESID else None
            out1 = torch.ones((M, N), dtype=torch.float32, device=dev()) if epi == EPI_RESID else None
            epoch += 1
            got = ops.gemm_nt(a, w, b, epilogue=epi, out=out1, out_dtype=torch.float32, use_mfma=1, fix_ws=ws, fix_epoch=epoch)
            ref = ops.gemm_nt(a, w, b, epilogue=epi, out=out0, out_dtype=torch.float32, use_mfma=0)
            err = float((got[:, :N] - ref[:, :N]).abs().max() / ref[:, :N].abs().max())
            assert err < 2e-5, (M, N, K, epi, err)
    assert _no_timeout()                     # no split-K consumer timed out


@pytest.mark.parametrize("epi", [EPI_STORE, EPI_GELU, EPI_RESID])
@pytest.mark.parametrize("shape", [(300, 320, 128), (1000, 96, 256), (513, 1184, 384), (256, 256, 1024), (700, 2560, 2560)])
def test_gemm_four_wave_form_edges(ops, epi, shape, gemm_policy):
    """gemm_w4.hip (policy 7) on shapes with edge tiles in M and N and 4 .. 80 stages, against the exact fp32-FMA kernel."""
    M, N, K = shape
    a = to_dev(bf16r(rnd(61, "w4.a", (M, K), 1.0)), torch.bfloat16)
    w = to_dev(bf16r(rnd(61, "w4.w", (N, K), 0.5)), torch.bfloat16)
    b = to_dev(rnd(61, "w4.b", (N,), 0.3))
    outs = []
    for mf in (1, 0):
        gemm_policy(7 if mf else 0)
        out = torch.ones((M, N), dtype=torch.float32, device=dev()) if epi == EPI_RESID else None
        outs.append(ops.gemm_nt(a, w, b, epilogue=epi, out=out, out_dtype=torch.float32 if epi != EPI_GELU else torch.bfloat16, use_mfma=mf))
    got, ref = outs[0][:, :N].float(), outs[1][:, :N].float()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < (1e-2 if epi == EPI_GELU else 2e-5), (shape, epi, err)
    if outs[0].shape[1] > N:
        assert not bool(outs[0][:, N:].any())


