"""Helpers for the -m gpu tests: numpy <-> device tensors, bf16 rounding, model construction."""
import numpy as np
import torch

from p2t_hip import specs, synth


def dev():
    return torch.device("cuda:0")


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t.to(dtype) if dtype is not None else t


def to_np(t):
    return t.detach().float().cpu().numpy() if t.dtype in (torch.bfloat16, torch.float16) else t.detach().cpu().numpy()


def bf16r(a):
    return synth.bf16_round(np.asarray(a, dtype=np.float32))


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


_OBS_PATH = None
_TOL_TABLE = None


def observe(name, value, tol, what="rel"):
    """Assert `value < tolerance` AND append the observed error to gpurun_out/observed_errors.jsonl (merged back from the GPU
    box), so every bf16 / fp8 tolerance in the suite can be audited against what was actually measured.
    The tolerance enforced is the one in tests/tolerances.json when `name` is listed there (tools/update_tolerances.py writes
    it from a previous full run: 2 x the largest observed value, never above the inline `tol`), else the inline `tol`."""
    import json
    import os
    global _OBS_PATH, _TOL_TABLE
    if _TOL_TABLE is None:
        tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tolerances.json")
        # P2T_TOL_TABLE=0: inline tolerances only -- the run that re-measures after a numerics change, before
        # tools/update_tolerances.py rewrites the table from it
        _TOL_TABLE = json.load(open(tp)) if os.path.exists(tp) and os.environ.get("P2T_TOL_TABLE", "1") != "0" else {}
    inline_tol = float(tol)
    if name in _TOL_TABLE:
        tol = min(inline_tol, float(_TOL_TABLE[name]["tol"]))
    if _OBS_PATH is None:
        root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        d = os.path.join(root, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        _OBS_PATH = os.path.join(d, "observed_errors.jsonl")
    value = float(value)
    with open(_OBS_PATH, "a") as f:
        f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "name": name, "kind": what,
                            "observed": value, "tol": float(tol), "inline_tol": inline_tol}) + "\n")
    assert value < tol, f"{name}: observed {what} error {value:.3e} >= tolerance {tol:.3e}"
    return value


def rnd(seed, name, shape, scale=1.0, offset=0.0):
    return synth.uniform_f32(seed, name, shape, scale, offset)


def build_model(esm, llama, ad, dtype, seed=0, adapter_dtype=None):
    from p2t_hip import Esm2LlamaInstructForCausalLM
    return Esm2LlamaInstructForCausalLM.from_specs(esm, llama, ad, dtype=dtype, device=dev(), seed=seed,
                                                   adapter_dtype=adapter_dtype)
