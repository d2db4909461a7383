"""Shared test helpers: rebuild a golden case's specs, inputs and synthetic weights."""
import numpy as np

from p2t_hip import specs, synth


def case_setup(meta):
    esm = specs.EsmSpec(**meta["esm"])
    llama = specs.LlamaSpec(**meta["llama"])
    ad = specs.AdapterSpec(**meta["adapter"])
    pid, pmask = synth.protein_batch(meta["seed_in"], meta["B"], meta["T_p"], meta["p_lens"])
    tid, tmask = synth.text_batch(meta["seed_in"], meta["B"], meta["T_t"], meta["id_high"], meta["t_lens"],
                                  meta["pad_id"], meta["eos_id"])
    return esm, llama, ad, pid, pmask, tid, tmask


from oracle.weights import LazyWeights, model_weights  # noqa: E402,F401  (kept importable from helpers)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
