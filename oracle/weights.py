"""Synthetic weights for the oracle -- TEST INFRASTRUCTURE (tests/, smoke(), bench.py's cpu_baseline leg only).

dict-like state dicts generated on first access from the repo's counter-hash generator (p2t_hip.synth), the same
values the HIP fill kernel writes into the GPU model for the same seed."""
import numpy as np

from p2t_hip import specs, synth


class LazyWeights:
    """dict-like synthetic state dict generated on first access (keeps cfg-1's 1B-parameter text
    tower out of memory until a layer is needed); embedding tables are gathered row-wise."""

    def __init__(self, tensors, seed=0, cache=True):
        self.table = {name: (shape, scale, offset) for name, shape, scale, offset in tensors}
        self.seed, self.cache, self._c = seed, cache, {}

    def __contains__(self, k):
        return k in self.table

    def __getitem__(self, k):
        if k in self._c:
            return self._c[k]
        shape, scale, offset = self.table[k]
        if k.endswith("embed_tokens.weight") or k.endswith("word_embeddings.weight"):
            v = _RowTable(self.seed, k, shape, scale, offset)
        elif scale == 0.0:
            v = np.full(shape, offset, dtype=np.float32)
        else:
            v = synth.uniform_f32(self.seed, k, shape, scale, offset)
        if self.cache:
            self._c[k] = v
        return v


class _RowTable:
    def __init__(self, seed, name, shape, scale, offset):
        self.seed, self.name, self.shape, self.scale, self.offset = seed, name, shape, scale, offset

    def __getitem__(self, ids):
        ids = np.asarray(ids)
        uniq, inv = np.unique(ids.reshape(-1), return_inverse=True)
        rows = synth.uniform_rows_f32(self.seed, self.name, uniq, self.shape[1], self.scale, self.offset)
        return rows[inv].reshape(*ids.shape, self.shape[1])


def model_weights(esm, llama, ad, seed=0, cache=True, lm_head=False):
    """lm_head=True adds `llama_decoder.lm_head.weight` (untied decoders; the stage-2 tests need the LM head)."""
    import itertools
    return LazyWeights(itertools.chain(specs.esm_tensors(esm, "esm_encoder."),
                                       specs.adapter_tensors(ad, "adapter."),
                                       specs.llama_tensors(llama, "llama_decoder.", lm_head=lm_head)), seed, cache)


