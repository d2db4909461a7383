"""Deterministic synthetic tensors: a counter-hash generator with a numpy and a HIP side.

There are no checkpoints on disk and no network, so weights and
inputs are *generated*.  Every element is a pure function of (seed, tensor name, flat index):

    h   = splitmix64(index + GOLDEN * (fnv1a64(name) | 1)  ^  seed * SEEDMUL)
    u24 = h >> 40                                   # 24 uniform bits
    v   = float32(int32(u24) - 2**23) * float32(scale / 2**23)     # uniform in [-scale, scale)
    out = v + offset                                # one more fp32 rounding (no FMA)

The integer -> float conversion is exact and the value needs a single fp32 multiply, so the numpy
implementation here and the HIP kernel `p2t_fill_hash` (csrc/fill.hip) agree bit for bit.  That
lets the CPU oracle, the torch reference used to make golden vectors, and the GPU materialise the
same multi-GB models without shipping them.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_SEEDMUL = np.uint64(0xD1B54A32D192ED03)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_MASK64 = (1 << 64) - 1


def fnv1a64(name: str) -> int:
    """64-bit FNV-1a of the utf-8 tensor name (the per-tensor stream id)."""
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK64
    return h


def stream_key(seed: int, name: str) -> tuple[int, int]:
    """(add, xor) 64-bit constants handed to both generators for one tensor."""
    tid = fnv1a64(name) | 1
    add = (int(_GOLDEN) * tid) & _MASK64
    xor = (int(seed) * int(_SEEDMUL)) & _MASK64
    return add, xor


def _mix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(30))
        x = x * _M1
        x = x ^ (x >> np.uint64(27))
        x = x * _M2
        x = x ^ (x >> np.uint64(31))
    return x


def hash_u24(seed: int, name: str, start: int, count: int) -> np.ndarray:
    add, xor = stream_key(seed, name)
    idx = np.arange(start, start + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (idx + np.uint64(add)) ^ np.uint64(xor)
    return (_mix(x) >> np.uint64(40)).astype(np.int64)


def scale_f32(scale: float) -> np.float32:
    return np.float32(float(scale) / 8388608.0)


_CHUNK = 1 << 18          # 256 Ki elements: the uint64 temporaries stay cache-resident


def _fill_chunk(out, lo, hi, start, add, xor, s23, offset):
    """out[lo:hi] = generator values for stream indices start+lo .. start+hi (in-place uint64 math)."""
    with np.errstate(over="ignore"):
        x = np.arange(start + lo, start + hi, dtype=np.uint64)
        x += np.uint64(add)
        x ^= np.uint64(xor)
        t = x >> np.uint64(30); x ^= t; x *= _M1
        np.right_shift(x, np.uint64(27), out=t); x ^= t; x *= _M2
        np.right_shift(x, np.uint64(31), out=t); x ^= t
        x >>= np.uint64(40)
    v = x.astype(np.int32)
    v -= np.int32(8388608)
    f = v.astype(np.float32)
    f *= s23
    if offset != 0.0:
        f += np.float32(offset)
    out[lo:hi] = f


def uniform_f32(seed: int, name: str, shape, scale: float, offset: float = 0.0,
                start: int = 0) -> np.ndarray:
    """fp32 tensor, uniform in [offset-scale, offset+scale); bit-identical to the HIP generator.
    Chunked and multi-threaded (numpy releases the GIL) so GB-sized towers materialise in seconds."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    add, xor = stream_key(seed, name)
    s23 = scale_f32(scale)
    out = np.empty((n,), dtype=np.float32)
    spans = [(lo, min(lo + _CHUNK, n)) for lo in range(0, n, _CHUNK)]
    if len(spans) <= 2:
        for lo, hi in spans:
            _fill_chunk(out, lo, hi, start, add, xor, s23, offset)
    else:
        import os
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(lambda sp: _fill_chunk(out, sp[0], sp[1], start, add, xor, s23, offset), spans))
    return out.reshape(shape)


def uniform_rows_f32(seed: int, name: str, rows: np.ndarray, ncols: int, scale: float,
                     offset: float = 0.0) -> np.ndarray:
    """Selected rows of a (nrows, ncols) tensor without generating the rest (embedding tables)."""
    rows = np.asarray(rows, dtype=np.int64).reshape(-1)
    out = np.empty((rows.size, ncols), dtype=np.float32)
    for i, r in enumerate(rows):
        out[i] = uniform_f32(seed, name, (ncols,), scale, offset, start=int(r) * ncols)
    return out


def randint(seed: int, name: str, shape, low: int, high: int) -> np.ndarray:
    """int64 tensor uniform in [low, high) (multiply-shift on the 24 hash bits)."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    u = hash_u24(seed, name, 0, n)
    return (low + ((u * (high - low)) >> 24)).astype(np.int64).reshape(shape)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 to the nearest-even bf16 and return it widened back to fp32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    out = r.astype(np.uint32).view(np.float32).reshape(x.shape)
    nan = np.isnan(x)
    if nan.any():
        out = np.where(nan, x, out)
    return out


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """Nearest-even bf16 bit patterns (uint16) of an fp32 array."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)
    return r.astype(np.uint16).reshape(x.shape)


# ----------------------------------------------------------------------------------------------
# Synthetic batches following the reference's batch contract (dataset/dataloader.py:98-197):
# right-padded protein ids  <cls> residues <eos> <pad>...,  right-padded description ids.
# ----------------------------------------------------------------------------------------------
ESM_CLS, ESM_PAD, ESM_EOS, ESM_MASK = 0, 1, 2, 32


def protein_batch(seed: int, batch: int, seq_len: int, lengths=None):
    """(ids, mask) int64 (B, T).  `lengths` (incl. <cls>/<eos>) default to full length."""
    ids = randint(seed, "protein_input_ids", (batch, seq_len), 4, 24)
    mask = np.ones((batch, seq_len), dtype=np.int64)
    lengths = [seq_len] * batch if lengths is None else list(lengths)
    for b, n in enumerate(lengths):
        n = max(2, min(int(n), seq_len))
        ids[b, 0] = ESM_CLS
        ids[b, n - 1] = ESM_EOS
        ids[b, n:] = ESM_PAD
        mask[b, n:] = 0
    return ids, mask


def text_batch(seed: int, batch: int, seq_len: int, id_high: int = 128000, lengths=None,
               pad_id: int = 128002, eos_id: int = 128009):
    """(ids, mask) int64 (B, T): random ids in [0, id_high), last valid token = eos_id, then pad_id
    (Llama-3 defaults: <|eot_id|> = 128009, <|reserved_special_token_0|> = 128002, README.md:134)."""
    ids = randint(seed, "description_input_ids", (batch, seq_len), 0, id_high)
    mask = np.ones((batch, seq_len), dtype=np.int64)
    lengths = [seq_len] * batch if lengths is None else list(lengths)
    for b, n in enumerate(lengths):
        n = max(1, min(int(n), seq_len))
        ids[b, n - 1] = eos_id
        ids[b, n:] = pad_id
        mask[b, n:] = 0
    return ids, mask
