"""Host-side pins of the fp8 oracle pieces (no GPU): the numpy e4m3fn rounding against torch's CPU float8_e4m3fn cast, the
E8M0 row-scale rule, and the row quantiser's invariants."""
import numpy as np
import torch

from oracle import p2t_oracle as O


def test_e4m3_round_matches_torch_float8_e4m3fn():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-448, 448, 20000), rng.normal(0, 1, 20000), rng.normal(0, 2 ** -7, 20000),
                        np.array([0.0, -0.0, 448.0, -448.0, 2 ** -9, 2 ** -10, 3 * 2 ** -10, 2 ** -6, 0.9375 * 2 ** -6, 17.0, 18.0, 19.0, 464.0 * 0 + 447.9])]).astype(np.float32)
    # exact ties between neighbouring codes, normal and subnormal ranges
    ties = np.array([(1 + (2 * k + 1) / 16) * 2.0 ** e for e in range(-6, 9) for k in range(8)] + [(2 * k + 1) * 2.0 ** -10 for k in range(8)], np.float32)
    ties = ties[np.abs(ties) <= 448]
    x = np.concatenate([x, ties, -ties])
    ref = torch.from_numpy(x).to(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(O.e4m3_round(x), ref)


def test_e8m0_scale_is_the_smallest_power_of_two_that_fits():
    rng = np.random.default_rng(1)
    amax = np.concatenate([np.exp(rng.uniform(-30, 30, 5000)), [448.0, 448.0001, 447.9999, 1.75, 1.7500001, 0.875, 3.5, 7.0, 1e-30, 1e30]]).astype(np.float32)
    E = O.e8m0_of_amax(amax).astype(np.int64)
    s = np.ldexp(1.0, E - 127)
    inside = (E > 1) & (E < 254)
    assert np.all(amax[inside].astype(np.float64) / s[inside] <= 448.0)
    assert np.all(amax[inside].astype(np.float64) / (s[inside] / 2) > 448.0)           # and no smaller power of two does
    assert O.e8m0_of_amax(np.zeros(3, np.float32)).tolist() == [127, 127, 127]


def test_quant_rows_invariants():
    rng = np.random.default_rng(2)
    x = (rng.normal(0, 1, (37, 200)) * np.exp(rng.uniform(-8, 8, (37, 1)))).astype(np.float32)
    x[5] = 0.0
    deq, E, q = O.quant_rows_e4m3(x)
    assert np.abs(q).max() <= 448 and np.array_equal(O.e4m3_round(q), q)              # codes are representable
    assert np.array_equal(deq[5], np.zeros(200, np.float32)) and E[5] == 127
    big = (np.abs(x) >= np.abs(x).max(axis=1, keepdims=True) * 2.0 ** -6) & (x != 0)    # elements within the normal range of the row
    rel = np.abs(deq - x)[big] / np.abs(x)[big]
    assert rel.max() <= 2.0 ** -4 + 1e-7                                                # half a unit of 3 mantissa bits
    # power-of-two scales commute with the quantiser: a row times 2^k gives the same codes, scale shifted by k
    deq2, E2, q2 = O.quant_rows_e4m3(x * np.float32(8.0))
    assert np.array_equal(q2, q) and np.array_equal(E2.astype(int) - E.astype(int), np.where(np.abs(x).max(1) > 0, 3, 0))


def test_gelu_bound_scale_bounds_every_row_and_mirrors_in_the_esm_layer():
    """The FFN-up fp8 output scale (oracle gelu_bound_scale): a valid scale for every row (no element above 448 * 2^e), and
    the fused / unfused oracle precisions differ only by that scale choice."""
    rng = np.random.default_rng(3)
    h = rng.standard_normal((50, 96)).astype(np.float32) * np.exp(rng.standard_normal((50, 1))).astype(np.float32)
    w = rng.standard_normal((200, 96)).astype(np.float32) * 0.3
    b = rng.standard_normal(200).astype(np.float32)
    hq, wq = O.quant_rows_e4m3(h)[0], O.quant_rows_e4m3(w)[0]
    E = O.gelu_bound_scale(h, w, b)
    f = O.gelu_erf(hq @ wq.T + b)
    cap = 448.0 * np.exp2(E.astype(np.float64) - 127)
    assert (np.abs(f).max(-1) <= cap).all()
    deq = O.quant_rows_e4m3_scaled(f, E)
    big = np.abs(f) > cap[:, None] * 2.0 ** -14          # e4m3 normals span 14 binades below the cap
    assert (np.abs(deq - f)[big] <= np.abs(f)[big] * 2.0 ** -4 + 1e-30).all()
    assert O.FP8.fused_gelu and not O.FP8_UNFUSED.fused_gelu and not O.BF16.fused_gelu
