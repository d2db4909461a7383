#!/usr/bin/env python3
"""Writes csrc/attn_fwd64_body.inc: the hand-placed instruction stream of csrc/attn_fwd64.hip (bf16 flash attention forward,
head_dim padded to 64, one wave per SIMD, two 32-query tiles per wave).

The whole K/V loop is ONE asm statement with literal registers; hipcc only sets up its pinned inputs and runs the epilogue.
Why literal registers: an inline-asm operand cannot name ONE register of a tuple, and the softmax works on single registers
of the 16-register MFMA results (tools/README.md, "attention forward").

Per wave: tile A = queries q0 .. q0+31, tile B = q0+32 .. q0+63.  Iteration j (key tile j, 64 keys) is two segments of
16 MFMAs each; the softmax of one tile runs on the vector pipe under the other tile's MFMAs:

    segment 1:  MFMA  QK_B(j)  [8]  PV_B(j-1) [8]      VALU  exp / sum / bf16-pack of S_A(j), row max of S_B(j)
                LDS   V(j) fragments (16 transposed reads)      DMA  V(j+3)
    segment 2:  MFMA  QK_A(j+1)[8]  PV_A(j)   [8]      VALU  exp / sum / bf16-pack of S_B(j), row max of S_A(j+1)
                LDS   K(j+2) fragments (8 reads)                DMA  K(j+5)

K and V fragments are double-buffered in AGPRs (set = tile & 1) so both query tiles use one LDS read of them.
Rare paths (not hand-placed): the rescale of the running maximum (a score more than 2^8 above it, the first tile, or a
mask that is not a prefix) and the masking of a tile that holds a hidden key.

    python tools/gen_attn_fwd64.py            # rewrite the .inc in place
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "prot2text-v2-esm3_amd", "csrc", "attn_fwd64_body.inc")

NS = 4                      # LDS ring slots per operand (K and V each); a slot = 64 keys x 128 B = 8 KiB
SLOT = 8192
SLACK = "0x41000000"        # 8.0: lazy-rescale threshold, log2 units
NINF = "0xff800000"

# ---- register map (must match attn_fwd64.hip) -------------------------------------------------
S = {"A": 0, "B": 32}                   # v: scores / probabilities, 32 per tile (t0: +0..15, t1: +16..31)
P = {"A": 64, "B": 80}                  # v: packed bf16 probabilities, 16 per tile
NEGM = {"A": 96, "B": 112}              # v: -m on 16 registers (accumulator input of QK^T)
LSUM = {"A": (128, 129), "B": (130, 131)}
MREF = {"A": 132, "B": 133}
SEEN = {"A": 134, "B": 135}
MLOC = {"A": 136, "B": 137}
KA = 138                                # v138..141: K fragment read address per k-step (slot 0)
VA = 142                                # v142..145: V transposed-read address per (dt, r)
CKA = 146                               # v146..149: the same for the slot being read this iteration
VOFF = 150                              # v150,151 DMA source offsets of this wave's two pieces; v152,153 the clamped ones of the tail tile
QOFF = {"A": 154, "B": 155}
CQ = {"A": 156, "B": 157}               # query + 1 - 8 hh (causal limit in the shifted key coordinates)
HH8 = 158
LANE = 159
CVA = 160                               # v160..163
DV = 164                                # DMA offset temporary
T = [166 + i for i in range(8)]         # temporaries of the rare paths
W = (174, 175)
LM = (176, 177)
NINFV = 178
MAXV = 179                              # highest VGPR the asm owns; the compiler keeps v[200:255]

O = {"A": (0, 16), "B": (32, 48)}       # a: O^T accumulators per d-tile
Q = {"A": 64, "B": 80}                  # a: + 4 kk


def KF(st, t, kk):
    return 96 + st * 32 + t * 16 + kk * 4


def VF(st, dt, ss):
    return 160 + st * 32 + (dt * 4 + ss) * 4


MAXA = 223

# SGPRs
KPTR, VPTR, QPTR, MPTR = 36, 38, 40, 42         # pairs
NIT, LDSK, SEQ, FLAGS = 44, 45, 46, 47          # FLAGS: bit0 = last tile partial, bit1 = mask is not a prefix
LDSV, Q0, TAILT, NITM1 = 48, 49, 50, 51
J, R, TK, TV, SL, SL1, SL2, SL3, DST, TMP, TMP2, KB, MF = 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64
VIS, CC = 66, 68                                # pairs
MAXS = 69

L = []
_lab = [0]


def e(s):
    L.append(s)


def lab(name):
    _lab[0] += 1
    return f".L{name}_{_lab[0]}_%="


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def ar(b, n=1):
    return f"a{b}" if n == 1 else f"a[{b}:{b + n - 1}]"


def sr(b, n=1):
    return f"s{b}" if n == 1 else f"s[{b}:{b + n - 1}]"


# ---- building blocks ---------------------------------------------------------------------------
def mfma_qk(Y, kset, i):
    t, kk = i >> 2, i & 3
    d = S[Y] + 16 * t
    c = vr(NEGM[Y], 16) if kk == 0 else vr(d, 16)
    return f"v_mfma_f32_32x32x16_bf16 {vr(d, 16)}, {ar(KF(kset, t, kk), 4)}, {ar(Q[Y] + 4 * kk, 4)}, {c}"


def mfma_pv(Y, vset, i):
    ss, dt = i >> 1, i & 1
    o = O[Y][dt]
    return f"v_mfma_f32_32x32x16_bf16 {ar(o, 16)}, {ar(VF(vset, dt, ss), 4)}, {vr(P[Y] + 4 * ss, 4)}, {ar(o, 16)}"


def max_chain(Y):
    """16 instructions: running maximum of the 32 scores of tile Y into MLOC[Y]."""
    s, m = S[Y], MLOC[Y]
    out = [f"v_max3_f32 {vr(m)}, {vr(s)}, {vr(s + 1)}, {vr(s + 2)}"]
    r = 3
    while r + 1 < 32:
        out.append(f"v_max3_f32 {vr(m)}, {vr(m)}, {vr(s + r)}, {vr(s + r + 1)}")
        r += 2
    out.append(f"v_max_f32 {vr(m)}, {vr(m)}, {vr(s + 31)}")
    assert len(out) == 16
    return out


def dma_piece(tile_s, dst_s, ptr_s, piece):
    """One 1-KiB LDS-DMA piece of tile `tile_s` (skipped past the last tile; the tail tile re-reads its last row)."""
    skip = lab("nodma")
    return [
        f"s_cmp_lt_u32 {sr(tile_s)}, {sr(NIT)}",
        f"s_cbranch_scc0 {skip}",
        f"s_cmp_eq_u32 {sr(tile_s)}, {sr(TAILT)}",
        f"s_cselect_b64 {sr(CC, 2)}, -1, 0",
        f"s_add_u32 {sr(TMP)}, {sr(dst_s)}, {piece * 4096}",
        f"s_mov_b32 m0, {sr(TMP)}",
        f"v_cndmask_b32 {vr(DV)}, {vr(VOFF + piece)}, {vr(VOFF + 2 + piece)}, {sr(CC, 2)}",
        f"global_load_lds_dwordx4 {vr(DV)}, {sr(ptr_s, 2)}",
        f"{skip}:",
    ]


def advance(ptr_s):
    return [f"s_add_u32 {sr(ptr_s)}, {sr(ptr_s)}, {SLOT}", f"s_addc_u32 {sr(ptr_s + 1)}, {sr(ptr_s + 1)}, 0"]


def k_reads(kset):
    out = []
    for t in range(2):
        for kk in range(4):
            out.append(f"ds_read_b128 {ar(KF(kset, t, kk), 4)}, {vr(CKA + kk)} offset:{t * 4096}")
    return out


def v_reads(vset):
    out = []
    for ss in range(4):
        for dt in range(2):
            for r in range(2):
                out.append(f"ds_read_b64_tr_b16 {ar(VF(vset, dt, ss) + 2 * r, 2)}, {vr(CVA + dt * 2 + r)} offset:{ss * 2048}")
    return out


def segment(X, Y, kset, vset, reads, dma):
    """16 MFMAs on tile Y beside the softmax of tile X.  reads: 16 or 8 LDS reads placed in the first 8 gaps;
    dma: [(gap, instruction list)]."""
    sx = S[X]
    la, lb = LSUM[X]
    mx = max_chain(Y)
    per = len(reads) // 8
    for g in range(16):
        e(mfma_qk(Y, kset, g) if g < 8 else mfma_pv(Y, vset, g - 8))
        if g < 8:
            for i in range(per):
                e(reads[g * per + i])
        e(f"v_exp_f32 {vr(sx + 2 * g)}, {vr(sx + 2 * g)}")
        if g >= 8:
            e(mx[2 * (g - 8)])
        if g > 0:
            e(f"v_add_f32 {vr(la)}, {vr(la)}, {vr(sx + 2 * g - 2)}")
        e(f"v_exp_f32 {vr(sx + 2 * g + 1)}, {vr(sx + 2 * g + 1)}")
        if g > 0:
            e(f"v_add_f32 {vr(lb)}, {vr(lb)}, {vr(sx + 2 * g - 1)}")
            e(f"v_cvt_pk_bf16_f32 {vr(P[X] + g - 1)}, {vr(sx + 2 * g - 2)}, {vr(sx + 2 * g - 1)}")
        if g >= 8:
            e(mx[2 * (g - 8) + 1])
        for gg, ins in dma:
            if gg == g:
                for i in ins:
                    e(i)
    e(f"v_add_f32 {vr(la)}, {vr(la)}, {vr(sx + 30)}")
    e("s_nop 0")
    e(f"v_add_f32 {vr(lb)}, {vr(lb)}, {vr(sx + 31)}")
    e(f"v_cvt_pk_bf16_f32 {vr(P[X] + 15)}, {vr(sx + 30)}, {vr(sx + 31)}")


def slow_path(X):
    """Exact online-softmax update of tile X's reference m (per query): scores, sums and O move to the new reference."""
    t0, t1, t2, t3, t4, t5 = T[:6]
    m, seen, mloc = MREF[X], SEEN[X], MLOC[X]
    e("s_nop 15")                                   # MFMA -> VALU distance for O
    e(f"v_mov_b32 {vr(t0)}, {vr(mloc)}")
    e(f"v_mov_b32 {vr(t1)}, {vr(mloc)}")
    e("s_nop 1")
    e(f"v_permlane32_swap_b32 {vr(t0)}, {vr(t1)}")
    e(f"v_max_f32 {vr(t0)}, {vr(t0)}, {vr(t1)}")                      # mx: the query's maximum, relative to m
    e(f"v_max_f32 {vr(t1)}, 0, {vr(t0)}")                             # seen: delta = max(mx, 0)
    e(f"v_cmp_neq_f32 vcc, {NINF}, {vr(t0)}")
    e(f"v_cndmask_b32 {vr(t2)}, 0, {vr(t0)}, vcc")                    # not seen: delta = mx (0 if no visible key yet)
    e(f"v_cndmask_b32_e64 {vr(t5)}, 0, 1, vcc")
    e(f"v_cmp_ne_u32 vcc, 0, {vr(seen)}")
    e(f"v_cndmask_b32 {vr(t3)}, {vr(t2)}, {vr(t1)}, vcc")             # delta
    e(f"v_exp_f32 {vr(t4)}, -{vr(t3)}")
    e(f"v_or_b32 {vr(seen)}, {vr(seen)}, {vr(t5)}")
    e(f"v_add_f32 {vr(m)}, {vr(m)}, {vr(t3)}")
    e(f"v_cndmask_b32 {vr(t4)}, 1.0, {vr(t4)}, vcc")                  # alpha (1 while nothing was accumulated)
    for i in range(16):
        e(f"v_sub_f32 {vr(NEGM[X] + i)}, 0, {vr(m)}")
    for r in range(32):
        e(f"v_sub_f32 {vr(S[X] + r)}, {vr(S[X] + r)}, {vr(t3)}")
    e(f"v_sub_f32 {vr(mloc)}, {vr(mloc)}, {vr(t3)}")
    for l in LSUM[X]:
        e(f"v_mul_f32 {vr(l)}, {vr(l)}, {vr(t4)}")
    for dt in range(2):
        for r in range(16):
            a = O[X][dt] + r
            tt = T[6 + (r & 1)]
            e(f"v_accvgpr_read_b32 {vr(tt)}, {ar(a)}")
            e("s_nop 0")
            e(f"v_mul_f32 {vr(tt)}, {vr(tt)}, {vr(t4)}")
            e("s_nop 0")
            e(f"v_accvgpr_write_b32 {ar(a)}, {vr(tt)}")
    e("s_nop 3")


def decide(X, causal):
    """Top of tile X's softmax segment: take the slow path on the first tile, for a non-prefix mask, or when a score is 2^8 above m."""
    slow, fast = lab("slow"), lab("fast")
    e(f"v_cmp_lt_f32 vcc, {SLACK}, {vr(MLOC[X])}")
    e(f"s_cmp_eq_u32 {sr(J)}, 0")
    e(f"s_cbranch_scc1 {slow}")
    e(f"s_bitcmp1_b32 {sr(FLAGS)}, 1")
    e(f"s_cbranch_scc1 {slow}")
    e(f"s_cbranch_vccz {fast}")
    e(f"{slow}:")
    slow_path(X)
    e(f"{fast}:")


def mask_tile(Y, tile_s, causal):
    """After the segment that produced S_Y(tile): if the tile holds a key hidden from a query of this wave, set those scores to
    -inf and redo the row maximum.  Visibility comes from the mask bytes (prefix or not) and, causal, the query index."""
    done = lab("nomask")
    t0, t1, t2, t3 = T[:4]
    # wave-uniform test
    e(f"s_lshr_b32 {sr(MF)}, {sr(FLAGS)}, 1")                                    # not a prefix: every tile
    e(f"s_cmp_eq_u32 {sr(tile_s)}, {sr(NITM1)}")
    e(f"s_cselect_b32 {sr(TMP)}, {sr(FLAGS)}, 0")
    e(f"s_or_b32 {sr(MF)}, {sr(MF)}, {sr(TMP)}")                                 # the last tile when it is partial (bit 0) / not a prefix
    e(f"s_lshl_b32 {sr(KB)}, {sr(tile_s)}, 6")
    if causal:
        e(f"s_add_u32 {sr(TMP)}, {sr(KB)}, 63")
        e(f"s_cmp_gt_u32 {sr(TMP)}, {sr(Q0)}")                                   # the tile reaches this wave's diagonal
        e(f"s_cselect_b32 {sr(TMP)}, 1, 0")
        e(f"s_or_b32 {sr(MF)}, {sr(MF)}, {sr(TMP)}")
    e(f"s_cmp_lt_u32 {sr(tile_s)}, {sr(NIT)}")
    e(f"s_cselect_b32 {sr(MF)}, {sr(MF)}, 0")
    e(f"s_cmp_eq_u32 {sr(MF)}, 0")
    e(f"s_cbranch_scc1 {done}")
    # 64-bit visibility of the tile's keys
    e(f"v_add_u32 {vr(t0)}, {sr(KB)}, {vr(LANE)}")
    e(f"v_cmp_gt_u32 {sr(CC, 2)}, {sr(SEQ)}, {vr(t0)}")
    e(f"s_sub_u32 {sr(TMP)}, {sr(SEQ)}, 1")
    e(f"v_min_u32 {vr(t0)}, {sr(TMP)}, {vr(t0)}")
    e(f"global_load_ubyte {vr(t1)}, {vr(t0)}, {sr(MPTR, 2)}")
    e("s_waitcnt vmcnt(0)")
    e(f"v_cmp_ne_u32 vcc, 0, {vr(t1)}")
    e(f"s_and_b64 {sr(VIS, 2)}, vcc, {sr(CC, 2)}")
    e(f"v_lshrrev_b64 {vr(W[0], 2)}, {vr(HH8)}, {sr(VIS, 2)}")
    if causal:
        e(f"v_subrev_u32 {vr(t2)}, {sr(KB)}, {vr(CQ[Y])}")                       # keys below this bit index are visible
        e(f"v_med3_i32 {vr(t2)}, {vr(t2)}, 0, 56")
        e(f"v_lshlrev_b64 {vr(LM[0], 2)}, {vr(t2)}, 1")
        e(f"v_add_co_u32 {vr(LM[0])}, vcc, -1, {vr(LM[0])}")
        e(f"v_addc_co_u32 {vr(LM[1])}, vcc, -1, {vr(LM[1])}, vcc")
        e(f"v_and_b32 {vr(W[0])}, {vr(W[0])}, {vr(LM[0])}")
        e(f"v_and_b32 {vr(W[1])}, {vr(W[1])}, {vr(LM[1])}")
    e(f"v_mov_b32 {vr(NINFV)}, {NINF}")
    for r in range(32):
        t, rr = r >> 4, r & 15
        c = 16 * (rr >> 3) + (rr & 7)
        tt = T[2 + (r & 1)]
        e(f"v_bfe_i32 {vr(tt)}, {vr(W[t])}, {c}, 1")
        e(f"v_bfi_b32 {vr(S[Y] + r)}, {vr(tt)}, {vr(S[Y] + r)}, {vr(NINFV)}")
    for i in max_chain(Y):
        e(i)
    e(f"{done}:")


def barrier_wait():
    """Pieces still allowed in flight when tile pair j is needed: 2 * min(n_it - 1 - j, 4)."""
    labs = {n: lab(f"w{n}") for n in (8, 6, 4, 2)}
    bar = lab("bar")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_sub_u32 {sr(R)}, {sr(NITM1)}, {sr(J)}")
    e(f"s_cmp_ge_u32 {sr(R)}, 4")
    e(f"s_cbranch_scc1 {labs[8]}")
    e(f"s_cmp_eq_u32 {sr(R)}, 3")
    e(f"s_cbranch_scc1 {labs[6]}")
    e(f"s_cmp_eq_u32 {sr(R)}, 2")
    e(f"s_cbranch_scc1 {labs[4]}")
    e(f"s_cmp_eq_u32 {sr(R)}, 1")
    e(f"s_cbranch_scc1 {labs[2]}")
    e("s_waitcnt vmcnt(0)")
    e(f"s_branch {bar}")
    for n in (2, 4, 6):
        e(f"{labs[n]}:")
        e(f"s_waitcnt vmcnt({n})")
        e(f"s_branch {bar}")
    e(f"{labs[8]}:")
    e("s_waitcnt vmcnt(8)")
    e(f"{bar}:")
    e("s_barrier")


def iteration(par, causal):
    barrier_wait()
    # ring slots of this iteration: read V(j) from slot j & 3, K(j+2) from (j+2) & 3; write V(j+3), K(j+5)
    e(f"s_and_b32 {sr(TMP)}, {sr(J)}, 3")
    e(f"s_lshl_b32 {sr(SL)}, {sr(TMP)}, 13")
    e(f"s_xor_b32 {sr(SL2)}, {sr(SL)}, 0x4000")
    e(f"s_add_u32 {sr(TV)}, {sr(J)}, 3")
    e(f"s_add_u32 {sr(TK)}, {sr(J)}, 5")
    e(f"s_and_b32 {sr(TMP)}, {sr(TV)}, 3")
    e(f"s_lshl_b32 {sr(SL3)}, {sr(TMP)}, 13")
    e(f"s_and_b32 {sr(TMP)}, {sr(TK)}, 3")
    e(f"s_lshl_b32 {sr(SL1)}, {sr(TMP)}, 13")
    for i in range(4):
        e(f"v_add_u32 {vr(CVA + i)}, {sr(SL)}, {vr(VA + i)}")
    for i in range(4):
        e(f"v_add_u32 {vr(CKA + i)}, {sr(SL2)}, {vr(KA + i)}")
    # segment 1: softmax A(j) beside QK_B(j), PV_B(j-1)
    decide("A", causal)
    e(f"s_add_u32 {sr(DST)}, {sr(LDSV)}, {sr(SL3)}")
    segment("A", "B", par, 1 - par, v_reads(par),
            [(3, dma_piece(TV, DST, VPTR, 0)), (11, dma_piece(TV, DST, VPTR, 1) + advance(VPTR))])
    mask_tile("B", J, causal)
    e("s_waitcnt lgkmcnt(0)")
    # segment 2: softmax B(j) beside QK_A(j+1), PV_A(j)
    decide("B", causal)
    e(f"s_add_u32 {sr(DST)}, {sr(LDSK)}, {sr(SL1)}")
    segment("B", "A", 1 - par, par, k_reads(par),
            [(3, dma_piece(TK, DST, KPTR, 0)), (11, dma_piece(TK, DST, KPTR, 1) + advance(KPTR))])
    e(f"s_add_u32 {sr(TMP2)}, {sr(J)}, 1")
    mask_tile("A", TMP2, causal)


def prologue(causal):
    fin0 = lab("fin0")
    # lane, zeroed state
    for X in "AB":
        for dt in range(2):
            for r in range(16):
                e(f"v_accvgpr_write_b32 {ar(O[X][dt] + r)}, 0")
        for i in range(16):
            e(f"v_mov_b32 {vr(NEGM[X] + i)}, 0")
        for l in LSUM[X]:
            e(f"v_mov_b32 {vr(l)}, 0")
        e(f"v_mov_b32 {vr(MREF[X])}, 0")
        e(f"v_mov_b32 {vr(SEEN[X])}, 0")
    for i in range(16):
        e(f"v_mov_b32 {vr(P['B'] + i)}, 0")
    for dt in range(2):
        for ss in range(4):
            for r in range(4):
                e(f"v_accvgpr_write_b32 {ar(VF(1, dt, ss) + r)}, 0")
    e(f"s_cmp_eq_u32 {sr(NIT)}, 0")
    e(f"s_cbranch_scc1 {fin0}")
    e(f"s_sub_u32 {sr(NITM1)}, {sr(NIT)}, 1")
    e(f"s_lshr_b32 {sr(TAILT)}, {sr(SEQ)}, 6")
    e(f"s_add_u32 {sr(LDSV)}, {sr(LDSK)}, {NS * SLOT}")
    # Q fragments first (vmcnt completes in order: any wait that covers a K/V piece covers them)
    for X in "AB":
        for kk in range(4):
            e(f"global_load_dwordx4 {ar(Q[X] + 4 * kk, 4)}, {vr(QOFF[X])}, {sr(QPTR, 2)} offset:{32 * kk}")
    # K0 K1 | V0 K2 | V1 K3   (slot = tile & 3)
    def tile_dma(kind, tile):
        ptr, base = (KPTR, LDSK) if kind == "K" else (VPTR, LDSV)
        e(f"s_mov_b32 {sr(TK)}, {tile}")
        e(f"s_add_u32 {sr(DST)}, {sr(base)}, {(tile & 3) * SLOT}")
        for p in range(2):
            for i in dma_piece(TK, DST, ptr, p):
                e(i)
        for i in advance(ptr):
            e(i)
    for kind, tile in (("K", 0), ("K", 1), ("V", 0), ("K", 2), ("V", 1), ("K", 3)):
        tile_dma(kind, tile)
    w8, wb = lab("pw8"), lab("pwb")
    e(f"s_cmp_ge_u32 {sr(NIT)}, 4")
    e(f"s_cbranch_scc1 {w8}")
    e("s_waitcnt vmcnt(0)")
    e(f"s_branch {wb}")
    e(f"{w8}:")
    e("s_waitcnt vmcnt(8)")
    e(f"{wb}:")
    e("s_barrier")
    # K(0) -> set 0, K(1) -> set 1
    for st in range(2):
        for i in range(4):
            e(f"v_add_u32 {vr(CKA + i)}, {st * SLOT}, {vr(KA + i)}")
        for i in k_reads(st):
            e(i)
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                                   # every wave has K(0) in registers: slot 0 may take K(4)
    tile_dma("V", 2)
    tile_dma("K", 4)
    for i in range(8):
        e(mfma_qk("A", 0, i))
    e("s_nop 15")
    for i in max_chain("A"):
        e(i)
    e(f"s_mov_b32 {sr(J)}, 0")
    e(f"s_mov_b32 {sr(TMP2)}, 0")
    mask_tile("A", TMP2, causal)
    return fin0


def body(causal):
    del L[:]
    fin0 = prologue(causal)
    loop, end0, end1, fin = lab("loop"), lab("end0"), lab("end1"), lab("fin")
    e(f"{loop}:")
    iteration(0, causal)
    e(f"s_add_u32 {sr(J)}, {sr(J)}, 1")
    e(f"s_cmp_ge_u32 {sr(J)}, {sr(NIT)}")
    e(f"s_cbranch_scc1 {end0}")
    iteration(1, causal)
    e(f"s_add_u32 {sr(J)}, {sr(J)}, 1")
    e(f"s_cmp_lt_u32 {sr(J)}, {sr(NIT)}")
    e(f"s_cbranch_scc1 {loop}")
    e(f"{end1}:")
    for i in range(8):
        e(mfma_pv("B", 1, i))
    e(f"s_branch {fin}")
    e(f"{end0}:")
    for i in range(8):
        e(mfma_pv("B", 0, i))
    e(f"{fin}:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e(f"{fin0}:")
    e("s_nop 15")
    e("s_nop 3")
    return list(L)


def clobbers():
    pinned_v = set(range(KA, KA + 4)) | set(range(VA, VA + 4)) | set(range(VOFF, VOFF + 4)) | set(range(154, 160)) \
        | set(range(128, 134))
    pinned_s = set(range(36, 52))
    out = [f"v{i}" for i in range(MAXV + 1) if i not in pinned_v]
    out += [f"a{i}" for i in range(64, MAXA + 1)]
    out += [f"s{i}" for i in range(MAXS + 1) if i >= 52 or i in (TAILT, NITM1, LDSV)]
    out += ["vcc", "scc", "memory"]
    return out, pinned_s


def main():
    parts = ["// GENERATED by tools/gen_attn_fwd64.py -- do not edit.\n"]
    for causal in (0, 1):
        lines = body(causal)
        parts.append(f"#define P2T_ATTN64_BODY_{causal} \\\n" + " \\\n".join('    "' + s + '\\n\\t"' for s in lines) + "\n")
        n_mfma = sum(1 for s in lines if s.startswith("v_mfma"))
        print(f"causal={causal}: {len(lines)} lines, {n_mfma} MFMAs")
    cl, _ = clobbers()
    parts.append("#define P2T_ATTN64_CLOBBERS " + ", ".join(f'"{c}"' for c in cl) + "\n")
    with open(OUT, "w") as f:
        f.write("\n".join(parts))


if __name__ == "__main__":
    main()
