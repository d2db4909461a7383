#!/usr/bin/env python3
"""Writes csrc/attn_fwd64_body.inc: the hand-placed instruction stream of csrc/attn_fwd64.hip (bf16 flash attention forward,
head_dim padded to 64, one wave per SIMD, two 32-query tiles per wave).

The whole K/V loop is ONE asm statement with literal registers; hipcc only sets up its pinned inputs and runs the epilogue.
Why literal registers: an inline-asm operand cannot name ONE register of a tuple, and the softmax works on single registers
of the 16-register MFMA results.

Per wave: tile A = queries q0 .. q0+31, tile B = q0+32 .. q0+63.  Iteration j (key tile j, 64 keys) is two segments of
20 MFMAs each; the softmax of one tile runs on the vector pipe under the other tile's MFMAs:

    segment 1:  MFMA  QK_B(j)  [8]  PV_B(j-1) + row sums [8 + 4]   VALU  exp / bf16-pack of S_A(j), row max of S_B(j)
                LDS   V(j) fragments (16 transposed reads)          DMA  V(j+3)
    segment 2:  MFMA  QK_A(j+1)[8]  PV_A(j)   + row sums [8 + 4]   VALU  exp / bf16-pack of S_B(j), row max of S_A(j+1)
                LDS   K(j+2) fragments (8 reads)                    DMA  K(j+5)

The row sums of P are a fifth "d-tile" of the PV product: an all-ones A operand against the same packed P (4 MFMAs per
segment on the matrix pipe, which has the slack, instead of 32 v_add_f32 on the vector pipe, which is the longer one:
profiles/r04_attn64_v1_diag.log).  K and V fragments are double-buffered in AGPRs (set = tile & 1) so both query tiles use
one LDS read of them; the loop is unrolled over the 4 ring slots so every LDS address is a register + immediate.
Rare paths (out of line, entered through s_setpc stubs): the rescale of the running maximum (a score more than 2^8 above
it, the first tile, or a mask that is not a prefix), the masking of a tile that holds a hidden key, and the switch of the
DMA source offsets to the clamped ones for the tile that crosses the end of the sequence.

    python tools/gen_attn_fwd64.py            # rewrite the .inc in place
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "prot2text-v2-esm3_amd", "csrc", "attn_fwd64_body.inc")

NS = 4                      # LDS ring slots per operand (K and V each); a slot = 64 keys x 128 B = 8 KiB
SLOT = 8192
NINF = "0xff800000"
ONES = "0x3f803f80"         # two bf16 1.0

# ---- register map (must match attn_fwd64.hip) -------------------------------------------------
S = {"A": 0, "B": 32}                   # v: scores / probabilities, 32 per tile (t0: +0..15, t1: +16..31)
P = {"A": 64, "B": 80}                  # v: packed bf16 probabilities, 16 per tile
NEGM = {"A": 96, "B": 112}              # v: -m on 16 registers (accumulator input of QK^T)
MREF = {"A": 132, "B": 133}
SEEN = {"A": 134, "B": 135}
MLOC = {"A": 136, "B": 137}
KA = 138                                # v138..141: K fragment read address per k-step (slot 0)
VA = 142                                # v142..145: V transposed-read address per (dt, r)
DVK, DVV = 146, 148                     # v146,147 / v148,149: DMA source offsets in use for K / V (switched to the clamped ones at the tail tile)
VOFF = 150                              # v150,151 offsets of this wave's two pieces; v152,153 the clamped ones of the tail tile
QOFF = {"A": 154, "B": 155}
CQ = {"A": 156, "B": 157}               # query + 1 - 8 hh (causal limit in the shifted key coordinates)
HH8 = 158
LANE = 159
T = [166 + i for i in range(8)]         # temporaries of the rare paths
W = (174, 175)
LM = (176, 177)
NINFV = 178
ONEV = 180                              # v180..183: all-ones bf16 fragment (A operand of the row-sum MFMAs)
MAXV = 183                              # highest VGPR the asm owns

O = {"A": (0, 16), "B": (32, 48)}       # a: O^T accumulators per d-tile
Q = {"A": 64, "B": 80}                  # a: + 4 kk
LACC = {"A": 224, "B": 240}             # a: row-sum accumulators (every register of a lane holds its query's sum)


def KF(st, t, kk):
    return 96 + st * 32 + t * 16 + kk * 4


def VF(st, dt, ss):
    return 160 + st * 32 + (dt * 4 + ss) * 4


# SGPRs
KPTR, VPTR, QPTR, MPTR = 36, 38, 40, 42         # pairs
NIT, LDSK, SEQ, FLAGS = 44, 45, 46, 47          # FLAGS: bit0 = last tile partial, bit1 = mask is not a prefix
LDSV, Q0, TAILT, NITM1 = 48, 49, 50, 51
J, R, TK, TV, J1, SLK, SLKS, MFROM, TMP, TMP2, KB = 52, 53, 54, 55, 56, 57, 58, 59, 61, 62, 63
VIS, CC, RET = 66, 68, 84                       # pairs
ACC0, PREV, CUR = 72, 80, 82                    # diagnostic build: s72..79 cycle sums per phase, s80 previous stamp, s[82:83] s_memtime
# the continuous K / V stream: the NEXT block's first tiles ride in the ring slots the last iterations of this block leave idle
KNEXT, VNEXT = 86, 88                           # pairs (inputs): the next block's K / V rows
NITN, C0, PREF = 90, 91, 92                     # inputs: tiles of the next block to request here (0: none), ring phase of this block's
                                                # tile 0 (cumulative tile count & 3), 1 = this block's first tiles were requested by its predecessor
LIMV, LIMK, TAILV, TAILK, LVT, LKT = 60, 64, 65, 70, 71, 93
MAXS = 93

# knobs of the diagnostic variants (lab build): the product uses the defaults
VAR = {"dma_gaps": (5, 15), "no_dma": False, "no_softmax": False, "no_lds": False, "no_exp": False, "no_cvtmax": False, "qk_agpr": False, "dma_form": "lds"}

L = []
TAIL = []                   # out-of-line code, emitted after the main stream
_lab = [0]
DIAG = [False]


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


def stamp(k):
    """Diagnostic build only: add the cycles since the previous stamp to phase k (s_memtime returns through lgkmcnt)."""
    if not DIAG[0]:
        return
    e(f"s_memtime {sr(CUR, 2)}")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_sub_u32 {sr(TMP)}, {sr(CUR)}, {sr(PREV)}")
    e(f"s_add_u32 {sr(ACC0 + k)}, {sr(ACC0 + k)}, {sr(TMP)}")
    e(f"s_mov_b32 {sr(PREV)}, {sr(CUR)}")


def call(cond_branch, sub, pre=()):
    """Hot path: `cond_branch stub`; the stub (out of line) loads the return address and jumps to the subroutine `sub`, which
    ends with s_setpc_b64 RET."""
    stub, back, pc = lab("stub"), lab("back"), lab("pc")
    e(f"{cond_branch} {stub}")
    e(f"{back}:")
    TAIL.append(f"{stub}:")
    TAIL.extend(pre)
    TAIL.append(f"s_getpc_b64 {sr(RET, 2)}")
    TAIL.append(f"{pc}:")
    TAIL.append(f"s_sub_u32 {sr(RET)}, {sr(RET)}, {pc} - {back}")
    TAIL.append(f"s_subb_u32 {sr(RET + 1)}, {sr(RET + 1)}, 0")
    TAIL.append(f"s_branch {sub}")


# ---- building blocks ---------------------------------------------------------------------------
def mfma_qk(Y, kset, i):
    t, kk = i >> 2, i & 3
    d = S[Y] + 16 * t
    c = vr(NEGM[Y], 16) if kk == 0 else vr(d, 16)
    if VAR["qk_agpr"]:          # timing experiment only: the scores land in the O accumulators
        return f"v_mfma_f32_32x32x16_bf16 {ar(d, 16)}, {ar(KF(kset, t, kk), 4)}, {ar(Q[Y] + 4 * kk, 4)}, {ar(d, 16)}"
    return f"v_mfma_f32_32x32x16_bf16 {vr(d, 16)}, {ar(KF(kset, t, kk), 4)}, {ar(Q[Y] + 4 * kk, 4)}, {c}"


def mfma_pv(Y, vset, i):
    """12 MFMAs: per 16-key step ss the two d-tiles of O^T and the row sums."""
    ss, k = divmod(i, 3)
    if k == 2:
        o = LACC[Y]
        return f"v_mfma_f32_32x32x16_bf16 {ar(o, 16)}, {vr(ONEV, 4)}, {vr(P[Y] + 4 * ss, 4)}, {ar(o, 16)}"
    o = O[Y][k]
    return f"v_mfma_f32_32x32x16_bf16 {ar(o, 16)}, {ar(VF(vset, k, ss), 4)}, {vr(P[Y] + 4 * ss, 4)}, {ar(o, 16)}"


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


def k_reads(kset, slot):
    out = []
    for t in range(2):
        for kk in range(4):
            out.append(f"ds_read_b128 {ar(KF(kset, t, kk), 4)}, {vr(KA + kk)} offset:{slot * SLOT + t * 4096}")
    return out


def v_reads(vset, slot):
    out = []
    for ss in range(4):
        for dt in range(2):
            for r in range(2):
                out.append(f"ds_read_b64_tr_b16 {ar(VF(vset, dt, ss) + 2 * r, 2)}, {vr(VA + dt * 2 + r)} offset:{slot * SLOT + ss * 2048}")
    return out


def spread(n, gaps):
    """n items over the listed gaps, as evenly as integers allow: {gap: count}."""
    out = {}
    for i, g in enumerate(gaps):
        out[g] = (n * (i + 1)) // len(gaps) - (n * i) // len(gaps)
    return out


def segment(X, Y, kset, vset, reads, read_gaps, dma, hooks):
    """20 MFMAs on tile Y beside the softmax of tile X.  reads: 16 or 8 LDS reads spread over the gaps read_gaps = (first, end);
    dma: {gap: (ring base sgpr, LDS byte offset, source offset vgpr, source pointer sgpr pair, tile index sgpr, limit sgpr)};
    hooks: {gap: function emitting that gap's scalar bookkeeping} -- everything an iteration needs besides the softmax rides in
    MFMA gaps, the segment boundaries hold only the rescale test."""
    sx = S[X]
    mx = max_chain(Y)
    if VAR["no_lds"]:
        reads = []
    reads = list(reads)
    n_rd = spread(len(reads), list(range(*read_gaps)))
    n_exp = spread(32, list(range(20)))
    # row maximum of S_Y: its first half is complete behind MFMA 3, the second behind MFMA 7 (+ the MFMA -> VALU distance);
    # it starts behind the hidden-key test of gap 8, so a masked tile needs no second pass
    n_max = spread(16, list(range(9, 20)))
    done_exp, done_cvt, done_max = 0, 0, 0
    for g in range(20):
        if g in dma and not VAR["no_dma"]:
            base, off = dma[g][0], dma[g][1]
            e(f"s_add_u32 m0, {sr(base)}, {off}")
        e(mfma_qk(Y, kset, g) if g < 8 else mfma_pv(Y, vset, g - 8))
        for _ in range(n_rd.get(g, 0)):
            e(reads.pop(0))
        # a pair is packed one gap after its second exponential (the transcendental result needs a wait state before its use)
        ready = done_exp // 2
        for _ in range(n_exp[g]):
            if not VAR["no_softmax"] and not VAR["no_exp"]:
                e(f"v_exp_f32 {vr(sx + done_exp)}, {vr(sx + done_exp)}")
            done_exp += 1
            if done_cvt < ready:
                if not VAR["no_softmax"] and not VAR["no_cvtmax"]:
                    e(f"v_cvt_pk_bf16_f32 {vr(P[X] + done_cvt)}, {vr(sx + 2 * done_cvt)}, {vr(sx + 2 * done_cvt + 1)}")
                done_cvt += 1
            if done_max < 16 and n_max.get(g, 0) > 0:
                if not VAR["no_softmax"] and not VAR["no_cvtmax"]:
                    e(mx[done_max])
                done_max += 1
                n_max[g] -= 1
        while n_max.get(g, 0) > 0:
            if not VAR["no_softmax"] and not VAR["no_cvtmax"]:
                e(mx[done_max])
            done_max += 1
            n_max[g] -= 1
        if g in hooks:
            hooks[g]()
        if g in dma and not VAR["no_dma"]:
            _, _, voff, ptr, tile, lim = dma[g]
            skip = lab("nodma")
            e(f"s_cmp_lt_u32 {sr(tile)}, {sr(lim)}")
            e(f"s_cbranch_scc0 {skip}")
            if VAR["dma_form"] == "lds":
                e(f"global_load_lds_dwordx4 {vr(voff)}, {sr(ptr, 2)}")
            else:
                # timing experiments (wrong results): what a register-staged piece would cost the issuing wave --
                # "reg": the plain load alone; "reg+write": plus the 16-byte LDS store of a piece loaded earlier
                e(f"global_load_dwordx4 {vr(T[4], 4)}, {vr(voff)}, {sr(ptr, 2)}")
                if VAR["dma_form"] == "reg+write":
                    e(f"v_lshlrev_b32 {vr(T[3])}, 4, {vr(LANE)}")
                    e(f"ds_write_b128 {vr(T[3])}, {vr(T[4], 4)} offset:64512")
            e(f"{skip}:")
    assert done_exp == 32 and done_max == 16 and not reads
    while done_cvt < 16:
        if done_cvt == 15:
            e("s_nop 0")
        if not VAR["no_softmax"] and not VAR["no_cvtmax"]:
            e(f"v_cvt_pk_bf16_f32 {vr(P[X] + done_cvt)}, {vr(sx + 2 * done_cvt)}, {vr(sx + 2 * done_cvt + 1)}")
        done_cvt += 1


def sub_slow(X, name):
    """Subroutine: exact online-softmax update of tile X's reference m (per query); scores, row sums and O move to the new reference."""
    t0, t1, t2, t3, t4, t5 = T[:6]
    m, seen, mloc = MREF[X], SEEN[X], MLOC[X]
    o = TAIL.append
    o(f"{name}:")
    o("s_nop 15")                                   # MFMA -> VALU distance for O and the row sums
    o(f"v_mov_b32 {vr(t0)}, {vr(mloc)}")
    o(f"v_mov_b32 {vr(t1)}, {vr(mloc)}")
    o("s_nop 1")
    o(f"v_permlane32_swap_b32 {vr(t0)}, {vr(t1)}")
    o(f"v_max_f32 {vr(t0)}, {vr(t0)}, {vr(t1)}")                      # mx: the query's maximum, relative to m
    o(f"v_max_f32 {vr(t1)}, 0, {vr(t0)}")                             # seen: delta = max(mx, 0)
    o(f"v_cmp_neq_f32 vcc, {NINF}, {vr(t0)}")
    o(f"v_cndmask_b32 {vr(t2)}, 0, {vr(t0)}, vcc")                    # not seen: delta = mx (0 if no visible key yet)
    o(f"v_cndmask_b32_e64 {vr(t5)}, 0, 1, vcc")
    o(f"v_cmp_ne_u32 vcc, 0, {vr(seen)}")
    o(f"v_cndmask_b32 {vr(t3)}, {vr(t2)}, {vr(t1)}, vcc")             # delta
    o(f"v_exp_f32 {vr(t4)}, -{vr(t3)}")
    o(f"v_or_b32 {vr(seen)}, {vr(seen)}, {vr(t5)}")
    o(f"v_add_f32 {vr(m)}, {vr(m)}, {vr(t3)}")
    o(f"v_cndmask_b32 {vr(t4)}, 1.0, {vr(t4)}, vcc")                  # alpha (1 while nothing was accumulated)
    for i in range(16):
        o(f"v_sub_f32 {vr(NEGM[X] + i)}, 0, {vr(m)}")
    for r in range(32):
        o(f"v_sub_f32 {vr(S[X] + r)}, {vr(S[X] + r)}, {vr(t3)}")
    o(f"v_sub_f32 {vr(mloc)}, {vr(mloc)}, {vr(t3)}")
    regs = [O[X][dt] + r for dt in range(2) for r in range(16)] + [LACC[X]]      # (only register 0 of the row sums is read at the end)
    for i, a in enumerate(regs):
        tt = T[6 + (i & 1)]
        o(f"v_accvgpr_read_b32 {vr(tt)}, {ar(a)}")
        o(f"v_mul_f32 {vr(tt)}, {vr(tt)}, {vr(t4)}")
        o(f"v_accvgpr_write_b32 {ar(a)}, {vr(tt)}")
    o("s_nop 3")
    o(f"s_setpc_b64 {sr(RET, 2)}")


def sub_mask(Y, name, causal):
    """Subroutine (KB = 64 * tile): set the scores of tile Y's hidden keys to -inf (called before the row maximum is taken).
    Visibility comes from the mask bytes (prefix or not) and, causal, the query index."""
    t0, t1, t2, t3 = T[:4]
    o = TAIL.append
    ret = lab("mret")
    o(f"{name}:")
    o(f"s_lshr_b32 {sr(TMP)}, {sr(KB)}, 6")
    o(f"s_cmp_ge_u32 {sr(TMP)}, {sr(NIT)}")                                      # scores of a tile past the end are never used
    o(f"s_cbranch_scc1 {ret}")
    o(f"v_add_u32 {vr(t0)}, {sr(KB)}, {vr(LANE)}")
    o(f"v_cmp_gt_u32 {sr(CC, 2)}, {sr(SEQ)}, {vr(t0)}")
    o(f"s_sub_u32 {sr(TMP)}, {sr(SEQ)}, 1")
    o(f"v_min_u32 {vr(t0)}, {sr(TMP)}, {vr(t0)}")
    o(f"global_load_ubyte {vr(t1)}, {vr(t0)}, {sr(MPTR, 2)}")
    o("s_waitcnt vmcnt(0)")
    o(f"v_cmp_ne_u32 vcc, 0, {vr(t1)}")
    o(f"s_and_b64 {sr(VIS, 2)}, vcc, {sr(CC, 2)}")
    o(f"v_lshrrev_b64 {vr(W[0], 2)}, {vr(HH8)}, {sr(VIS, 2)}")
    if causal:
        o(f"v_subrev_u32 {vr(t2)}, {sr(KB)}, {vr(CQ[Y])}")                       # keys below this bit index are visible
        o(f"v_med3_i32 {vr(t2)}, {vr(t2)}, 0, 56")
        o(f"v_lshlrev_b64 {vr(LM[0], 2)}, {vr(t2)}, 1")
        o(f"v_add_co_u32 {vr(LM[0])}, vcc, -1, {vr(LM[0])}")
        o(f"v_addc_co_u32 {vr(LM[1])}, vcc, -1, {vr(LM[1])}, vcc")
        o(f"v_and_b32 {vr(W[0])}, {vr(W[0])}, {vr(LM[0])}")
        o(f"v_and_b32 {vr(W[1])}, {vr(W[1])}, {vr(LM[1])}")
    o(f"v_mov_b32 {vr(NINFV)}, {NINF}")
    for r in range(32):
        t, rr = r >> 4, r & 15
        c = 16 * (rr >> 3) + (rr & 7)
        tt = T[2 + (r & 1)]
        o(f"v_bfe_i32 {vr(tt)}, {vr(W[t])}, {c}, 1")
        o(f"v_bfi_b32 {vr(S[Y] + r)}, {vr(tt)}, {vr(S[Y] + r)}, {vr(NINFV)}")
    o(f"{ret}:")
    o(f"s_setpc_b64 {sr(RET, 2)}")


def sub_tail(name, dv):
    o = TAIL.append
    o(f"{name}:")
    o(f"v_mov_b32 {vr(dv)}, {vr(VOFF + 2)}")
    o(f"v_mov_b32 {vr(dv + 1)}, {vr(VOFF + 3)}")
    o(f"s_setpc_b64 {sr(RET, 2)}")


SUBS = {}


def decide(X):
    """Top of tile X's softmax segment: rescale when a score is more than SLK above the reference (SLK = -inf on the first
    tile and for a mask that is not a prefix: every lane that saw a key takes the exact path)."""
    e(f"v_cmp_lt_f32 vcc, {sr(SLK)}, {vr(MLOC[X])}")
    call("s_cbranch_vccnz", SUBS["slow" + X])


def mask_check(Y, tile_s):
    e(f"s_cmp_ge_u32 {sr(tile_s)}, {sr(MFROM)}")
    call("s_cbranch_scc1", SUBS["mask" + Y], pre=[f"s_lshl_b32 {sr(KB)}, {sr(tile_s)}, 6"])


def barrier_wait():
    """Pieces still allowed in flight when tile pair j is needed = those issued behind it: V(j+1), K(j+3), V(j+2), K(j+4), two pieces
    each, as far as they exist (this block's tiles, then the next block's first ones: LVT / LKT = total tiles of the V / K stream)."""
    slow, bar = lab("wslow"), lab("bar")
    labs = {n: lab(f"w{n}") for n in (6, 4, 2, 0)}
    e(f"s_add_u32 {sr(R)}, {sr(J)}, 2")
    e(f"s_cmp_lt_u32 {sr(R)}, {sr(LVT)}")
    e(f"s_cbranch_scc0 {slow}")
    e(f"s_add_u32 {sr(R)}, {sr(J)}, 4")
    e(f"s_cmp_lt_u32 {sr(R)}, {sr(LKT)}")
    e(f"s_cbranch_scc0 {slow}")
    e("s_waitcnt vmcnt(8)")
    e(f"{bar}:")
    e("s_barrier")
    o = TAIL.append
    o(f"{slow}:")
    o(f"s_mov_b32 {sr(R)}, 0")
    for d, lim in ((1, LVT), (3, LKT), (2, LVT), (4, LKT)):
        o(f"s_add_u32 {sr(TMP)}, {sr(J)}, {d}")
        o(f"s_cmp_lt_u32 {sr(TMP)}, {sr(lim)}")
        o(f"s_cselect_b32 {sr(TMP)}, 2, 0")
        o(f"s_add_u32 {sr(R)}, {sr(R)}, {sr(TMP)}")
    o(f"s_cmp_ge_u32 {sr(R)}, 8")
    o(f"s_cbranch_scc0 {labs[6]}")
    o("s_waitcnt vmcnt(8)")
    o(f"s_branch {bar}")
    for n, nxt in ((6, 4), (4, 2), (2, 0)):
        o(f"{labs[n]}:")
        o(f"s_cmp_eq_u32 {sr(R)}, {n}")
        o(f"s_cbranch_scc0 {labs[nxt]}")
        o(f"s_waitcnt vmcnt({n})")
        o(f"s_branch {bar}")
    o(f"{labs[0]}:")
    o("s_waitcnt vmcnt(0)")
    o(f"s_branch {bar}")


def iteration(c, end_label):
    """Iteration whose tile sits at ring position c (= (C0 + j) & 3): reads V(j) from slot c and K(j+2) from slot c ^ 2, writes V(j+3)
    and K(j+5) -- or, past this block's last tile, the next block's tiles V'(j+3-n_it) and K'(j+5-n_it), up to K'(3).
    The tile barrier sits in gap 1 of segment 1: nothing ahead of it needs tile pair j (QK_B(j) runs on fragments read an
    iteration ago, the softmax on registers)."""
    par = c & 1
    sv, sk = (c + 3) & 3, (c + 1) & 3
    g0, g1 = VAR["dma_gaps"]
    stamp(6)
    e("s_waitcnt lgkmcnt(0)")
    # segment 1: softmax A(j) beside QK_B(j), PV_B(j-1)
    decide("A")
    stamp(2)

    def tile_v():
        e(f"s_add_u32 {sr(TV)}, {sr(J)}, 3")
        e(f"s_cmp_eq_u32 {sr(TV)}, {sr(NIT)}")             # the V stream moves on to the next block's tiles
        call("s_cbranch_scc1", SUBS["nextV"])
        e(f"s_cmp_eq_u32 {sr(TV)}, {sr(TAILV)}")
        call("s_cbranch_scc1", SUBS["tailV"])

    def adv_v():
        e(f"s_add_u32 {sr(VPTR)}, {sr(VPTR)}, {SLOT}")
        e(f"s_addc_u32 {sr(VPTR + 1)}, {sr(VPTR + 1)}, 0")

    segment("A", "B", par, 1 - par, v_reads(par, c), (2, 10),
            {g0: (LDSV, sv * SLOT, DVV, VPTR, TV, LIMV), g1: (LDSV, sv * SLOT + 4096, DVV + 1, VPTR, TV, LIMV)},
            {1: barrier_wait, 3: tile_v, 8: lambda: mask_check("B", J), g1 + 1: adv_v})
    stamp(3)
    e("s_waitcnt lgkmcnt(0)")
    # segment 2: softmax B(j) beside QK_A(j+1), PV_A(j)
    decide("B")
    stamp(4)

    def tile_k():
        e(f"s_add_u32 {sr(TK)}, {sr(J)}, 5")
        e(f"s_cmp_eq_u32 {sr(TK)}, {sr(NIT)}")
        call("s_cbranch_scc1", SUBS["nextK"])
        e(f"s_cmp_eq_u32 {sr(TK)}, {sr(TAILK)}")
        call("s_cbranch_scc1", SUBS["tailK"])

    def adv_k():
        e(f"s_add_u32 {sr(KPTR)}, {sr(KPTR)}, {SLOT}")
        e(f"s_addc_u32 {sr(KPTR + 1)}, {sr(KPTR + 1)}, 0")

    def loop_ctl():
        e(f"s_add_u32 {sr(J)}, {sr(J)}, 1")
        e(f"s_mov_b32 {sr(SLK)}, {sr(SLKS)}")             # past the first tile: the lazy threshold

    def loop_cmp():
        e(f"s_add_u32 {sr(J1)}, {sr(J1)}, 1")
        e(f"s_cmp_ge_u32 {sr(J)}, {sr(NIT)}")              # SCC holds until the branch behind the segment

    segment("B", "A", 1 - par, par, k_reads(par, c ^ 2), (0, 8),
            {g0: (LDSK, sk * SLOT, DVK, KPTR, TK, LIMK), g1: (LDSK, sk * SLOT + 4096, DVK + 1, KPTR, TK, LIMK)},
            {2: tile_k, 8: lambda: mask_check("A", J1), g1 + 1: adv_k, 18: loop_ctl, 19: loop_cmp})
    stamp(5)
    if DIAG[0]:
        e(f"s_cmp_ge_u32 {sr(J)}, {sr(NIT)}")
    e(f"s_cbranch_scc1 {end_label}")


def tile_dma(kind, tile):
    """Prologue form of one tile's two pieces (ring slot (C0 + tile) & 3), skipped past the last tile."""
    ptr, base, dv = (KPTR, LDSK, DVK) if kind == "K" else (VPTR, LDSV, DVV)
    skip = lab("nopro")
    e(f"s_cmp_le_u32 {sr(NIT)}, {tile}")
    e(f"s_cbranch_scc1 {skip}")
    e(f"s_cmp_eq_u32 {sr(TAILT)}, {tile}")
    call("s_cbranch_scc1", SUBS["tail" + kind])
    e(f"s_add_u32 {sr(TMP2)}, {sr(C0)}, {tile}")
    e(f"s_and_b32 {sr(TMP2)}, {sr(TMP2)}, 3")
    e(f"s_lshl_b32 {sr(TMP2)}, {sr(TMP2)}, 13")
    e(f"s_add_u32 {sr(TMP2)}, {sr(TMP2)}, {sr(base)}")
    for p in range(2):
        e(f"s_add_u32 m0, {sr(TMP2)}, {p * 4096}")
        e("s_nop 0")
        e(f"global_load_lds_dwordx4 {vr(dv + p)}, {sr(ptr, 2)}")
    e(f"{skip}:")
    e(f"s_add_u32 {sr(ptr)}, {sr(ptr)}, {SLOT}")
    e(f"s_addc_u32 {sr(ptr + 1)}, {sr(ptr + 1)}, 0")


def q_loads():
    for X in "AB":
        for kk in range(4):
            e(f"global_load_dwordx4 {ar(Q[X] + 4 * kk, 4)}, {vr(QOFF[X])}, {sr(QPTR, 2)} offset:{32 * kk}")


def pre_body(full=True):
    """The PRE statement (a workgroup's first block, and any block whose predecessor was too short to request its tiles): the
    block's Q fragments and its first six K / V tiles are requested BEFORE the previous block's epilogue runs, so their flight is
    hidden under it.  full = False (PREQ): the Q fragments only -- the tiles came in through the predecessor's K / V stream.
    Leaves: Q in a[64:95] (in flight); full: the DMA offsets in use in v146..149, KPTR / VPTR advanced past the tiles issued."""
    del L[:]
    del TAIL[:]
    for k in ("tailK", "tailV"):
        SUBS[k] = lab(k)
    done = lab("predone")
    # Q fragments first (vmcnt completes in order: any wait that covers a K/V piece covers them)
    q_loads()
    if full:
        e(f"s_lshr_b32 {sr(TAILT)}, {sr(SEQ)}, 6")
        e(f"s_add_u32 {sr(LDSV)}, {sr(LDSK)}, {NS * SLOT}")
        for i in range(2):
            e(f"v_mov_b32 {vr(DVK + i)}, {vr(VOFF + i)}")
            e(f"v_mov_b32 {vr(DVV + i)}, {vr(VOFF + i)}")
        # K0 K1 | V0 K2 | V1 K3
        for kind, tile in (("K", 0), ("K", 1), ("V", 0), ("K", 2), ("V", 1), ("K", 3)):
            tile_dma(kind, tile)
        e(f"s_branch {done}")
        sub_tail(SUBS["tailK"], DVK)
        sub_tail(SUBS["tailV"], DVV)
        for x in TAIL:
            e(x)
        e(f"{done}:")
    return list(L)


def zero16(reg, agpr):
    z = vr(T[0], 4)
    return f"v_mfma_f32_32x32x16_bf16 {ar(reg, 16) if agpr else vr(reg, 16)}, {z}, {z}, 0"


def sub_next(name, kind):
    """Subroutine: the stream of `kind` moves on to the next block's tiles (its pointer, the plain DMA offsets, the stream's limit and
    the index of its tail tile)."""
    ptr, nxt, dv, lim, tot, tail = (KPTR, KNEXT, DVK, LIMK, LKT, TAILK) if kind == "K" else (VPTR, VNEXT, DVV, LIMV, LVT, TAILV)
    o = TAIL.append
    o(f"{name}:")
    o(f"s_mov_b64 {sr(ptr, 2)}, {sr(nxt, 2)}")
    o(f"v_mov_b32 {vr(dv)}, {vr(VOFF)}")
    o(f"v_mov_b32 {vr(dv + 1)}, {vr(VOFF + 1)}")
    o(f"s_mov_b32 {sr(lim)}, {sr(tot)}")
    o(f"s_add_u32 {sr(tail)}, {sr(NIT)}, {sr(TAILT)}")
    o(f"s_setpc_b64 {sr(RET, 2)}")


def prologue(causal, entries):
    fin0 = lab("fin0")
    if DIAG[0]:
        for k in range(8):
            e(f"s_mov_b32 {sr(ACC0 + k)}, 0")
        e(f"s_memtime {sr(CUR, 2)}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_mov_b32 {sr(PREV)}, {sr(CUR)}")
    e(f"s_sub_u32 {sr(NITM1)}, {sr(NIT)}, 1")
    e(f"s_lshr_b32 {sr(TAILT)}, {sr(SEQ)}, 6")
    e(f"s_add_u32 {sr(LDSV)}, {sr(LDSK)}, {NS * SLOT}")
    # the two streams: this block's tiles, then up to V'(2) / K'(3) of the next block
    e(f"s_mov_b32 {sr(LIMV)}, {sr(NIT)}")
    e(f"s_mov_b32 {sr(LIMK)}, {sr(NIT)}")
    e(f"s_mov_b32 {sr(TAILV)}, {sr(TAILT)}")
    e(f"s_mov_b32 {sr(TAILK)}, {sr(TAILT)}")
    e(f"s_min_u32 {sr(LVT)}, {sr(NITN)}, 3")
    e(f"s_add_u32 {sr(LVT)}, {sr(LVT)}, {sr(NIT)}")
    e(f"s_min_u32 {sr(LKT)}, {sr(NITN)}, 4")
    e(f"s_add_u32 {sr(LKT)}, {sr(LKT)}, {sr(NIT)}")
    # lane constants
    e(f"v_mbcnt_lo_u32_b32 {vr(LANE)}, -1, 0")
    e(f"v_mbcnt_hi_u32_b32 {vr(LANE)}, -1, {vr(LANE)}")
    e(f"v_lshrrev_b32 {vr(HH8)}, 5, {vr(LANE)}")
    e(f"v_lshlrev_b32 {vr(HH8)}, 3, {vr(HH8)}")
    # zeroed state: the matrix pipe writes 16 registers per instruction (0 x 0 + 0)
    for i in range(4):
        e(f"v_mov_b32 {vr(T[0] + i)}, 0")
    e("s_nop 1")
    for X in "AB":
        e(zero16(NEGM[X], False))
    e(zero16(P["B"], False))
    for X in "AB":
        for dt in range(2):
            e(zero16(O[X][dt], True))
        e(zero16(LACC[X], True))
    for X in "AB":
        e(f"v_mov_b32 {vr(MREF[X])}, 0")
        e(f"v_mov_b32 {vr(SEEN[X])}, 0")
    for i in range(4):
        e(f"v_mov_b32 {vr(ONEV + i)}, {ONES}")
    # lazy-rescale threshold: 8 (log2 units); -inf on the first tile and for non-prefix masks.  Tiles from MFROM on hold hidden keys.
    e(f"s_mov_b32 {sr(SLK)}, {NINF}")
    e(f"s_mov_b32 {sr(SLKS)}, 0x41000000")
    e(f"s_bitcmp1_b32 {sr(FLAGS)}, 1")
    e(f"s_cselect_b32 {sr(SLKS)}, {sr(SLK)}, {sr(SLKS)}")
    e(f"s_bitcmp1_b32 {sr(FLAGS)}, 0")
    e(f"s_cselect_b32 {sr(MFROM)}, {sr(NITM1)}, {sr(NIT)}")                       # partial last tile
    e(f"s_bitcmp1_b32 {sr(FLAGS)}, 1")
    e(f"s_cselect_b32 {sr(MFROM)}, 0, {sr(MFROM)}")                               # not a prefix: every tile
    if causal:
        e(f"s_lshr_b32 {sr(TMP)}, {sr(Q0)}, 6")                                   # tiles from q0 / 64 on reach this wave's diagonal
        e(f"s_min_u32 {sr(MFROM)}, {sr(MFROM)}, {sr(TMP)}")
    e(f"s_cmp_eq_u32 {sr(NIT)}, 0")
    e(f"s_cbranch_scc1 {fin0}")
    # K(0), K(1) have landed.  Issue order behind K1 -- PRE: [Q first] K0 K1 | V0 K2 V1 K3 (8 pieces); a predecessor's stream:
    # K'0 K'1 | V'0 K'2 V'1 K'3 V'2 (10 pieces), then the 8 Q loads of PREQ (and the predecessor's epilogue stores, which only make
    # the counted wait err on the safe side)
    short, pre, wb, pz = lab("pshort"), lab("ppre"), lab("pwb"), lab("pz")
    e(f"s_cmp_ge_u32 {sr(NIT)}, 4")
    e(f"s_cbranch_scc0 {short}")
    e(f"s_cmp_eq_u32 {sr(PREF)}, 0")
    e(f"s_cbranch_scc1 {pre}")
    e("s_waitcnt vmcnt(18)")
    e(f"s_branch {wb}")
    e(f"{pre}:")
    e("s_waitcnt vmcnt(8)")
    e(f"s_branch {wb}")
    e(f"{short}:")
    e(f"s_cmp_eq_u32 {sr(PREF)}, 0")
    e(f"s_cbranch_scc1 {pz}")
    e("s_waitcnt vmcnt(8)")
    e(f"s_branch {wb}")
    e(f"{pz}:")
    e("s_waitcnt vmcnt(0)")
    e(f"{wb}:")
    e("s_barrier")
    # by ring phase: K(0) sits in slot C0 and goes to fragment set C0 & 1
    pro = [lab(f"pro{c}") for c in range(4)]
    for c in range(1, 4):
        e(f"s_cmp_eq_u32 {sr(C0)}, {c}")
        e(f"s_cbranch_scc1 {pro[c]}")
    for c in range(4):
        e(f"{pro[c]}:")
        for i in k_reads(c & 1, c) + k_reads((c + 1) & 1, (c + 1) & 3):
            e(i)
        e(zero16(VF(1 - (c & 1), 0, 0), True))       # the V fragments of "tile -1": PV_B(-1) adds 0 x 0
        e(zero16(VF(1 - (c & 1), 1, 0), True))
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")                               # every wave has K(0) in registers: its slot may take K(4)
        nopref, k4 = lab("nopref"), lab("k4")
        e(f"s_cmp_eq_u32 {sr(PREF)}, 0")
        e(f"s_cbranch_scc1 {nopref}")
        e("s_waitcnt vmcnt(0)")                      # the Q fragments of PREQ (behind the stream's pieces, ahead of nothing this wave needs)
        e(f"s_branch {k4}")
        e(f"{nopref}:")
        tile_dma("V", 2)
        e(f"{k4}:")
        tile_dma("K", 4)
        for i in range(8):
            e(mfma_qk("A", c & 1, i))
        e("s_nop 15")
        e(f"s_mov_b32 {sr(J)}, 0")
        e(f"s_mov_b32 {sr(J1)}, 1")
        mask_check("A", J)
        for i in max_chain("A"):
            e(i)
        stamp(0)
        e(f"s_branch {entries[c]}")
    return fin0


def body(causal, diag=False):
    del L[:]
    del TAIL[:]
    DIAG[0] = diag
    for k in ("slowA", "slowB", "maskA", "maskB", "tailK", "tailV", "nextK", "nextV"):
        SUBS[k] = lab(k)
    entries = [lab(f"entry{c}") for c in range(4)]
    fin0 = prologue(causal, entries)
    ends, fin, done = [lab("end0"), lab("end1")], lab("fin"), lab("done")
    for c in range(4):
        e(f"{entries[c]}:")
        iteration(c, ends[c & 1])
    e(f"s_branch {entries[0]}")
    e(f"{ends[1]}:")
    for i in range(12):
        e(mfma_pv("B", 1, i))
    e(f"s_branch {fin}")
    e(f"{ends[0]}:")
    for i in range(12):
        e(mfma_pv("B", 0, i))
    e(f"{fin}:")
    e("s_waitcnt lgkmcnt(0)")                        # (the stream's last pieces stay in flight: the next block waits for them)
    stamp(7)
    e(f"{fin0}:")
    e("s_nop 15")
    e("s_nop 3")
    e(f"s_branch {done}")
    sub_slow("A", SUBS["slowA"])
    sub_slow("B", SUBS["slowB"])
    sub_mask("A", SUBS["maskA"], causal)
    sub_mask("B", SUBS["maskB"], causal)
    sub_tail(SUBS["tailK"], DVK)
    sub_tail(SUBS["tailV"], DVV)
    sub_next(SUBS["nextK"], "K")
    sub_next(SUBS["nextV"], "V")
    for s in TAIL:
        e(s)
    e(f"{done}:")
    return list(L)


def clobbers():
    pinned_v = set(range(KA, KA + 4)) | set(range(VA, VA + 4)) | set(range(DVK, DVK + 4)) | set(range(VOFF, VOFF + 4)) | set(range(154, 158)) | {132, 133}
    out = [f"v{i}" for i in range(MAXV + 1) if i not in pinned_v]
    out += [f"a{i}" for i in range(96, 256) if i not in (LACC["A"], LACC["B"])]
    out += [f"s{i}" for i in range(MAXS + 1) if (i >= 52 and not 72 <= i <= 83 and not 86 <= i <= 92) or i in (TAILT, NITM1, LDSV)]
    out += ["vcc", "scc", "memory"]
    return out


def emit_macro(name, lines):
    return f"#define {name} \\\n" + " \\\n".join('    "' + s + '\\n\\t"' for s in lines) + "\n"


DIAG_VARIANTS = [
    {},                                             # 1: the product's stream, stamped
    {"no_dma": True},                               # 2: no LDS-DMA in the loop
    {"no_exp": True},                               # 3: no exponentials
    {"no_cvtmax": True},                            # 4: exponentials only
    {"no_softmax": True},                           # 5: MFMA + LDS + DMA only
    {"qk_agpr": True},                              # 6: QK^T results written to AGPRs (does a VGPR-destination MFMA slow the VALU?)
    {"no_dma": True, "no_softmax": True, "no_lds": True},   # 7: bare MFMA stream
    {"dma_form": "reg"},                            # 8: plain global_load_dwordx4 in place of the LDS-DMA piece (timing only)
    {"dma_form": "reg+write"},                      # 9: ... plus a ds_write_b128 per piece (timing only)
]


def main():
    parts = ["// GENERATED by tools/gen_attn_fwd64.py -- do not edit.\n"]
    for causal in (0, 1):
        lines = body(causal)
        parts.append(emit_macro(f"P2T_ATTN64_BODY_{causal}", lines))
        n_mfma = sum(1 for s in lines if s.startswith("v_mfma"))
        print(f"causal={causal}: {len(lines)} lines, {n_mfma} MFMAs")
    parts.append("#define P2T_ATTN64_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers()) + "\n")
    parts.append(emit_macro("P2T_ATTN64_PRE", pre_body(True)))
    parts.append(emit_macro("P2T_ATTN64_PREQ", pre_body(False)))
    parts.append('#define P2T_ATTN64_PRE_CLOBBERS "s48", "s50", "s54", "s61", "s62", "s84", "s85", "vcc", "scc", "memory"\n')
    # diagnostic builds (lab library only): per-phase cycle sums in s70..s77, returned as outputs; variants 2.. are ablations
    # (their results are wrong by construction: what they measure is what the removed part costs)
    parts.append("#ifdef P2T_LAB\n")
    base = dict(VAR)
    for k, over in enumerate(DIAG_VARIANTS):
        VAR.update(base)
        VAR.update(over)
        parts.append(emit_macro(f"P2T_ATTN64_BODY_DIAG{k + 1}", body(0, diag=True)))
    VAR.update(base)
    parts.append("#define P2T_ATTN64_CLOBBERS_DIAG P2T_ATTN64_CLOBBERS, \"s80\", \"s82\", \"s83\"\n#endif\n")
    with open(OUT, "w") as f:
        f.write("\n".join(parts))


if __name__ == "__main__":
    main()
