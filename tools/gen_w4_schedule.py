#!/usr/bin/env python3
"""Writes the hand-placed instruction order of one 64-deep stage of csrc/gemm_w4.hip (between the GENERATED markers).

A stage is 2 x 32 MFMA pairs (K half 0 from (xa, wa), K half 1 from (xb, wb)); around the pairs go the 16 fragment reads of
K half 1 (first half), the 16 reads of the next stage's K half 0 (second half), the 16 LDS-DMA pieces of stage s+2 and the
two waits.  DMA pieces cost the issuing wave ~60+ cycles each when they queue behind each other (MI355X_MICROARCH.md, LDS-DMA
piece issue cost), so they are spread over the whole window in which their buffer is free and their data is not yet needed:
from the mid-stage barrier to the end of the stage, one piece every ~3 pairs.

    python tools/gen_w4_schedule.py            # rewrite the block in place
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "prot2text-v2-esm3_amd", "csrc", "gemm_w4.hip")

READ_ORDER = ["W0", "W1", "X0", "W2", "W3", "X1", "W4", "W5", "X2", "W6", "W7", "X3", "X4", "X5", "X6", "X7"]
DMA_FIRST = [20, 22, 25, 28, 31]                       # pairs of the first half that carry a DMA piece (after the barrier behind pair 19)
DMA_SECOND = [1, 4, 7, 10, 13, 16, 19, 22, 25, 28, 31]  # pairs of the second half
WAIT_AFTER_SECOND = 3                                   # vmcnt + barrier behind this pair of the second half
READS_SECOND_FROM = 4                                   # next-stage K-half-0 reads behind pairs 4 .. 19


def main():
    assert len(DMA_FIRST) + len(DMA_SECOND) == 16
    issued_before_wait = len(DMA_FIRST) + sum(1 for p in DMA_SECOND if p <= WAIT_AFTER_SECOND)
    out = []
    q = 0
    line = []

    def flush():
        if line:
            out.append("        " + " ".join(line))
            line.clear()

    for p in range(32):                                 # first half
        if p < 16:
            r = READ_ORDER[p]
            line.append(f"P2T_W4_R1{r[0]}({r[1]})")
        if p in DMA_FIRST:
            line.append(f"P2T_W4_G({q})")
            q += 1
        line.append(f"P2T_W4_PAIR(FI, wa, xa, {p})")
        if p % 4 == 3:
            flush()
        if p == 19:
            flush()
            out.append('        asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");')
    flush()
    for p in range(32):                                 # second half
        if READS_SECOND_FROM <= p < READS_SECOND_FROM + 16:
            r = READ_ORDER[p - READS_SECOND_FROM]
            line.append(f"P2T_W4_R0{r[0]}({r[1]})")
        if p in DMA_SECOND:
            line.append(f"P2T_W4_G({q})")
            q += 1
        line.append(f"P2T_W4_PAIR(F, wb, xb, {p})")
        if p % 4 == 3:
            flush()
        if p == WAIT_AFTER_SECOND:
            flush()
            out.append("        P2T_W4_WAIT_NEXT_STAGE")
    flush()
    assert q == 16
    body = "\n".join(out)
    with open(PATH) as f:
        s = f.read()
    a = s.index("        // GENERATED (tools/gen_w4_schedule.py) BEGIN")
    b = s.index("        // GENERATED END")
    s = s[:a] + f"        // GENERATED (tools/gen_w4_schedule.py) BEGIN -- DMA pieces issued before the second-half wait: {issued_before_wait}\n" + body + "\n" + s[b:]
    s = re.sub(r"constexpr int kIssuedBeforeWait = \d+;", f"constexpr int kIssuedBeforeWait = {issued_before_wait};", s)
    with open(PATH, "w") as f:
        f.write(s)
    print("pieces before the wait:", issued_before_wait)


if __name__ == "__main__":
    main()
