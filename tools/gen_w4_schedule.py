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
# schedule = (reads per pair in the first half, pair behind which the mid-stage lgkmcnt(0) + barrier sits, DMA pairs of the first
#             half, DMA pairs of the second half, pair of the second half behind which the vmcnt wait + barrier sits,
#             first pair of the second half that carries a read, reads per pair there)
SCHEDULES = {
    0: dict(rpp1=1, bar1=19, dma1=[20, 22, 25, 28, 31], dma2=[1, 4, 7, 10, 13, 16, 19, 22, 25, 28, 31], wait2=3, rd2=4, rpp2=1),
    1: dict(rpp1=2, bar1=10, dma1=[11, 14, 17, 20, 23, 26, 29], dma2=[0, 3, 6, 9, 12, 15, 18, 21, 24], wait2=3, rd2=4, rpp2=1),
}


def emit(sc):
    out, line = [], []
    q = 0
    nread = 0

    def flush():
        if line:
            out.append("            " + " ".join(line))
            line.clear()

    assert len(sc["dma1"]) + len(sc["dma2"]) == 16 and min(sc["dma1"]) > sc["bar1"]
    for p in range(32):                                 # first half
        for _ in range(sc["rpp1"]):
            if nread < 16:
                r = READ_ORDER[nread]
                line.append(f"P2T_W4_R1{r[0]}({r[1]})")
                nread += 1
        if p in sc["dma1"]:
            line.append(f"P2T_W4_GPAIR({q}, FI, wa, xa, {p})")
            q += 1
        else:
            line.append(f"P2T_W4_PAIR(FI, wa, xa, {p})")
        if p % 4 == 3:
            flush()
        if p == sc["bar1"]:
            assert nread == 16
            flush()
            out.append('            asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");')
    flush()
    nread = 0
    for p in range(32):                                 # second half
        if p >= sc["rd2"]:
            for _ in range(sc["rpp2"]):
                if nread < 16:
                    r = READ_ORDER[nread]
                    line.append(f"P2T_W4_R0{r[0]}({r[1]})")
                    nread += 1
        if p in sc["dma2"]:
            line.append(f"P2T_W4_GPAIR({q}, F, wb, xb, {p})")
            q += 1
        else:
            line.append(f"P2T_W4_PAIR(F, wb, xb, {p})")
        if p % 4 == 3:
            flush()
        if p == sc["wait2"]:
            flush()
            out.append("            P2T_W4_WAIT_NEXT_STAGE")
    flush()
    assert q == 16 and nread == 16 and sc["rd2"] > sc["wait2"]
    before = len(sc["dma1"]) + sum(1 for p in sc["dma2"] if p <= sc["wait2"])
    return "\n".join(out), before


def main():
    # order 1 is the product's (+0.5 % in-step over order 0, profiles/r02_*); order 0 is compiled into the lab build only
    bodies, counts = {}, {}
    for k in sorted(SCHEDULES):
        bodies[k], counts[k] = emit(SCHEDULES[k])
    text = ("        if constexpr (SCHED == 1) {\n" + bodies[1] + "\n        }\n#ifdef P2T_LAB\n        else if constexpr (SCHED == 0) {\n"
            + bodies[0] + "\n        }\n#endif\n")
    with open(PATH) as f:
        s = f.read()
    a = s.index("        // GENERATED (tools/gen_w4_schedule.py) BEGIN")
    b = s.index("        // GENERATED END")
    s = s[:a] + "        // GENERATED (tools/gen_w4_schedule.py) BEGIN\n" + text + s[b:]
    s = re.sub(r"constexpr int kIssuedBeforeWait = [^;]*;", f"constexpr int kIssuedBeforeWait = SCHED == 0 ? {counts[0]} : {counts[1]};", s)
    with open(PATH, "w") as f:
        f.write(s)
    print("pieces before the wait:", counts)


if __name__ == "__main__":
    main()
