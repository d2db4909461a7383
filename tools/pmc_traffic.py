#!/usr/bin/env python3
"""Aggregate rocprofv3 PMC passes into profiles/r01_pmc_traffic.json (HBM-side bytes per launch, clocks, MFMA-busy).

Collect on the GPU box, one counter group per pass (MI355X_MICROARCH.md, "rocprofv3 PMC slots": FETCH_SIZE and
WRITE_SIZE do not fit one pass; never combined with the hip/hsa trace domains):

    for c in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
        rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc/$(echo $c | cut -d" " -f1) --output-format csv -- \
            python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch64-check
    done
    python3 tools/pmc_traffic.py gpurun_out/pmc profiles/r01_pmc_traffic.json "<label>"

Corrections applied (same guide, HBM section): FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads (LDS-DMA and 16-B global loads alike), so reads = 2 x FETCH_SIZE x 1024;
WRITE_SIZE is exact for 16-B stores.  Infinity-Cache hits are included (the counters sit on the L2's fabric side).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_src_sha16():
    """Stamp of the kernel sources the counters were taken on (bench.py recomputes it and reports `traffic: null` when the
    sources have changed since): sha256 over csrc/*.hip and *.h in name order."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "prot2text-v2-esm3_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".inc")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


FAMILIES = {"gemm_nt_mfma": "gemm_nt_mfma*", "gemm_nt_w4": "gemm_nt_mfma*",     # one family: the bf16 MFMA GEMM in its eight- and four-wave forms
            "gemm_nt_fp8": "gemm_nt_fp8*", "attn_mfma_kernel": "attn_mfma_kernel", "attn_fwd64_kernel": "attn_fwd64_kernel", "norm_kernel": "norm_kernel",
            "gemm_skinny_kernel": "gemm_skinny_kernel", "attn_decode_mfma_kernel": "attn_decode_mfma_kernel"}   # the decode step (generate)


def family(name):
    for key, fam in FAMILIES.items():
        if key in name:
            return fam
    return None


def read_pass(directory):
    """{family: {counter: [sum, launches]}} and {family: {dispatch: duration_ns}} from every counter CSV below directory."""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                fam = family(row.get("Kernel_Name", ""))
                if fam is None:
                    continue
                a = acc[fam][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def main():
    root, out_path = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else ""
    cfg = sys.argv[4] if len(sys.argv) > 4 else "cfg3"
    sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
    from p2t_hip import specs
    m = re.search(r"--batch\s+(\d+)", label)
    # config / per-GPU batch (bench.py's default unless the label carries --batch N); any other name (e.g. "generate/8") is kept as it is
    workload = f"{cfg}/{int(m.group(1)) if m else specs.CONFIGS[cfg][3]}" if cfg in specs.CONFIGS else cfg
    merged = defaultdict(dict)
    for sub in sorted(os.listdir(root)):
        for fam, counters in read_pass(os.path.join(root, sub)).items():
            for cname, (total, n) in counters.items():
                merged[fam][cname] = {"avg": total / max(n, 1), "launches": n}
    out = {"source": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-batch64-check; aggregated by tools/pmc_traffic.py",
           "label": label, "kernel_src_sha16": kernel_src_sha16(), "workload": workload,
           "units": "FETCH_SIZE / WRITE_SIZE in KiB; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE "
                    "reports half of wide coalesced reads; Infinity-Cache hits included)",
           "kernels": {}}
    for fam, c in merged.items():
        k = {"launches": max(v["launches"] for v in c.values())}
        for cname, v in c.items():
            k[cname.lower() + "_avg"] = v["avg"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            k["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"]["avg"] + c["WRITE_SIZE"]["avg"]) * 1024.0
        if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c:
            # MFMA-busy is tallied per SIMD (4 per CU), CU-busy per CU: fraction of CU-busy cycles with the matrix pipe busy
            k["mfma_busy_frac_of_cu_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / max(4.0 * c["SQ_BUSY_CU_CYCLES"]["avg"], 1.0)
        out["kernels"][fam] = k
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
