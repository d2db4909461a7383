#!/bin/bash
# PMC passes over the isolated attention kernel (tools/attn_only.py: 16 x 1024 tokens, 40 heads, head_dim 64, the towers' log2-scores
# form = csrc/attn_fwd64.hip), one counter group per pass, never combined with a trace domain other than --kernel-trace:
#   bash tools/collect_attn_pmc.sh r04      -> gpurun_out/pmc_attn_r04/ + gpurun_out/r04_pmc_attention.json
TAG=${1:-r04}
OUT=gpurun_out/pmc_attn_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    echo "== pass $i: $c"
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $OUT/p$i --output-format csv -- python3 tools/attn_only.py 4 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 - "$OUT" "gpurun_out/${TAG}_pmc_attention.json" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if "attn_fwd64_kernel" not in row.get("Kernel_Name", ""):
                continue
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
res = {"source": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 tools/attn_only.py 4 (attn_fwd64_kernel<false>, 16 x 1024 tokens, 40 heads, head_dim 64; 256 persistent workgroups)",
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs (= 32 x MFMA count); GRBM_GUI_ACTIVE summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x 2 = bytes read, MI355X_MICROARCH.md HBM section)",
       "counters": {k: {"avg_per_launch": v[0] / max(v[1], 1), "launches": v[1]} for k, v in sorted(acc.items())}}
c = {k: v["avg_per_launch"] for k, v in res["counters"].items()}
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c:
    res["matrix_pipe_busy_of_cu_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
if "SQ_ACTIVE_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
    res["instruction_active_of_wave_cycles"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    for k in ("VALU", "SCA", "LDS", "MISC"):
        if "SQ_ACTIVE_INST_" + k in c:
            res["active_share_" + k.lower()] = c["SQ_ACTIVE_INST_" + k] / c["SQ_ACTIVE_INST_ANY"]
if "FETCH_SIZE" in c:
    res["bytes_read_per_launch"] = 2.0 * 1024.0 * c["FETCH_SIZE"]
if "WRITE_SIZE" in c:
    res["bytes_written_per_launch"] = 1024.0 * c["WRITE_SIZE"]
import hashlib
h = hashlib.sha256()                       # the stamp of bench.py / tools/pmc_traffic.py: which kernel sources these counters belong to
src = os.path.join(os.path.dirname(os.path.abspath(d)), "..", "prot2text-v2-esm3_amd", "csrc")
src = src if os.path.isdir(src) else os.path.join("prot2text-v2-esm3_amd", "csrc")
for name in sorted(os.listdir(src)):
    if name.endswith((".hip", ".h", ".inc")):
        with open(os.path.join(src, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
res["kernel_src_sha16"] = h.hexdigest()[:16]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ("counters", "source", "units")}, indent=1))
PY
find $OUT -name "*.csv" -size +5M -delete
