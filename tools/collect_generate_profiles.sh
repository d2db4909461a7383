#!/bin/bash
# Evidence pass for `generate` on the GPU box (run from the repo root through gpurun): bench lines at 8 / 32 / 1 prompts and with beams,
# rocprofv3 kernel statistics + the kernel-time / gap split of the decode steps, and the three PMC passes over a short run.
#   bash tools/collect_generate_profiles.sh r03   -> gpurun_out/prof_r03_generate/{generate_bench.log, kernel_stats.csv, trace_gaps.log, pmc_traffic.json}
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_${TAG}_generate
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/generate_bench.log
for args in "8 64" "32 64" "1 64" "8 32 4" "8 64 1 fp8" "32 64 1 fp8"; do
    timeout -k 10 300 python3 tools/generate_bench.py $args 2>$OUT/bench.err | grep -E "^generate|^prompt" >> $OUT/generate_bench.log || { echo "generate_bench $args failed"; tail -5 $OUT/bench.err; exit 1; }
done
cut -c1-260 $OUT/generate_bench.log
timeout -k 10 200 python3 tools/microbench.py skinny m8 m32 2>&1 | grep "^skinny" > $OUT/microbench_skinny.log || { echo "microbench failed"; exit 1; }
timeout -k 10 200 python3 tools/decode_attn_bench.py 8 1088 2>&1 | grep "^decode" > $OUT/decode_attn_bench.log && timeout -k 10 200 python3 tools/decode_attn_bench.py 32 1088 2>&1 | grep "^decode" >> $OUT/decode_attn_bench.log
cat $OUT/microbench_skinny.log $OUT/decode_attn_bench.log
echo "== rocprofv3 kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/generate_bench.py 8 32 > $OUT/bench_under_rocprof.log 2> $OUT/rocprof.err || { echo "rocprof failed"; tail -5 $OUT/rocprof.err; exit 1; }
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
f=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $f | tee $OUT/trace_gaps.log
rm -f $f
for c in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
    name=$(echo $c | cut -d" " -f1)
    echo "== pmc $name"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc/$name --output-format csv -- python3 tools/generate_bench.py 8 12 > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
done
python3 tools/pmc_traffic.py $OUT/pmc $OUT/pmc_traffic.json "$TAG generate 8 prompts" generate/8 | head -40
find $OUT/pmc -name "*.csv" -size +20M -delete
du -sh $OUT
