#!/bin/bash
# Evidence pass on the GPU box (run from the repo root through gpurun): bench lines, rocprofv3 kernel statistics and the
# PMC passes (one counter group per pass, never combined with a trace domain other than --kernel-trace) for one config.
#   bash tools/collect_profiles.sh cfg3 r02      -> gpurun_out/prof_r02_cfg3/{bench.json.log, stats/, pmc/, pmc_traffic.json}
#   PMC_ONLY=1 EXTRA="--batch 64" bash tools/collect_profiles.sh cfg3 r02_b64    (counter passes only, extra bench.py arguments)
set -o pipefail
CFG=${1:-cfg3}; TAG=${2:-r02}
OUT=gpurun_out/prof_${TAG}_${CFG}
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$PMC_ONLY" ]; then
echo "== bench $CFG"
timeout -k 10 400 python3 bench.py --config $CFG --steps 10 --warmup 3 $EXTRA > $OUT/bench.json.log 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
tail -c 600 $OUT/bench.json.log; echo
echo "== rocprofv3 kernel stats"
# (--no-overlap: per-kernel durations on ONE stream, comparable with the roofline block's single-stream event pass)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-batch64-check --no-overlap $EXTRA > $OUT/bench_under_rocprof.json.log 2> $OUT/rocprof.err || { echo "rocprof stats failed"; tail -5 $OUT/rocprof.err; exit 1; }
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/stats -name "*kernel_trace.csv" -delete                     # per-dispatch trace: large, the stats file is what is kept
head -5 $OUT/kernel_stats.csv
fi
for c in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
    name=$(echo $c | cut -d" " -f1)
    echo "== pmc $name"
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc/$name --output-format csv -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-batch64-check --no-overlap --event-steps 0 $EXTRA > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
done
python3 tools/pmc_traffic.py $OUT/pmc $OUT/pmc_traffic.json "$TAG $CFG $EXTRA" $CFG | head -60
find $OUT/pmc -name "*.csv" -size +20M -delete
du -sh $OUT
