#!/bin/bash
# usage: tools/ab.sh VARIANT...   -- bench.py once per variant, twice round-robin, on ONE box ("base" = in-tree library,
# "pN" = --gemm-policy N on the LAB build tools/build/libp2t_lab.so (p9 and p0 also work on the product library), anything else =
# tools/ab/libp2t_<VARIANT>.so through P2T_HIP_LIB); prints
# samples/s, ms/step, GEMM ms/step, GEMM TF/s.  AB_ARGS = extra bench.py arguments (e.g. "--config cfg5")
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    POL=""
    if [ "$v" = base ]; then unset P2T_HIP_LIB; elif [[ "$v" =~ ^p[0-9]+$ ]]; then export P2T_HIP_LIB=$PWD/tools/build/libp2t_lab.so; POL="--gemm-policy ${v#p}"; else export P2T_HIP_LIB=$PWD/tools/ab/libp2t_$v.so; fi
    timeout -k 10 300 python bench.py --no-batch64-check --no-cpu-baseline $POL $AB_ARGS > gpurun_out/ab_${v}_$rep.log 2>&1 || { echo "$v failed"; tail -5 gpurun_out/ab_${v}_$rep.log; exit 1; }
    tail -1 gpurun_out/ab_${v}_$rep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r['gemm_ms_per_step'], r['achieved'], r['attention_ms_per_step'])"
  done
done
