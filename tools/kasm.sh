#!/bin/bash
# Compile one csrc/*.hip to gfx950 assembly with the product's flags:  tools/kasm.sh attn_fwd64 [out.s]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/prot2text-v2-esm3_amd/csrc/$1.hip"
OUT="${2:-/tmp/$1.s}"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on \
    -mllvm -amdgpu-mfma-vgpr-form=1 -S --cuda-device-only "$SRC" -o "$OUT" 2>&1 | grep -v "hip-link" || true
grep -n "^_Z.*:$\|NumVgprs\|NumAgprs\|ScratchSize\|; Occupancy\|LDSByteSize\|NumSgprs" "$OUT" || true
