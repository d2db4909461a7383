#!/bin/bash
# usage: tools/kres.sh file.hip [filter]  -- VGPR / spill / LDS per kernel (demangled)
cd /root/repo/prot2text-v2-esm3_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form=1 \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 | \
  awk '/Function Name:/{n=$NF} /VGPRs:/{v=$0; sub(/.*VGPRs: /,"",v); sub(/ .*/,"",v)} /VGPR Spill/{sp=$0; sub(/.*Spill: /,"",sp); sub(/ .*/,"",sp)} /ScratchSize/{sc=$0; sub(/.*: /,"",sc); sub(/ .*/,"",sc)} /LDS Size/{l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); print n, "vgpr="v, "spill="sp, "scratch="sc, "lds="l}' | \
  c++filt | grep -E "${2:-.}" | cut -c1-200
