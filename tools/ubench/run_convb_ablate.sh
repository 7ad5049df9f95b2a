#!/bin/bash
# builds ablation variants of the split-bf16 conv kernel and runs them (on the GPU box).  VARIANTS = list of -D sets
cd "$(dirname "$0")"
IFS=';' read -ra VS <<< "${VARIANTS:-;-DCONVB_NO_PRODUCER;-DCONVB_NO_CONSUMER}"
i=0
for v in "${VS[@]}"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -I ../../include -I ../../codd_amd/csrc -mllvm -amdgpu-mfma-vgpr-form \
    -Wno-unused-result -Wno-unused-value -Wno-pass-failed convb_ablate.hip -o /tmp/convb_ablate_$i || exit 1
  echo -n "[$v] "; /tmp/convb_ablate_$i "$@" | grep -E "us|rc"
done
