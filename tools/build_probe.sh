#!/bin/bash
# Diagnostic / experiment variants of libuavsal_hip.so for tools/k32_probe.py: only conv_gemm_k32.hip is recompiled
# (one library per line of VARIANTS: name + extra -D flags) and linked with the product objects into tools/_tmp/.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/iip_uavsal_saliency_amd/csrc
mkdir -p $R/tools/_tmp
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$R/include -I$C"
OTHERS="$C/conv_gemm.o $C/dw_conv.o $C/fused_ir.o $C/glue.o $C/post.o $C/plan.o $C/winograd.o"
VARIANTS=${VARIANTS:-"probe:-DUAVSAL_PROBE stamps:-DUAVSAL_K32_STAMPS"}
for v in $VARIANTS; do
  name=${v%%:*}; defs=$(echo ${v#*:} | tr ',' ' ')
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $C/conv_gemm_k32.hip -o $R/tools/_tmp/conv_gemm_k32_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_tmp/libuavsal_hip_$name.so $OTHERS $R/tools/_tmp/conv_gemm_k32_$name.o ) &
done
wait
ls -la $R/tools/_tmp/*.so
