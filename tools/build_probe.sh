#!/bin/bash
# Diagnostic / experiment variants of libuavsal_hip.so for tools/k32_probe.py / tools/fused_probe.py: only $SRC.hip
# (default conv_gemm_k32) is recompiled (one library per entry of VARIANTS: name:-Dflag,-Dflag) and linked with the
# product objects into tools/_tmp/.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/iip_uavsal_saliency_amd/csrc
mkdir -p $R/tools/_tmp
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$R/include -I$C"
SRC=${SRC:-conv_gemm_k32}
OTHERS=""
for o in conv_gemm conv_gemm_k32 dwproj dw_conv fused_ir fused_mid glue post plan winograd; do [ $o = $SRC ] || OTHERS="$OTHERS $C/$o.o"; done
VARIANTS=${VARIANTS:-"probe:-DUAVSAL_PROBE stamps:-DUAVSAL_K32_STAMPS"}
for v in $VARIANTS; do
  name=${v%%:*}; defs=$(echo ${v#*:} | tr ',' ' ')
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $C/$SRC.hip -o $R/tools/_tmp/${SRC}_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_tmp/libuavsal_hip_$name.so $OTHERS $R/tools/_tmp/${SRC}_$name.o ) &
done
wait
ls -la $R/tools/_tmp/*.so
