#!/bin/bash
# Developer A/B builds: libdcvc_hip.so with extra -D flags for ONE translation unit, written to tools/probes/variants/<name>.so
# (git-ignored; travels to the GPU box).  Use with DCVC_HIP_LIB=tools/probes/variants/<name>.so.
# usage: tools/build_variant.sh name unit.hip -DFLAG=...   (run from the repo root after `make -C vcm_ts_amd/csrc`)
set -e
name=$1; unit=$2; shift 2
C=vcm_ts_amd/csrc; O=tools/probes/variants; mkdir -p $O/obj
SRC=$C/$unit
if [ "$unit" = "conv_k32.hip" ]; then  # the developer switches live in a patch, not in the product source
  make -s -C tools/probes conv_k32_dev.hip && cp tools/probes/conv_k32_dev.hip $O/obj/conv_k32.hip && SRC=$O/obj/conv_k32.hip
fi
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude"
/opt/rocm/bin/hipcc $F -cuid=dcvc_${unit%.hip} --offload-device-only -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c $SRC -o $O/obj/$name.hipfb
/opt/rocm/bin/hipcc $F -cuid=dcvc_${unit%.hip} --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $O/obj/$name.hipfb "$@" -c $SRC -o $O/obj/$name.o
objs=""; for f in $C/build/*.o; do [ "$(basename $f)" = "${unit%.hip}.o" ] || objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/$name.so $objs $O/obj/$name.o
echo built $O/$name.so
