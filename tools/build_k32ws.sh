#!/bin/bash
# Developer build of the wave-specialised persistent experiment (tools/probes/conv_k32ws.hip) into
# tools/probes/variants/k32ws.so: the product's dcvc_conv2d_k32 is renamed dcvc_conv2d_k32_base in that build and the
# experiment's entry point takes the name (DCVC_K32_WS=0 routes everything to the base kernel again).
# usage: tools/build_k32ws.sh [extra -D flags for the experiment]   (after `make -C vcm_ts_amd/csrc`)
set -e
C=vcm_ts_amd/csrc; O=tools/probes/variants; mkdir -p $O/obj
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
one() {  # name source extra-flags...
  local name=$1 src=$2; shift 2
  /opt/rocm/bin/hipcc $F -cuid=dcvc_$name --offload-device-only $NOPK "$@" -c $src -o $O/obj/$name.hipfb
  /opt/rocm/bin/hipcc $F -cuid=dcvc_$name --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $O/obj/$name.hipfb "$@" -c $src -o $O/obj/$name.o
}
one k32base $C/conv_k32.hip -Ddcvc_conv2d_k32=dcvc_conv2d_k32_base
one k32ws tools/probes/conv_k32ws.hip "$@"
objs=""; for f in $C/build/*.o; do [ "$(basename $f)" = "conv_k32.o" ] || objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/${NAME:-k32ws}.so $objs $O/obj/k32base.o $O/obj/k32ws.o
echo built $O/${NAME:-k32ws}.so
