cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in 0 1 0 1; do DCVC_DEV=1 DCVC_XCD_PAD=$v python tools/conv_probe.py 64 64 3 fp16x3 2>/dev/null | grep "f32-act"; done
for v in 0 1; do
  DCVC_DEV=1 DCVC_XCD_PAD=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/xcd_$v -o f -- python3 tools/conv_probe.py 64 64 3 fp16x3 > /dev/null 2>&1
  python3 tools/rocprof_pmc.py gpurun_out/xcd_$v/f_results.db FETCH_SIZE | grep "conv_mfma" ; rm -rf gpurun_out/xcd_$v
done
