#!/bin/bash
# The two PMC passes of the exact-fp32 mode alone (profiles/pmc_traffic_fp32.json): part of tools/profile_round.sh, kept
# separately runnable.  usage: tools/profile_fp32_traffic.sh r04
set -u
R=${1:-r04}; OUT=gpurun_out/prof_$R; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="bench.py --no-cpu-baseline --no-parity-leg --no-extra-workloads --steps 2 --warmup 1 --gop-streams 1 --precision fp32"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o fetch32 -- python3 $B > $OUT/fetch32.log 2>&1 && echo fetch32 done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o write32 -- python3 $B > $OUT/write32.log 2>&1 && echo write32 done
python3 tools/rocprof_traffic.py $OUT/fetch32_results.db $OUT/write32_results.db 'conv_mfma<3, 1, _, 2, false' fp32 1080 1920 > $OUT/traffic_bench_fp32.json
cp profiles/pmc_traffic_fp32.json $OUT/pmc_traffic_fp32.json
rm -f $OUT/*.db
cat $OUT/pmc_traffic_fp32.json
