#!/bin/bash
# Round profiles on the GPU box (run through gpurun): kernel-trace stats and the two PMC traffic passes of the
# bench command, plus stats / traffic of both 3x3 kernels through the probe.  Writes under gpurun_out/prof_rNN.
# usage: tools/profile_round.sh r02
set -u
R=${1:-r04}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="bench.py --no-cpu-baseline --no-parity-leg --no-extra-workloads --steps 2 --warmup 1"
rocprofv3 --kernel-trace --stats -d $OUT -o stats1 -- python3 $B --gop-streams 1 > $OUT/stats1.log 2>&1 && echo stats1 done
rocprofv3 --kernel-trace --stats -d $OUT -o stats2 -- python3 $B > $OUT/stats2.log 2>&1 && echo stats2 done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o fetch -- python3 $B --gop-streams 1 > $OUT/fetch.log 2>&1 && echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o write -- python3 $B --gop-streams 1 > $OUT/write.log 2>&1 && echo write done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o fetch32 -- python3 $B --gop-streams 1 --precision fp32 > $OUT/fetch32.log 2>&1 && echo fetch32 done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o write32 -- python3 $B --gop-streams 1 --precision fp32 > $OUT/write32.log 2>&1 && echo write32 done
rocprofv3 --kernel-trace --stats -d $OUT -o probestats -- python3 tools/conv_probe.py 64 64 3 fp16x3 > $OUT/probestats.log 2>&1 && echo probestats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o probefetch -- python3 tools/conv_probe.py 64 64 3 fp16x3 > $OUT/probefetch.log 2>&1 && echo probefetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o probewrite -- python3 tools/conv_probe.py 64 64 3 fp16x3 > $OUT/probewrite.log 2>&1 && echo probewrite done

# summaries (small text / JSON) -- the databases are too big to travel back
python3 tools/rocprof_summary.py $OUT/stats1_results.db > $OUT/${R}_bench_gop_streams_1_kernel_stats.txt
python3 tools/rocprof_hbm.py $OUT/stats1_results.db $OUT/stats1.log > $OUT/${R}_hbm_kernels.txt
python3 tools/rocprof_summary.py $OUT/stats2_results.db > $OUT/${R}_bench_gop_streams_2_kernel_stats.txt
python3 tools/rocprof_pmc.py $OUT/fetch_results.db FETCH_SIZE > $OUT/${R}_fp16x3_pmc_fetch_size.txt
python3 tools/rocprof_pmc.py $OUT/write_results.db WRITE_SIZE > $OUT/${R}_fp16x3_pmc_write_size.txt
python3 tools/rocprof_traffic.py $OUT/fetch_results.db $OUT/write_results.db 'conv_k32<3, 4, 8>' fp16x3 1080 1920 > $OUT/traffic_bench.json
cp profiles/pmc_traffic_fp16x3.json $OUT/pmc_traffic_fp16x3.json
python3 tools/rocprof_traffic.py $OUT/fetch32_results.db $OUT/write32_results.db 'conv_mfma<3, 1, _, 2, false' fp32 1080 1920 > $OUT/traffic_bench_fp32.json
cp profiles/pmc_traffic_fp32.json $OUT/pmc_traffic_fp32.json
python3 tools/rocprof_summary.py $OUT/probestats_results.db > $OUT/${R}_conv_probe_64_kernel_stats.txt
python3 tools/rocprof_pmc.py $OUT/probefetch_results.db FETCH_SIZE > $OUT/${R}_conv_probe_64_pmc_fetch_size.txt
python3 tools/rocprof_pmc.py $OUT/probewrite_results.db WRITE_SIZE > $OUT/${R}_conv_probe_64_pmc_write_size.txt
rm -f $OUT/*.db
ls -la $OUT
