#!/bin/bash
# Round profiles of the other workloads and modes (run through gpurun): writes small summaries under gpurun_out/prof_wl
R=${1:-r04}
OUT=gpurun_out/prof_wl; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --workload train --steps 20 --warmup 5 > $OUT/${R}_bench_train_n1_fp16x3.json 2> $OUT/train.err
python3 bench.py --workload decode --steps 5 --warmup 1 > $OUT/${R}_bench_decode_n1_fp16x3.json 2> $OUT/decode.err
python3 bench.py --precision fp32 --steps 3 --warmup 1 --no-extra-workloads > $OUT/${R}_bench_n1_fp32.json 2> $OUT/fp32.err
rocprofv3 --kernel-trace --stats -d $OUT -o train -- python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $OUT/train_prof.log 2>&1
python3 tools/rocprof_summary.py $OUT/train_results.db > $OUT/${R}_train_step_kernel_stats.txt
rocprofv3 --kernel-trace --stats -d $OUT -o decode -- python3 bench.py --workload decode --steps 2 --warmup 1 > $OUT/decode_prof.log 2>&1
python3 tools/rocprof_summary.py $OUT/decode_results.db > $OUT/${R}_decode_kernel_stats.txt
rm -f $OUT/*.db
for f in $OUT/*.json; do echo $f; head -c 400 $f; echo; done
head -12 $OUT/${R}_train_step_kernel_stats.txt
