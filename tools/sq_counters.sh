#!/bin/bash
# SQ counters of the two 3x3 convolution kernels (conv_mfma, conv_k32) on the probe layer (64->64 3x3, 1088x1920), two PMC passes;
# writes a per-kernel summary to gpurun_out/sq_counters.txt (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/sq_tmp; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT -o p1 -- python3 tools/conv_probe.py 64 64 3 fp16x3 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $OUT -o p2 -- python3 tools/conv_probe.py 64 64 3 fp16x3 > $OUT/p2.log 2>&1
python3 - <<'PY' > gpurun_out/sq_counters.txt
import sqlite3, glob
print("# rocprofv3 --pmc (2 passes, SQ counters) -- python3 tools/conv_probe.py 64 64 3 fp16x3  (64->64 3x3, 1088x1920; averages per")
print("# launch over the probe's variants; one record per shader engine; SQ_WAVE_CYCLES-family counters tick every 4 cycles)")
for db in sorted(glob.glob("gpurun_out/sq_tmp/*_results.db")):
    c = sqlite3.connect(db)
    rows = c.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events where name like '%conv_%' group by name, counter_name order by name, counter_name").fetchall()
    for n, cn, k, a in rows:
        short = "conv_k32<3,4,8>" if "conv_k32" in n else ("conv_mfma<3,1,2,2,true>" if "conv_mfma" in n else n[:30])
        print(f"{short:26s} {cn:28s} records {k:6d} avg {a:16.1f}")
PY
rm -rf $OUT
cat gpurun_out/sq_counters.txt
