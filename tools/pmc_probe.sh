#!/bin/bash
# Hardware counters of the two 3x3 kernels on the probe layer (64->64 3x3, 1088x1920, with and without a residual):
# several rocprofv3 --pmc passes (kernel trace only, as the pool requires), one per counter group, summarised per kernel
# as averages per launch into gpurun_out/<name>.txt.  Run through gpurun.
# usage: tools/pmc_probe.sh [out-name]
NAME=${1:-r03_conv_probe_counters}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_tmp; rm -rf $OUT; mkdir -p $OUT
PASSES=(
 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
 "TCC_BUSY_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
 "GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace -d $OUT -o p$i -- python3 tools/conv_probe.py 64 64 3 fp16x3 > $OUT/p$i.log 2>&1 && echo "pass $i done: $P" || { echo "pass $i FAILED: $P"; tail -3 $OUT/p$i.log; }
done
python3 - "$NAME" <<'PY'
import sqlite3, glob, sys
out = open(f"gpurun_out/{sys.argv[1]}.txt", "w")
out.write("# rocprofv3 --pmc passes over `python3 tools/conv_probe.py 64 64 3 fp16x3` (64->64 3x3, 1088x1920; each kernel's launches\n"
          "# with and without a residual averaged together); averages per launch; *_sum counters are summed over the chip's\n"
          "# instances, SQ_* are per shader engine (32 records per launch) and tick every 4 cycles where the guide says so\n")
for db in sorted(glob.glob("gpurun_out/pmc_tmp/*_results.db")):
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events where name like '%conv_%' group by name, counter_name order by counter_name, name").fetchall()
    except Exception as e:
        out.write(f"# {db}: {e}\n"); continue
    for n, cn, k, a, d in rows:
        short = "conv_k32<3,4,8>" if "conv_k32" in n else ("conv_mfma<3,1,2,2,true>" if "conv_mfma" in n else n[:30])
        out.write(f"{cn:40s} {short:26s} records {k:6d}  avg {a:18.1f}   avg duration {d/1e3:8.1f} us\n")
out.close()
print(open(f"gpurun_out/{sys.argv[1]}.txt").read())
PY
rm -rf $OUT
