#!/bin/bash
# PMC passes on one tools/conv_probe.py signature (run through gpurun):  tools/probe_counters.sh TAG <probe args...>
# writes gpurun_out/probe_counters_TAG.txt: per kernel the average launch time, FETCH/WRITE_SIZE and SQ counters
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pc_tmp; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o f -- python3 tools/conv_probe.py "$@" > $OUT/f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o w -- python3 tools/conv_probe.py "$@" > $OUT/w.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT -o p1 -- python3 tools/conv_probe.py "$@" > $OUT/p1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $OUT -o p2 -- python3 tools/conv_probe.py "$@" > $OUT/p2.log 2>&1
python3 - "$TAG" "$@" <<'PY' > gpurun_out/probe_counters_$TAG.txt
import sqlite3, glob, sys
print("# tools/probe_counters.sh", " ".join(sys.argv[1:]))
print("# per launch averages; FETCH_SIZE/WRITE_SIZE in the counter's native unit (KB; FETCH_SIZE is doubled for gfx950 when quoted as bytes);")
print("# SQ_* are sums over one record per shader engine / XCD; *_CYCLES-family counters tick every 4 cycles")
for db in sorted(glob.glob("gpurun_out/pc_tmp/*_results.db")):
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events where name like '%conv_%' group by name, counter_name order by name, counter_name").fetchall()
    except Exception:
        rows = [(n, cn, k, a, None) for n, cn, k, a in c.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events where name like '%conv_%' group by name, counter_name order by name, counter_name").fetchall()]
    for n, cn, k, a, d in rows:
        print(f"{n[:44]:44s} {cn:28s} records {k:6d} avg {a:16.1f}" + (f"  avg_ns {d:10.0f}" if d else ""))
PY
tail -5 $OUT/f.log >> gpurun_out/probe_counters_$TAG.txt
rm -rf $OUT
cat gpurun_out/probe_counters_$TAG.txt
