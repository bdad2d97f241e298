"""GPU occupancy of a run from a rocprofv3 --kernel-trace database: busy time (union of all kernel intervals), per-stream
busy time, and the idle gaps -- is a workload bound by the GPU or by the host that feeds it?
    python tools/rocprof_timeline.py results.db [skip_fraction]   (skip_fraction: leading part of the run to ignore, e.g. 0.5)"""
import sqlite3
import sys


def main(path, skip=0.0):
    c = sqlite3.connect(path)
    rows = c.execute("select start, end, stream_id, name from kernels order by start").fetchall()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[0] >= lo]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    span = t1 - t0
    busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
    for s, e, _, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"# {path}: {len(rows)} dispatches over {span / 1e6:.1f} ms (after skipping the first {skip:.0%} of the run)")
    print(f"GPU busy (union of kernel intervals): {busy / 1e6:.1f} ms = {busy / span:.1%} of the span; idle {sum(gaps) / 1e6:.1f} ms in {len(gaps)} gaps")
    per = {}
    for s, e, st, _ in rows:
        per[st] = per.get(st, 0) + (e - s)
    for st, v in sorted(per.items(), key=lambda kv: -kv[1]):
        print(f"  stream {st}: sum of kernel durations {v / 1e6:.1f} ms = {v / span:.1%} of the span")
    gaps.sort()
    if gaps:
        n = len(gaps)
        print(f"gap between consecutive busy intervals, us: median {gaps[n // 2] / 1e3:.1f}, 90th {gaps[int(n * 0.9)] / 1e3:.1f}, max {gaps[-1] / 1e3:.1f}; "
              f"gaps > 20 us: {sum(g > 20000 for g in gaps)} holding {sum(g for g in gaps if g > 20000) / 1e6:.1f} ms")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
