"""Compact per-kernel summary of a rocprofv3 results database (ROCm 7.2 writes SQLite).

    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^()]*?>)?)", name)
    s = m.group(1) if m else name
    return s[:70]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats  ({path})")
    print(f"# total kernel time {total / 1e3:.1f} ms over {sum(r[1] for r in rows)} dispatches (durations in us)")
    print(f"{'kernel':72s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for n, calls, tot, avg, pct in rows[:40]:
        print(f"{short(n):72s} {calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
