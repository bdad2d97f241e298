"""Per-kernel average of one PMC counter from a rocprofv3 results database.
    python tools/rocprof_pmc.py gpurun_out/prof2/fetch_results.db FETCH_SIZE"""
import re, sqlite3, sys
from rocprof_summary import short

def main(path, counter):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(counter_value), avg(counter_value), avg(duration) from pmc_events where counter_name=? group by name order by sum(counter_value) desc", (counter,)).fetchall()
    print(f"# {counter} per kernel ({path}); values are the counter's native unit (KB for FETCH_SIZE / WRITE_SIZE)")
    print(f"{'kernel':60s} {'calls':>6s} {'sum':>14s} {'avg/launch':>12s} {'avg_us':>9s}")
    for n, k, s, a, d in rows[:25]:
        print(f"{short(n):60s} {k:6d} {s:14.0f} {a:12.1f} {d/1e3:9.1f}")

if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    main(sys.argv[1], sys.argv[2])
