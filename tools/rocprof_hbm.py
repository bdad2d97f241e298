"""HBM-bound kernels of a rocprofv3 --kernel-trace database, per (kernel, grid size): launches and average duration.
The bench line's roofline.hbm_kernels carries the same kernels per SHAPE with their algorithmic bytes (HIP events);
this is the profiler's view of the same command, to check the durations against.
    python tools/rocprof_hbm.py results.db [bench_line.json]"""
import json
import re
import sqlite3
import sys

NAMES = ("warp_shfl", "warp_scalar", "warp_vec4", "up2_kernel", "down2_kernel", "dual_prior_kernel", "nchw_to_nhwc", "nhwc_to_nchw",
         "copy_channels", "scale_channels", "round_symbols", "maxpool2", "se_gate", "channel_finish")


def main(path, bench=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gcols = [k for k in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z") if k in cols][:3]
    sel = ", ".join(gcols) if gcols else "0, 0, 0"
    rows = c.execute(f"select name, {sel}, start, end from kernels").fetchall()
    agg = {}
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", r[0]))
        m = re.match(r"([A-Za-z0-9_:]+(<[^()]*?>)?)", name)
        name = m.group(1) if m else name
        if not any(n in name for n in NAMES):
            continue
        key = (name, tuple(r[1:-2]))
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (r[-1] - r[-2]) / 1e3
    print(f"# rocprofv3 --kernel-trace ({path}): HBM-bound kernels per (kernel, grid {'x'.join(gcols) or 'n/a'}); durations in us")
    print(f"{'kernel':28s} {'grid':>22s} {'calls':>7s} {'avg_us':>10s} {'total_us':>12s}")
    for (name, grid), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:28s} {'x'.join(str(g) for g in grid):>22s} {n:7d} {tot / n:10.2f} {tot:12.1f}")
    if bench:
        line = [ln for ln in open(bench) if ln.startswith("{")][-1]
        d = json.loads(line)
        print("\n# the same kernels in the bench line of the same command (HIP events, per shape; algorithmic bytes: each operand once)")
        print(f"{'kernel':22s} {'shape':>18s} {'calls':>6s} {'avg_us':>9s} {'MB':>9s} {'TB/s':>7s} {'of 8 TB/s':>9s}")
        for h in d["roofline"].get("hbm_kernels", []):
            print(f"{h['kernel']:22s} {h['shape']:>18s} {h['launches']:6d} {h['avg_us']:9.2f} {h['algorithmic_bytes'] / 1e6:9.2f} "
                  f"{h['tb_per_s'] if h['tb_per_s'] is not None else 0:7.3f} {h['frac_of_8_tb_s'] if h['frac_of_8_tb_s'] is not None else 0:9.3f}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
