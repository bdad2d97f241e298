"""profiles/rNN_parity_vs_reference.{txt,json} from the curves the GPU suite prints (DCVC_CURVE_OUT=file pytest -m gpu ...):
per-depth relative deviation from the REFERENCE's 1088x1920 fixtures -- free-running on the 5-bpp fixture, teacher-forced on
both fixtures, free-running on the low-rate fixture -- in both arithmetic modes.  bench.py quotes the JSON next to `value`.
    python tools/parity_summary.py gpurun_out/r4/parity_curves.txt profiles/r04_parity_vs_reference"""
import json
import re
import sys


def main(src, dst):
    text = open(src).read()
    out, cur = {}, None
    for line in text.splitlines():
        if line.startswith("#"):
            m = re.match(r"# (seq_\S+) (fp32|fp16x3), (teacher-forced|free-running)", line)
            if m:
                cur = out.setdefault(f"{m.group(1)} {m.group(3)}", {}).setdefault(m.group(2), [])
                continue
            m = re.match(r"# (fp32|fp16x3): picture", line)  # the free-running GOP-8 curve on the 5-bpp fixture (round 3's test)
            cur = out.setdefault("seq_1088x1920 free-running", {}).setdefault(m.group(1), []) if m else None
            continue
        if cur is None or not line.startswith("P"):
            continue
        m = re.match(r"P(\d): totals (\S+)\s+components (\S+)", line)
        if m:
            cur.append({"picture": int(m.group(1)), "totals": float(m.group(2)), "components": float(m.group(3))})
            continue
        m = re.match(r"P(\d): bpp (\S+)\s+mse (\S+)\s+PSNR (\S+)", line)
        if m:
            cur.append({"picture": int(m.group(1)), "totals": max(float(m.group(k)) for k in (2, 3, 4))})
    for case in out.values():  # a test that ran twice in one file: keep the last run
        for prec, rows in case.items():
            last = {}
            for r in rows:
                last[r["picture"]] = r
            case[prec] = [last[k] for k in sorted(last)]
    summary = {name: {prec: {"worst_total": max(r["totals"] for r in rows), "per_picture_totals": [r["totals"] for r in rows]}
                      for prec, rows in case.items() if rows} for name, case in out.items()}
    json.dump({"source": "GPU suite on MI355X, DCVC_CURVE_OUT; relative deviation of bpp / bits / mse / PSNR from the reference's "
                         "own run (tests/golden/seq_1088x1920*.npz), P1..P7", "cases": summary}, open(dst + ".json", "w"), indent=1)
    open(dst + ".txt", "w").write("# " + " ".join(sys.argv) + "\n" + text)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
