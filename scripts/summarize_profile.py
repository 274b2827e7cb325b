#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small per-kernel tables kept under profiles/.

    python scripts/summarize_profile.py stats  <dir with *_kernel_stats.csv>        > profiles/rNN_kernel_stats.md
    python scripts/summarize_profile.py pmc    <dir with *_counter_collection.csv>... > profiles/rNN_pmc.md
    python scripts/summarize_profile.py db     <rocprofv3 *_results.db (rocpd sqlite)>   > profiles/rNN_kernel_stats.md

`pmc` averages every counter per kernel over its dispatches (rocprofv3 sums over XCDs / SEs: GRBM_GUI_ACTIVE
is the sum over the 8 XCDs).  FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE
counts 128-B requests as 64 B for wide streaming reads (MI355X_MICROARCH.md, HBM section), so the table shows
both the raw and the x2-corrected value.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 *
1024 SIMDs)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:64]


def stats(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for r in rows[:24]:
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")


def db(path):
    """rocprofv3's default output in ROCm 7.2 is a rocpd sqlite file; its `top_kernels` view is the --stats table."""
    import sqlite3

    c = sqlite3.connect(path)
    print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 28"):
        print(f"| `{short(name)}` | {calls} | {total / 1e3:.3f} | {avg:.1f} | {pct:.2f} |")


def pmc(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if (f, r["Dispatch_Id"]) not in seen:
                    seen.add((f, r["Dispatch_Id"]))
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    names = sorted({c for v in acc.values() for c in v})
    print("| kernel | dispatches | avg us (profiled) | " + " | ".join(names) + " | MFMA util | HBM read MB (x2) | HBM write MB |")
    print("|---|---|---|" + "---|" * (len(names) + 3))
    order = sorted(acc, key=lambda k: -sum(dur[k]))
    for k in order[:16]:
        v = acc[k]
        m = {c: sum(v[c]) / len(v[c]) for c in v}
        util = ""
        if m.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            util = f"{m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}"
        rd = f"{m['FETCH_SIZE'] * 2 * 1024 / 1e6:.1f}" if "FETCH_SIZE" in m else ""
        wr = f"{m['WRITE_SIZE'] * 1024 / 1e6:.1f}" if "WRITE_SIZE" in m else ""
        n = max(len(x) for x in v.values())
        print(f"| `{k}` | {n} | {sum(dur[k]) / len(dur[k]):.1f} | " + " | ".join(f"{m.get(c, float('nan')):.4g}" for c in names) +
              f" | {util} | {rd} | {wr} |")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "db":
        db(sys.argv[2])
    else:
        pmc(sys.argv[2:])
