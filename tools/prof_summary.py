"""Turn a rocprofv3 results .db (rocpd sqlite) into a small CSV kernel summary for profiles/.
Usage: python tools/prof_summary.py gpurun_out/prof/X_results.db profiles/X_kernel_stats.csv"""
import csv
import sqlite3
import sys


def main(db, out):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(top_kernels)")]
    rows = list(cur.execute("select * from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["name", "calls", "total_ns", "avg_ns", "pct"])
        for r in rows:
            d = dict(zip(cols, r))
            w.writerow([d["name"][:160], d["total_calls"], f'{d["total_duration"] * 1e3:.0f}', f'{d["average"] * 1e3:.0f}', f'{d["percentage"]:.3f}'])
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
