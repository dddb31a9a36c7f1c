"""Print every counter of every sparf kernel found in a directory of rocprofv3 --pmc passes
(average of the last 5 launches = the timed repetitions of tools/kernel_bench.py).
Usage: python tools/pmc_raw.py gpurun_out/pmc_dbg [out.csv]"""
import collections
import csv
import glob
import os
import sys


def main(src, dst=None):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "sparf::" in r["Kernel_Name"]:
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for k in vals.values() for c in k})
    rows = []
    for k in sorted(vals):
        short = k.replace("sparf::", "").split("(")[0].replace("void ", "")
        rows.append([short] + [(sum(vals[k][c][-5:]) / len(vals[k][c][-5:])) if c in vals[k] else "" for c in names])
    if dst:
        with open(dst, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel"] + names)
            w.writerows(rows)
    for r in rows:
        print(r[0])
        for c, v in zip(names, r[1:]):
            if v != "":
                print(f"    {c:36s} {v:16.0f}")


if __name__ == "__main__":
    main(*sys.argv[1:])
