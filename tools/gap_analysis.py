"""Idle time between kernels of a rocprofv3 --kernel-trace CSV: how much of a step the GPU waits for the host, and behind which
kernels.  Usage: python tools/gap_analysis.py <kernel_trace.csv> [min_gap_us]"""
import collections
import csv
import sys


def main(path, min_gap_us=15.0):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    rows = rows[len(rows) // 3:]                       # skip warm-up
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    print(f"{len(rows)} kernels, span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({busy / span * 100:.1f} %), idle {100 - busy / span * 100:.1f} %")
    gaps = collections.defaultdict(lambda: [0, 0.0])
    last_end, last_name = rows[0][1], rows[0][2]
    for s, e, n in rows[1:]:
        g = (s - last_end) / 1e3
        if g >= min_gap_us:
            k = (last_name.split("(")[0][-60:], n.split("(")[0][-60:])
            gaps[k][0] += 1
            gaps[k][1] += g
        if e > last_end:
            last_end, last_name = e, n
    tot = sum(v[1] for v in gaps.values())
    print(f"gaps >= {min_gap_us} us: {tot / 1e3:.1f} ms = {tot * 1e3 / span * 100:.1f} % of the span")
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {v[1] / 1e3:7.2f} ms in {v[0]:5d} gaps (avg {v[1] / v[0]:6.1f} us)  after {k[0]}  ->  before {k[1]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 15.0)
