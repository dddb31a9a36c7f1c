"""Summarise the PMC passes of tools/pmc_profile.sh into profiles/<tag>_pmc_<prec>.json/.csv.

Per kernel (last launches = the timed repetitions of tools/kernel_bench.py on the 786 432-row
fine pass): HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (KiB units; FETCH_SIZE doubled
on gfx950 for wide coalesced reads, MI355X_MICROARCH.md section HBM), MFMA utilisation =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), LDS conflict share.
Usage: python tools/pmc_summary.py gpurun_out/pmc_r01 profiles/r01_pmc_bf16"""
import collections
import csv
import glob
import json
import os
import sys


def main(src, dst):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "sparf::" in r["Kernel_Name"]:
                vals[r["Kernel_Name"]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {}
    for k, c in vals.items():
        m = lambda name: (sum(v for v, _ in c[name][-5:]) / len(c[name][-5:])) if name in c else None
        dur = lambda name: (sum(d for _, d in c[name][-5:]) / len(c[name][-5:])) if name in c else None
        e = {"launches_seen": max(len(v) for v in c.values())}
        if m("FETCH_SIZE") is not None:
            e["hbm_read_bytes"] = m("FETCH_SIZE") * 1024 * 2
            e["hbm_read_bytes_uncorrected"] = m("FETCH_SIZE") * 1024
        if m("WRITE_SIZE") is not None:
            e["hbm_write_bytes"] = m("WRITE_SIZE") * 1024
        if m("SQ_VALU_MFMA_BUSY_CYCLES") is not None and m("GRBM_GUI_ACTIVE"):
            e["mfma_busy_cycles"] = m("SQ_VALU_MFMA_BUSY_CYCLES")
            e["mfma_util"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * m("GRBM_GUI_ACTIVE") / 8)
            e["duration_ns_under_pmc"] = dur("GRBM_GUI_ACTIVE")
        if m("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_share"] = m("SQ_LDS_BANK_CONFLICT") / m("SQ_LDS_IDX_ACTIVE")
            e["wave_wait_share"] = m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES")
            if m("GRBM_GUI_ACTIVE"):      # LDS-array cycles per CU over the kernel's cycles (256 CUs; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
                e["lds_active_share"] = m("SQ_LDS_IDX_ACTIVE") / (256 * m("GRBM_GUI_ACTIVE") / 8)
        if m("SQ_WAIT_INST_ANY") is not None and m("SQ_ACTIVE_INST_ANY"):
            tot = m("SQ_WAIT_INST_ANY") + m("SQ_ACTIVE_INST_ANY")
            e["issue_stall_share_of_issue_cycles"] = m("SQ_WAIT_INST_ANY") / tot
            if m("SQ_WAIT_INST_LDS") is not None:
                e["lds_issue_stall_share_of_issue_cycles"] = m("SQ_WAIT_INST_LDS") / tot
        out[k] = e
    out["_meta"] = {"rows": int(os.environ.get("SPARF_PMC_ROWS", 786432)), "source": "tools/pmc_profile.sh over tools/kernel_bench.py"}
    json.dump(out, open(dst + ".json", "w"), indent=1, sort_keys=True)
    del out["_meta"]
    keys = sorted({kk for e in out.values() for kk in e})
    with open(dst + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel"] + keys)
        for k, e in sorted(out.items()):
            w.writerow([k] + [e.get(kk, "") for kk in keys])
    for k, e in sorted(out.items()):
        print(k[:70], {kk: (round(v, 4) if isinstance(v, float) and v < 10 else int(v)) for kk, v in e.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
