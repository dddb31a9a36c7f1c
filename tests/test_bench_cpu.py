"""Host-side pieces of bench.py that need no GPU: which committed profile a line quotes (newest ROUND, not the
lexicographically largest name -- r100 > r99, r03b > r03), the stale-kernel flag, the PMC traffic lookup."""
import json
import os

import bench


def test_measured_parity_picks_the_newest_round_and_flags_stale_kernels(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    body = lambda tag, stamp: dict(_meta=dict(kernel_source_hash=stamp), summary={"bf16x3": {"metric_depth": {"tag": tag}}})
    from sparf_amd.build import source_hash
    for name, tag, stamp in (("r99_parity_scale.json", "r99", "0" * 16), ("r100_parity_scale.json", "r100", source_hash()),
                             ("r03_parity_scale.json", "r03", None), ("r03b_parity_scale.json", "r03b", "0" * 16)):
        d = body(tag, stamp)
        if stamp is None:
            d["_meta"].pop("kernel_source_hash")
        (prof / name).write_text(json.dumps(d))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.measured_parity("bf16x3")
    assert got["metric_depth"]["tag"] == "r100" and got["stale"] is False and got["source"] == os.path.join("profiles", "r100_parity_scale.json")
    (prof / "r100_parity_scale.json").unlink()
    got = bench.measured_parity("bf16x3")
    assert got["metric_depth"]["tag"] == "r99" and got["stale"] is True          # measured on other kernel sources
    (prof / "r99_parity_scale.json").unlink()
    assert bench.measured_parity("bf16x3")["metric_depth"]["tag"] == "r03b"        # suffix sessions of a round sort after the plain tag
    (prof / "r03b_parity_scale.json").unlink()
    got = bench.measured_parity("bf16x3")
    assert got["metric_depth"]["tag"] == "r03" and got["stale"] is None           # unstamped profile: unknown
    assert bench.measured_parity("fp32") is None


def test_pmc_traffic_reads_the_committed_profile():
    total, src, util = bench.pmc_traffic("mlp_fwd", "bf16x3", 786432)
    assert src is not None and src.startswith("profiles/") and 3.5e9 < total < 4.5e9
    # matrix-pipe busy: ~62 % of the cycles at the ~1.85-1.9 GHz the chip held = ~48 % of the issue slots at the 2.4 GHz peak clock
    assert 0.3 < util["at_clock_held"] < 0.8 and util["at_peak_clock"] < util["at_clock_held"] and 1.5 < util["clock_held_ghz"] < 2.45
    assert bench.pmc_traffic("mlp_fwd", "bf16x3", 12345) == (None, None, None)      # no profile at that row count


def test_telemetry_sampler_and_box_summary_without_a_gpu():
    """bench_telemetry.Sampler with a stand-in sensor back end: samples are attributed to the window they fall into, and bench.box_summary
    turns them + the calibration figures into the `config.box` / `roofline.box` entry"""
    import time
    import bench_telemetry as BT

    class Fake:
        name = "fake"

        def __init__(self):
            self.n = 0

        def read(self):
            self.n += 1
            return 1.9 + 0.001 * (self.n % 3), 1340.0, 52.0

        def describe(self):
            return dict(device="0000:00:00.0", cards_seen=1, power_cap_w=1400.0)

    s = BT.Sampler.__new__(BT.Sampler)
    s.backend, s.unavailable, s.period, s.samples, s.marks = Fake(), {}, 1.0 / 200.0, [], []
    import threading
    s._stop, s._thread = threading.Event(), None
    s.start()
    with s.window("contract"):
        time.sleep(0.1)
    time.sleep(0.03)
    with s.window("sustained"):
        time.sleep(0.1)
    s.stop()
    tel = s.summary()
    assert tel["backend"] == "fake" and tel["regions"]["contract"]["n"] >= 5 and tel["regions"]["sustained"]["n"] >= 5
    assert abs(tel["regions"]["contract"]["clock_ghz"]["mean"] - 1.901) < 2e-3 and tel["regions"]["contract"]["power_w"]["mean"] == 1340.0
    assert tel["regions"]["contract"]["n"] + tel["regions"]["sustained"]["n"] < len(s.samples)          # the samples between the windows belong to neither
    cal = dict(mfma=dict(tflops_second_half=1950.0), hbm=dict(read_lds_dma_tbs=5.9, copy_tbs=4.9), mix=dict(cycles_per_s_second_half=250.0))
    box = bench.box_summary(tel, dict(calib_before=cal, calib_after=dict(cal, mfma=dict(tflops_second_half=1930.0))), 610000.0, [6.7, 6.6],
                            dict(value=612000.0))
    assert box["calib_mfma_tflops"] == 1940.0 and box["calib_mfma_tflops_before_after"] == [1950.0, 1930.0]
    assert abs(box["value_per_calib_mfma_tflop"] - 610000.0 / 1940.0) < 1e-9 and box["contract_step_ms"] == [6.7, 6.6]
    assert box["clock_ghz_min"] >= 1.9 and box["power_w_mean"] == 1340.0 and box["sustained_value"] == 612000.0
    assert box["calib_mix_cycles_per_s"] == 250.0 and box["value_per_calib_mix_cycle"] == 610000.0 / 250.0
    # no sensors at all (a box that offers none): the summary says so and the calibration figures still stand
    s2 = BT.Sampler.__new__(BT.Sampler)
    s2.backend, s2.unavailable, s2.period, s2.samples, s2.marks = None, {"sysfs-hwmon": "x"}, 0.02, [], []
    box2 = bench.box_summary(s2.summary(), dict(calib_before=cal), 600000.0, [], None)
    assert box2["sensor_backend"] is None and box2["clock_ghz_mean"] is None and box2["calib_mfma_tflops"] == 1950.0


def test_live_pmc_traffic_parses_what_rocprofv3_writes(tmp_path, monkeypatch):
    """bench.live_pmc_traffic against a stand-in `rocprofv3` that writes counter-collection CSVs in rocprofv3's layout: the figures are the
    mean of the LAST five launches of the dominant kernel, FETCH_SIZE in KiB x 2 (gfx950), WRITE_SIZE in KiB; other kernels' rows are ignored;
    a failing pass or a missing tool yields (None, reason) and the bench line falls back to the committed profile"""
    import stat
    import shutil
    fake = tmp_path / "rocprofv3"
    fake.write_text('''#!/usr/bin/env python3
import sys, os, csv
a = sys.argv[1:]
out_dir, name = a[a.index("-d") + 1], a[a.index("-o") + 1]
counters = a[a.index("--pmc") + 1:a.index("--kernel-trace")]
if os.environ.get("FAKE_FAIL") == name:
    sys.exit(3)
os.makedirs(os.path.join(out_dir, "host", "123"), exist_ok=True)
kern = "void sparf::mlp_fwd_kernel<2, 1>(sparf::MlpFwdArgs)"
with open(os.path.join(out_dir, "host", "123", name + "_counter_collection.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
    for i in range(12):
        for c in counters:
            v = {"FETCH_SIZE": 1000.0 + (i >= 7) * 10.0, "WRITE_SIZE": 4000.0 + (i >= 7) * 40.0, "SQ_VALU_MFMA_BUSY_CYCLES": 2.0e9, "GRBM_GUI_ACTIVE": 8 * 4.0e6}[c]
            w.writerow([kern, c, v, 1000, 1000 + 2000000])
        w.writerow(["void sparf::wgrad_kernel<2, false>(sparf::WgradArgs)", counters[0], 9.9e9, 0, 1])
''')
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setattr(shutil, "which", lambda name: str(fake) if name == "rocprofv3" else None)
    total, how = bench.live_pmc_traffic("mlp_fwd", "bf16x3", 786432)
    assert total == 1010.0 * 2048 + 4040.0 * 1024 and how.startswith("measured by this run")
    ex = bench.live_pmc_traffic.extra
    assert abs(ex["pmc_mfma_busy"] - 2.0e9 / (1024 * 4.0e6)) < 1e-12 and abs(ex["clock_held_ghz_under_pmc"] - 2.0) < 1e-12 and ex["launch_ms_under_pmc"] == 2.0
    monkeypatch.setenv("FAKE_FAIL", "WRITE_SIZE")
    total, how = bench.live_pmc_traffic("mlp_fwd", "bf16x3", 786432)
    assert total is None and "WRITE_SIZE failed" in how
    monkeypatch.setenv("FAKE_FAIL", "SQ_VALU_MFMA_BUSY_CYCLES")                   # the optional third pass may fail: the traffic figures stand
    total, how = bench.live_pmc_traffic("mlp_fwd", "bf16x3", 786432)
    assert total is not None and bench.live_pmc_traffic.extra is None
    assert bench.live_pmc_traffic("mlp_fwd", "bf16x3", 12345)[0] is None          # kernel_bench measures the 786 432-row pass only
    monkeypatch.delenv("FAKE_FAIL")
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")        # bench.py itself under rocprofv3: no nested profiler
    total, how = bench.live_pmc_traffic("mlp_fwd", "bf16x3", 786432)
    assert total is None and "nested" in how
