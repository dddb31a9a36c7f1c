"""What clock / power sensors this box offers, and what the calibration kernels read on it (bench_telemetry.py).
Usage: python tools/telemetry_probe.py   (GPU box)"""
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
import bench_telemetry as BT                     # noqa: E402


def main():
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        print(dev, "->", os.path.realpath(dev), "hwmon:", [sorted(os.listdir(h))[:40] for h in hw][:1])
    for mk, nm in ((lambda: BT._Sysfs(None), "sysfs"), (lambda: BT._Rsmi(0), "rsmi"), (lambda: BT._AmdSmi(0), "amdsmi")):
        try:
            b = mk()
            t0 = time.perf_counter()
            for _ in range(20):
                r = b.read()
            print(nm, "ok:", r, b.describe(), f"{(time.perf_counter() - t0) / 20 * 1e6:.0f} us per read")
        except Exception as exc:
            print(nm, "unavailable:", type(exc).__name__, str(exc)[:200])
    dev = torch.device("cuda:0")
    s = BT.Sampler(0).start()
    cal = BT.Calibration(dev)
    with s.window("idle"):
        time.sleep(0.3)
    with s.window("calib_mfma_1s"):
        m = cal.mfma(1.0)
    with s.window("calib_hbm"):
        hb = cal.hbm()
    s.stop()
    print(json.dumps(dict(mfma=m, hbm=hb, telemetry=s.summary())))


if __name__ == "__main__":
    main()
