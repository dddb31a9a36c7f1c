"""What the BOX was doing while bench.py measured: engine clock, socket power and temperature sampled from
a side thread, and two fixed calibration kernels of the library (csrc/calib.hip) timed in-process.  The renderer's kernels
change from round to round; the calibration kernels do not: they say which STATE a lease found the chip in (round 6's survey,
profiles/r06_box_survey.jsonl: the same GPU read 1 788 and 1 980 TF on two leases; the training step follows the pure-MFMA
figure with an elasticity of ~0.37, so the ratio value / calib is an indicator, not a box-independent constant).

Sensor back ends, first that answers: (1) the amdgpu hwmon files under /sys/class/drm/card*/device/hwmon (plain file reads, a
few microseconds each), (2) librocm_smi64 through ctypes, (3) the amdsmi Python package.  None of them needs root.  A box that
offers none yields `{"backend": None}`; the calibration kernels still run."""
import ctypes
import glob
import os
import threading
import time

import torch


# ------------------------------------------------------------------ sensors
class _Sysfs:
    """amdgpu hwmon: freq1_input (Hz, sclk), power1_average / power1_input (uW), temp*_input (millidegrees)"""
    name = "sysfs-hwmon"

    def __init__(self, bdf=None):
        cands = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            real = os.path.realpath(dev)
            try:
                if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
            if hw:
                cands.append((os.path.basename(real), hw[0], dev))
        if not cands:
            raise RuntimeError("no amdgpu hwmon directory")
        pick = [c for c in cands if bdf and c[0].lower().endswith(bdf.lower())] or cands[:1]
        self.bdf, self.hw, self.dev = pick[0]
        self.n_cards = len(cands)
        first = lambda names: next((os.path.join(self.hw, n) for n in names if os.path.exists(os.path.join(self.hw, n))), None)
        self.f_clk = first(["freq1_input"])
        self.f_pow = first(["power1_input", "power1_average"])
        self.f_tmp = first(["temp2_input", "temp1_input"])          # junction if present, else edge
        self.f_cap = first(["power1_cap"])
        if self.f_clk is None and self.f_pow is None:
            raise RuntimeError("hwmon has neither freq1_input nor power1_*")
        self.read()                                                  # permission / EIO errors surface here

    @staticmethod
    def _num(path, scale):
        if path is None:
            return None
        with open(path) as f:
            return float(f.read().strip()) * scale

    def read(self):
        return self._num(self.f_clk, 1e-9), self._num(self.f_pow, 1e-6), self._num(self.f_tmp, 1e-3)

    def describe(self):
        cap = None
        try:
            cap = self._num(self.f_cap, 1e-6)
        except OSError:
            pass
        return dict(device=self.bdf, cards_seen=self.n_cards, power_cap_w=cap,
                    files=[os.path.basename(p) for p in (self.f_clk, self.f_pow, self.f_tmp) if p])


class _Rsmi:
    """librocm_smi64: rsmi_dev_gpu_clk_freq_get (current level of the system clock), rsmi_dev_power_get, rsmi_dev_temp_metric_get"""
    name = "librocm_smi64"

    class _Freqs(ctypes.Structure):
        _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                    ("frequency", ctypes.c_uint64 * 33)]

    def __init__(self, index=0):
        path = next((p for p in ("/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so") if p.startswith("lib") or os.path.exists(p)), None)
        self.lib = ctypes.CDLL(path)
        if self.lib.rsmi_init(ctypes.c_uint64(0)) != 0:
            raise RuntimeError("rsmi_init failed")
        n = ctypes.c_uint32(0)
        self.lib.rsmi_num_monitor_devices(ctypes.byref(n))
        if n.value == 0:
            raise RuntimeError("no rsmi devices")
        self.idx = min(index, n.value - 1)
        self.n = n.value
        if all(v is None for v in self.read()):
            raise RuntimeError("rsmi answers nothing")

    def read(self):
        f = self._Freqs()
        clk = None
        if self.lib.rsmi_dev_gpu_clk_freq_get(self.idx, 0, ctypes.byref(f)) == 0 and f.current < 33:
            clk = f.frequency[f.current] * 1e-9
        p, typ, pw = ctypes.c_uint64(0), ctypes.c_int(0), None
        if self.lib.rsmi_dev_power_get(self.idx, ctypes.byref(p), ctypes.byref(typ)) == 0:
            pw = p.value * 1e-6
        t, tj = ctypes.c_int64(0), None
        for sensor in (1, 0):                                        # junction, edge
            if self.lib.rsmi_dev_temp_metric_get(self.idx, sensor, 0, ctypes.byref(t)) == 0:
                tj = t.value * 1e-3
                break
        return clk, pw, tj

    def describe(self):
        return dict(device=self.idx, cards_seen=self.n)


class _AmdSmi:
    name = "amdsmi"

    def __init__(self, index=0):
        import amdsmi
        self.m = amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        if not hs:
            raise RuntimeError("no amdsmi devices")
        self.h = hs[min(index, len(hs) - 1)]
        self.n = len(hs)
        if all(v is None for v in self.read()):
            raise RuntimeError("amdsmi answers nothing")

    def read(self):
        clk = pw = tj = None
        try:
            c = self.m.amdsmi_get_clock_info(self.h, self.m.AmdSmiClkType.GFX)
            clk = float(c.get("clk", c.get("cur_clk"))) * 1e-3
        except Exception:
            pass
        try:
            p = self.m.amdsmi_get_power_info(self.h)
            v = p.get("current_socket_power", p.get("average_socket_power"))
            pw = float(v) if v not in (None, "N/A") else None
        except Exception:
            pass
        try:
            tj = float(self.m.amdsmi_get_temp_metric(self.h, self.m.AmdSmiTemperatureType.HOTSPOT, self.m.AmdSmiTemperatureMetric.CURRENT))
        except Exception:
            pass
        return clk, pw, tj

    def describe(self):
        return dict(device=0, cards_seen=self.n)


def open_sensors(device_index=0):
    """-> (backend object or None, {backend name: why it was not used})"""
    bdf = None
    try:
        pr = torch.cuda.get_device_properties(device_index)
        if hasattr(pr, "pci_bus_id"):
            bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        pass
    why = {}
    for make in (lambda: _Sysfs(bdf), lambda: _Rsmi(device_index), lambda: _AmdSmi(device_index)):
        try:
            return make(), why
        except Exception as exc:                                     # a back end that is absent or unreadable on this box
            why[["sysfs-hwmon", "librocm_smi64", "amdsmi"][len(why)]] = f"{type(exc).__name__}: {str(exc)[:120]}"
    return None, why


class Sampler:
    """Samples (time, clock GHz, power W, temperature C) at `hz` in a daemon thread; `window(name)` contexts tag the samples taken
    while a region ran (entered / left on the host: the GPU work of a region is bracketed by synchronize() in bench.py)."""

    def __init__(self, device_index=0, hz=50.0):
        self.backend, self.unavailable = open_sensors(device_index)
        self.period = 1.0 / hz
        self.samples, self.marks = [], []
        self._stop = threading.Event()
        self._thread = None

    def start(self):
        if self.backend is not None and self._thread is None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def _run(self):
        nxt = time.perf_counter()
        while not self._stop.is_set():
            try:
                self.samples.append((time.perf_counter(),) + tuple(self.backend.read()))
            except Exception:
                pass
            nxt += self.period
            d = nxt - time.perf_counter()
            if d > 0:
                self._stop.wait(d)
            else:
                nxt = time.perf_counter()

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)

    class _Window:
        def __init__(self, s, name):
            self.s, self.name = s, name

        def __enter__(self):
            self.t0 = time.perf_counter()
            return self

        def __exit__(self, *exc):
            self.s.marks.append((self.name, self.t0, time.perf_counter()))
            return False

    def window(self, name):
        return Sampler._Window(self, name)

    @staticmethod
    def _stats(vals):
        vals = [v for v in vals if v is not None]
        if not vals:
            return None
        return dict(mean=round(sum(vals) / len(vals), 4), min=round(min(vals), 4), max=round(max(vals), 4))

    def summary(self):
        out = dict(backend=self.backend.name if self.backend else None, hz=round(1.0 / self.period, 1), samples=len(self.samples))
        if self.backend is None:
            out["unavailable"] = self.unavailable
            return out
        out.update(self.backend.describe())
        if self.unavailable:
            out["skipped_backends"] = self.unavailable
        regions = {}
        for name, t0, t1 in self.marks:
            sel = [s for s in self.samples if t0 <= s[0] <= t1]
            regions[name] = dict(seconds=round(t1 - t0, 4), n=len(sel), clock_ghz=self._stats([s[1] for s in sel]),
                                 power_w=self._stats([s[2] for s in sel]), temp_c=self._stats([s[3] for s in sel]))
        out["regions"] = regions
        return out


# ------------------------------------------------------------------ calibration kernels
class Calibration:
    """The library's two calibration kernels on the current stream, timed with stream events.  `mfma(seconds)` runs launches of a
    fixed size back to back for about `seconds` (the socket needs ~0.1 s to settle on its power-capped clock) and reports the rate
    of the whole run and of its second half; `hbm()` streams a 2 GiB buffer (8x the 256 MB last-level cache)."""
    MFMA_ITERS = 1 << 14          # 16 Ki x 16 MFMAs per wave and launch: ~9 ms at 1.2 PF
    HBM_BYTES = 1 << 31

    def __init__(self, device):
        from sparf_amd import lib as L
        self.L, self.lib, self.device = L, L.load(), device
        self.sink = torch.zeros(L.CALIB_SINK_FLOATS, device=device)
        self.buf = None

    def _events(self, n):
        return [torch.cuda.Event(enable_timing=True) for _ in range(n)]

    def mfma(self, seconds=0.25):
        L, s = self.L, self.L.stream_ptr(self.device)
        launch = lambda: self.lib.sparf_calib_mfma(self.MFMA_ITERS, L.ptr(self.sink), s)
        flops = launch()
        if flops <= 0:
            raise L.SparfError(f"sparf_calib_mfma failed with {flops}")
        e = self._events(2)
        e[0].record(); launch(); e[1].record()
        torch.cuda.synchronize()
        n = max(4, int(seconds / (e[0].elapsed_time(e[1]) * 1e-3)))
        ev = self._events(n + 1)
        ev[0].record()
        for i in range(n):
            launch()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        tf = lambda sel: flops * len(sel) / (sum(sel) * 1e-3) / 1e12
        return dict(tflops=tf(ms), tflops_second_half=tf(ms[n // 2:]), tflops_first_launch=tf(ms[:1]), launches=n, seconds=sum(ms) * 1e-3,
                    flops_per_launch=flops, what="v_mfma_f32_32x32x16_bf16 back to back, random operand bits, 2 waves per SIMD on every CU (issued bf16 TFLOP/s)")

    def hbm(self, reps=6):
        L, s = self.L, self.L.stream_ptr(self.device)
        if self.buf is None:
            self.buf = torch.empty(2, self.HBM_BYTES, dtype=torch.uint8, device=self.device)
            self.buf.random_(0, 256)
        out = {}
        for mode, name in ((0, "read_lds_dma"), (1, "copy")):
            launch = lambda: L.check(self.lib.sparf_calib_hbm(L.ptr(self.buf[0]), L.ptr(self.buf[1]), self.HBM_BYTES, mode, L.ptr(self.sink), s), "sparf_calib_hbm")
            launch()
            e = self._events(2)
            e[0].record()
            for _ in range(reps):
                launch()
            e[1].record()
            torch.cuda.synchronize()
            moved = self.HBM_BYTES * (2 if mode == 1 else 1)
            out[name] = moved * reps / (e[0].elapsed_time(e[1]) * 1e-3) / 1e12
        return dict(read_lds_dma_tbs=out["read_lds_dma"], copy_tbs=out["copy"], bytes=self.HBM_BYTES,
                    what="2 GiB: read once through LDS-DMA (global_load_lds_dwordx4 nt) / copied (read + written bytes counted)")

    def mix(self, seconds=0.3):
        """The two calibration kernels ALTERNATING at the training step's own cadence -- ~2.3 ms of the MFMA stream, then ~1.8 ms of the
        HBM read stream (five 2 GiB reads), as the step alternates its matrix-bound forward / data-gradient kernels with its HBM-bound
        weight-gradient kernel -- for about `seconds`: cycles per second.  Under the socket power cap the clock a chip holds is an
        average over milliseconds, so what a chip sustains on the pure MFMA stream over-states how much a box that clocks lower
        loses on the mixed step (round 6: a box 6 % slower on `mfma` ran the step 1.7 % slower)."""
        L, s = self.L, self.L.stream_ptr(self.device)
        if self.buf is None:
            self.hbm(reps=1)
        iters = self.MFMA_ITERS // 4

        def cycle():
            self.lib.sparf_calib_mfma(iters, L.ptr(self.sink), s)
            for _ in range(5):
                self.lib.sparf_calib_hbm(L.ptr(self.buf[0]), L.ptr(self.buf[1]), self.HBM_BYTES, 0, L.ptr(self.sink), s)
        cycle()
        e = self._events(2)
        e[0].record(); cycle(); e[1].record()
        torch.cuda.synchronize()
        n = max(8, int(seconds / (e[0].elapsed_time(e[1]) * 1e-3)))
        ev = self._events(n + 1)
        ev[0].record()
        for i in range(n):
            cycle()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        half = ms[n // 2:]
        return dict(cycles_per_s=len(ms) / (sum(ms) * 1e-3), cycles_per_s_second_half=len(half) / (sum(half) * 1e-3), cycles=n, ms_per_cycle=sum(ms) / n,
                    what="one cycle = sparf_calib_mfma (4 Ki x 16 MFMAs per wave, ~2.3 ms) + 5 x sparf_calib_hbm read of 2 GiB (~1.8 ms)")

    def run(self, seconds=0.25):
        return dict(mfma=self.mfma(seconds), hbm=self.hbm(), mix=self.mix())

    def release(self):
        self.buf = None
        torch.cuda.empty_cache()
