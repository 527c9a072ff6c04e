"""Board power and shader clock of one GPU, read from sysfs by a side thread (no subprocess, no SMI library).

Used by bench.py to record the operating point the dense kernel actually ran at (the T=20 fp64 build is bounded by
the package power cap): `power1_average` / `power1_input` (microwatts) and `freq1_input` (Hz) of the
amdgpu hwmon node, `pp_dpm_sclk` (the line marked `*`) as a second clock source, `power1_cap` for the cap itself, and the
`junction` / `mem` temperatures (a chip that throttles below the power cap is usually at a thermal limit).
Every source is optional; what could not be read is reported as such, never guessed.
"""
import glob
import os
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _dpm_current_mhz(text):
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            try:
                return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            except (IndexError, ValueError):
                return None
    return None


class GpuTelemetry:
    def __init__(self, pci_bus_id, period_s=0.02, sysfs_root="/sys/bus/pci/devices"):
        self.period = float(period_s)
        self.dev_dir = os.path.join(sysfs_root, pci_bus_id)
        hw = sorted(glob.glob(os.path.join(self.dev_dir, "hwmon", "hwmon*")))
        self.hwmon = hw[0] if hw else None
        self.power_path = None
        if self.hwmon:
            for name in ("power1_average", "power1_input"):
                if _read(os.path.join(self.hwmon, name)) is not None:
                    self.power_path = os.path.join(self.hwmon, name)
                    break
        self.freq_path = os.path.join(self.hwmon, "freq1_input") if self.hwmon and _read(os.path.join(self.hwmon, "freq1_input")) else None
        self.dpm_path = os.path.join(self.dev_dir, "pp_dpm_sclk") if _read(os.path.join(self.dev_dir, "pp_dpm_sclk")) else None
        # junction (hot spot) and HBM temperatures, millidegrees; found by label because the numbering differs between parts
        self.temp_paths = {}
        if self.hwmon:
            for lab in sorted(glob.glob(os.path.join(self.hwmon, "temp*_label"))):
                name = _read(lab)
                if name in ("junction", "mem", "edge") and _read(lab.replace("_label", "_input")) is not None:
                    self.temp_paths[name] = lab.replace("_label", "_input")
            self.temp_crit_c = {n: (float(v) / 1e3 if (v := _read(p.replace("_input", "_crit"))) and v.lstrip("-").isdigit() else None)
                                for n, p in self.temp_paths.items()}
        else:
            self.temp_crit_c = {}
        cap = _read(os.path.join(self.hwmon, "power1_cap")) if self.hwmon else None
        self.cap_w = float(cap) / 1e6 if cap and cap.isdigit() else None
        self._samples = []
        self._stop = threading.Event()
        self._thread = None

    def available(self):
        return bool(self.power_path or self.freq_path or self.dpm_path)

    def sample(self):
        p = _read(self.power_path) if self.power_path else None
        f = _read(self.freq_path) if self.freq_path else None
        d = _dpm_current_mhz(_read(self.dpm_path)) if self.dpm_path else None
        temps = {n: (float(v) / 1e3 if (v := _read(path)) and v.lstrip("-").isdigit() else None) for n, path in self.temp_paths.items()}
        return (time.perf_counter(), float(p) / 1e6 if p and p.isdigit() else None,
                float(f) / 1e6 if f and f.isdigit() else None, d, temps)

    def start(self):
        self._samples, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                self._samples.append(self.sample())
                self._stop.wait(self.period)

        self._thread = threading.Thread(target=loop, name="gpu-telemetry", daemon=True)
        self._thread.start()

    def stop(self):
        """-> summary dict of the samples taken since start()."""
        self._stop.set()
        if self._thread:
            self._thread.join()
        s = self._samples

        def stats(vals, unit):
            vals = [v for v in vals if v is not None]
            if not vals:
                return None
            return {"mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals), "unit": unit}

        # the first quarter of the window is the ramp from idle (clocks up, power1_input is itself an average): steady = the rest
        steady = s[len(s) // 4:] if len(s) >= 8 else s
        return {"available": self.available(), "samples": len(s),
                "window_s": (s[-1][0] - s[0][0]) if len(s) > 1 else 0.0,
                "power": stats([x[1] for x in s], "W"), "power_steady": stats([x[1] for x in steady], "W"),
                "sclk_steady": stats([x[2] if x[2] is not None else x[3] for x in steady], "MHz"), "power_cap_w": self.cap_w,
                "sclk_hwmon": stats([x[2] for x in s], "MHz"), "sclk_dpm": stats([x[3] for x in s], "MHz"),
                "temperature_steady": {n: stats([x[4].get(n) for x in steady], "C") for n in self.temp_paths} or None,
                "temperature_crit_c": self.temp_crit_c or None,
                "source": {"power": self.power_path, "sclk_hwmon": self.freq_path, "sclk_dpm": self.dpm_path}}
