"""Driver of tools/probes/mfma_sub (the FP64-MFMA subtraction probe): runs the binary once per configuration, samples the sysfs
telemetry of same_amd/telemetry.py (held clock, watts) BETWEEN the binary's TIMED_START / TIMED_END marks, prints one JSON line per
configuration and whether the outputs are bit-identical across the configurations of one data set.
Usage: python3 tools/probes/mfma_sub.py [n=100000] [seconds=4]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from same_amd import _lib                      # noqa: E402
from same_amd.telemetry import GpuTelemetry    # noqa: E402

n = sys.argv[1] if len(sys.argv) > 1 else "100000"
seconds = sys.argv[2] if len(sys.argv) > 2 else "4"
exe = os.path.join(ROOT, "tools", "probes", "mfma_sub")
ctx = _lib.Context(0)
pci = ctx.pci_bus_id()
ctx.close()
# (mfma units, period, LDS pad of the VALU kernel, LDS pad of the MFMA kernel, data set)
CONFIGS = ((0, 1, 0, 0, 0), (0, 1, 0, 0, 1),                                       # the VALU form alone on both data sets
           (1, 8, 26624, 32768, 1), (1, 4, 26624, 32768, 1), (2, 5, 26624, 32768, 1),   # one MFMA block + three VALU blocks per CU
           (1, 4, 26624, 0, 1), (1, 4, 0, 0, 1), (1, 1, 0, 0, 1))                      # up to three MFMA blocks per CU; no caps; MFMA alone
rows = []
for n_mfma, period, pad_v, pad_m, data in CONFIGS:
    tel = GpuTelemetry(pci)
    p = subprocess.Popen([exe, str(n_mfma), str(period), n, seconds, str(pad_v), str(pad_m), str(data)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t, rec = None, None
    for line in p.stdout:
        line = line.strip()
        if line == "TIMED_START":
            tel.start()
        elif line == "TIMED_END":
            t = tel.stop()
        elif line.startswith("{"):
            rec = json.loads(line)
    p.wait(timeout=600)
    if rec is None:
        print(json.dumps({"config": [n_mfma, period, pad_v, pad_m, data], "error": p.stderr.read()[-500:], "rc": p.returncode}), flush=True)
        continue
    if t:
        rec["sclk_mhz"] = (t.get("sclk_steady") or {}).get("mean")
        rec["power_w"] = (t.get("power_steady") or {}).get("mean")
        rec["power_cap_w"] = t.get("power_cap_w")
    rows.append(rec)
    print(json.dumps(rec), flush=True)
for data in (0, 1):
    mine = [r for r in rows if r["data"] == data]
    if not mine:
        continue
    base = mine[0]
    same = all(r["checksum_xor"] == base["checksum_xor"] and r["checksum_sum"] == base["checksum_sum"] for r in mine)
    best = min(mine, key=lambda r: r["ms"])
    print(json.dumps({"data": data, "configurations": len(mine), "bit_identical_across_configurations": same,
                      "host_mismatches_total": sum(r["host_mismatches"] for r in mine), "valu_only_ms": base["ms"], "best_ms": best["ms"],
                      "best_fraction": best["mfma_fraction"], "gain": 1.0 - best["ms"] / base["ms"]}), flush=True)
