#!/bin/bash
# round 4, visit D: the driver's own sequence -- GPU suite, smoke, the default bench line (with its in-job cfg5 record)
set -o pipefail
tag=${1:-r04d}
out=gpurun_out/$tag; mkdir -p $out
echo "== pytest -m gpu" && timeout -k 10 1100 python3 -m pytest tests -m gpu -q --durations=8 > $out/pytest_gpu.log 2>&1; rc=$?; tail -4 $out/pytest_gpu.log; [ $rc -eq 0 ] || { grep -n "Error\|FAILED" $out/pytest_gpu.log | head -20; exit $rc; }
echo "== smoke" && timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
echo "== bench (default)" && SECONDS=0; timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
echo "bench wall $SECONDS s"; grep "^\[bench" $out/bench.err | tail -12
python3 - $out/bench.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.3e  ms/step %.3f  frac %.4f  binding %r  frac_of_binding_ceiling %s target_met %s" % (d["value"], d["ms_per_step"], r["frac"], r["binding"], r["frac_of_binding_ceiling"], r["target_met"]))
c = d["cfg5"]
print("cfg5:", {k: c.get(k) for k in ("windows_per_s", "ms_per_step", "host_glue_share", "runtime_calls_per_window", "merged_matches", "error")})
print("cfg5 parity:", c.get("parity_spot_check"))
P
echo "== done"
