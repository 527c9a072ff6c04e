"""Dev probe: the rates the labelling of same_dev_alloc_spread reads (SAME_SPREAD_DEBUG=1), as a histogram: how far apart
are the same-region and the other-region level?  Usage: python tools/probes/spread_levels.py [GiB]"""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 75
code = f"import sys; sys.path.insert(0, {root!r}); from same_amd import _lib; c = _lib.Context(0); b = c.alloc_spread({gib} << 30); print(b.spread_info)"
p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SAME_SPREAD_DEBUG="1"), capture_output=True, text=True)
print(p.stdout.strip())
hist = {"pair": collections.Counter(), "halves": collections.Counter()}
again = 0
for m in re.finditer(r"\[spread\] (pair|halves) rate (\d+)( \(again: (\d+)\))? vs level (\d+) -> (fast|slow)", p.stderr):
    hist[m.group(1)][(int(m.group(4) or m.group(2)) // 100 * 100, m.group(6))] += 1
    again += bool(m.group(3))
for kind, h in hist.items():
    print(kind, "(GB/s bin of 100, verdict): count")
    for k in sorted(h):
        print(f"  {k[0]:5d} {k[1]}: {h[k]}")
print("readings taken twice:", again)
if p.returncode:
    print(p.stderr[-2000:]); sys.exit(p.returncode)
