"""cProfile of one `bench.py --workload cfg5` rank (host glue is ~90 % of that step): top functions by cumulative and own time.
Usage: python tools/cfg5_profile.py [cells]"""
import cProfile, io, os, pstats, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.argv = ["bench.py", "--workload", "cfg5", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--cfg5-threads", "1", "--cfg5-pipeline", os.environ.get("CFG5_PIPELINE", "device"), "--cfg5-cells", sys.argv[1] if len(sys.argv) > 1 else "1000000"]
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(60)
        print(s.getvalue()[:14000], file=sys.stderr)
