"""One line per cfg 5 bench file: windows/s (whole job and per rank), Python / Qhull-wait shares.  Usage: python3 tools/cfg5_lines.py a.json b.json ..."""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    d = d.get("cfg5", d) if "windows_per_s" not in d else d
    print(f, "windows/s", round(d["windows_per_s"], 1), "per rank", [round(v, 1) for v in d["per_rank"]["windows_per_s"]],
          "threads", d.get("threads_per_rank"), "python_share", round(d["python_share"], 2), "qhull_wait_share", round(d["qhull_wait_share"], 2))
    nd = d.get("native_delaunay")
    if nd:
        print("   native triangulator (opt-in): windows/s", round(nd["windows_per_s"], 1), "table identical", nd["table_identical_to_the_timed_step"],
              "sent back to Qhull", nd["sent_back_to_qhull"], "of", nd["windows_triangulated"], "threads", nd["threads_per_rank"])
