"""The parent of `python3 bench.py --gpus N`: N rank processes of the same script on this host, rank 0's JSON line relayed."""
import os
import subprocess
import sys
import tempfile
import time


def launch(args, script):
    """Spawn N rank processes of `script` (fresh children: nothing here has touched the GPU), relay rank 0's JSON line.  Every rank
    is told how many share the host (LOCAL_WORLD_SIZE): the Qhull helper budget is divided by it (same_amd/qhull_pool.py)."""
    n = args.gpus
    rdv = tempfile.mkdtemp(prefix="same_bench_rdv_")
    limit = float(os.environ.get("SAME_BENCH_LAUNCH_TIMEOUT", "1500"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), SAME_RDV_DIR=rdv,
                   MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))   # if RCCL has something to complain about, keep it on stderr
        out = subprocess.PIPE if r == 0 else sys.stderr   # only rank 0 writes the line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + sys.argv[1:], env=env, stdout=out))
    deadline = time.monotonic() + limit
    rc, line = 0, None
    try:
        import selectors

        sel = selectors.DefaultSelector()
        sel.register(procs[0].stdout, selectors.EVENT_READ)
        buf, open_out = b"", True
        while True:
            if open_out:
                for _key, _ in sel.select(timeout=0.2):
                    chunk = os.read(procs[0].stdout.fileno(), 65536)
                    if chunk:
                        buf += chunk
                    else:
                        open_out = False
                        sel.unregister(procs[0].stdout)
            else:
                time.sleep(0.1)
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0] if bad[0] > 0 else 1
                print(f"[bench launcher] a rank exited with {bad[0]}; stopping the others", file=sys.stderr)
                break
            if all(c == 0 for c in codes) and not open_out:
                break
            if time.monotonic() > deadline:
                rc = 124
                print(f"[bench launcher] ranks still running after {limit:.0f} s; stopping them", file=sys.stderr)
                break
        for ln in buf.decode(errors="replace").splitlines():
            if ln.startswith("{") and ln.rstrip().endswith("}"):
                line = ln
    finally:
        for p in procs:      # exactly the PIDs started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        try:
            for f in os.listdir(rdv):
                os.remove(os.path.join(rdv, f))
            os.rmdir(rdv)
        except OSError:
            pass
    if rc == 0 and line is None:
        print("[bench launcher] rank 0 finished without a JSON line", file=sys.stderr)
        rc = 1
    if line is not None and rc == 0:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    return rc
