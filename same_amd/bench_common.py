"""What bench.py and its cfg 5 runner share: the line's metric string, progress notes, the rank watchdog, the communicator (RCCL, or
the host transport when RCCL cannot be created) and its self-description.  No torch anywhere (tests/test_dist_cpu.py checks)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

_LAST_STAGE = ["start"]


def note(group, msg):
    """Progress on stderr (rank 0): a cold box can spend minutes in imports / RCCL bootstrap, and stdout is reserved for the line."""
    _LAST_STAGE[0] = msg
    if group.rank == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def arm_rank_watchdog(rank):
    """A collective that never completes (a rank lost mid-run, a link that stops moving) has no timeout of its own: after
    SAME_BENCH_RANK_TIMEOUT seconds (default 900) the rank says where it was and leaves with code 4, so that the launcher (ours or
    torch.distributed.run) ends the job at once instead of at its own limit, with the GPUs still spinning."""
    import threading

    limit = float(os.environ.get("SAME_BENCH_RANK_TIMEOUT", "900"))

    def fire():
        print(f"[rank {rank}] still running after {limit:.0f} s; last stage: {_LAST_STAGE[0]!r}; giving up", file=sys.stderr, flush=True)
        os._exit(4)

    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    return t


def baseline_metric():
    """The metric string of BASELINE.json, verbatim (the file ships with the repo)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8"))["metric"]
    except Exception:
        return "cell-pairs/sec on 100k×100k cost build + edge-cross sweep; % HBM roofline"


def stats3(values):
    """[min, mean, max] of a list of numbers."""
    v = [float(x) for x in values]
    return [min(v), sum(v) / len(v), max(v)] if v else None


class Env:
    """What every problem of this rank shares: the host group, the two contexts, the communicator."""

    def __init__(self, args, group, ctx, tctx, comm, transport):
        self.args, self.group, self.ctx, self.tctx, self.comm, self.transport = args, group, ctx, tctx, comm, transport
        self.L, self.H, self.TH, self.chk = ctx.lib, ctx.handle, tctx.handle, ctx.check


def make_comm(args, group, tctx, what="pruned lists"):
    """The communicator, BEFORE any spread allocation (spread.hip never reuses an address for a mapping, but it cannot speak
    for RCCL's own use of the virtual-memory calls).  -> (comm or None, transport text)."""
    from same_amd.dist import HostTransport, RcclGroup

    # the env switch exercises the RCCL branch on one GPU (size-1 communicator)
    if not (group.world > 1 or os.environ.get("SAME_BENCH_FORCE_COMM")):
        return None, "none (single rank)"
    comm = None
    try:
        if os.environ.get("SAME_BENCH_FAIL_RCCL"):
            raise RuntimeError("forced by SAME_BENCH_FAIL_RCCL (test switch)")
        # ncclCommInitRank is a collective without a timeout: if it never returns (a rank lost, a bootstrap interface that
        # does not route) say so and leave, so the launcher stops the job at once instead of at its own limit
        import threading

        limit_s = float(os.environ.get("SAME_BENCH_RCCL_TIMEOUT", "300"))

        def stuck():
            print(f"[rank {group.rank}] RCCL communicator init has not returned after {limit_s:.0f} s; giving up", file=sys.stderr,
                  flush=True)
            os._exit(3)

        watchdog = threading.Timer(limit_s, stuck)
        watchdog.daemon = True
        watchdog.start()
        try:
            comm = RcclGroup(tctx, group.world, group.rank, lambda b: group.bcast_bytes(b or b""))
        finally:
            watchdog.cancel()
        ok_here = 1.0
    except Exception as e:  # TRANSPORT fallback only (compute stays on the GPU): reported in the JSON line
        print(f"[rank {group.rank}] RCCL communicator init failed ({e}); gathering through the host group instead", file=sys.stderr)
        ok_here = 0.0
    if group.min(ok_here) < 1.0:  # any rank failed -> every rank uses the host transport
        if comm is not None:
            comm.close()
        return HostTransport(tctx, group), f"HOST (loopback TCP) all-gather of {what}: RCCL init failed on this node"
    v = comm.rccl_version()
    return comm, f"RCCL {v // 10000}.{v // 100 % 100}.{v % 100} all-gather of {what}"


def comm_report(env, np):
    """`rccl`: what every rank's communicator says about itself -- the size and rank from ncclCommCount / ncclCommUserRank."""
    me = dict(env.comm.info(), host_rank=env.group.rank, local_rank=int(os.environ.get("LOCAL_RANK", str(env.group.rank))),
              hip_device=env.tctx.device, kind="rccl" if not env.comm.synchronous else "host")
    every = env.group.allgather_object(me)
    if env.group.rank != 0:
        return None
    v = me["version"]
    return {"kind": me["kind"], "nranks": me["nranks"], "rank": me["rank"], "device": me["device"],
            "version": f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v else None,
            "source": "ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the live communicator" if me["kind"] == "rccl"
                      else "host transport (no RCCL communicator): the host group's world and rank",
            "every_rank": [[r["host_rank"], r["rank"], r["nranks"], r["device"]] for r in every],
            "every_rank_columns": ["host rank", "communicator rank", "communicator size", "device"],
            "consistent": all(r["nranks"] == env.group.world and r["rank"] == r["host_rank"] for r in every),
            "distinct_devices": len({r["device"] for r in every})}
