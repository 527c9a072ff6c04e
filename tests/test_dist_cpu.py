"""Multi-rank tests on CPU: the plain-Python rendezvous (world 2, 3, 8), the sharded cost build + prune, the
triangle-block sharded sweeps, and bench.py's own launcher.

There is no GPU here, so the per-rank compute is supplied by the CPU oracle (test infrastructure); what is under test is
the product's sharding logic: row / triangle blocks, padding, the gathers and reductions, and the compaction that must
make the result identical for any world size.  The same sharding functions are also driven through a
torch.distributed gloo group (world 2) via a small adapter that lives here in tests/ -- the product itself is torch-free."""
import json
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------- helpers
def _run_ranks(target, world, *args):
    """Start `world` processes of target(rank, world, rdv_dir, *args); fail if any of them does."""
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory(prefix="same_rdv_test_") as rdv:
        procs = [ctx.Process(target=target, args=(r, world, rdv) + args) for r in range(world)]
        [p.start() for p in procs]
        [p.join(300) for p in procs]
        for p in procs:
            if p.is_alive():
                p.terminate()
        assert [p.exitcode for p in procs] == [0] * world


def _setup_paths():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))


class GlooGroup:
    """HostGroup's interface on a torch.distributed gloo process group (tests only)."""

    def __init__(self, rank, world, port):
        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.dist, self.rank, self.world = dist, rank, world

    def allgather_array(self, arr):
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return np.concatenate([o.numpy() for o in outs], axis=0)

    def sum_int(self, values):
        import torch

        t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.int64).copy())
        self.dist.all_reduce(t)
        return t.numpy()

    def close(self):
        self.dist.destroy_process_group()


def _make_group(kind, rank, world, rdv):
    if kind == "gloo":
        port = int(open(os.path.join(rdv, "port")).read()) if rank else None
        if rank == 0:
            import socket

            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            tmp = os.path.join(rdv, "port.tmp")
            open(tmp, "w").write(str(port))
            os.replace(tmp, os.path.join(rdv, "port"))
        return GlooGroup(rank, world, port)
    from same_amd.rendezvous import HostGroup

    return HostGroup(rank, world, rdv_dir=rdv, timeout=120)


def _wait_port(rdv):
    import time

    for _ in range(3000):
        if os.path.exists(os.path.join(rdv, "port")):
            return
        time.sleep(0.01)


# ---------------------------------------------------------------------------------------------- rendezvous
def _rdv_worker(rank, world, rdv, out_dir):
    _setup_paths()
    from same_amd.rendezvous import HostGroup

    with HostGroup(rank, world, rdv_dir=rdv, timeout=120) as g:
        uid = g.bcast_bytes(bytes(range(128)) if rank == 0 else b"")       # how the RCCL unique id travels
        g.barrier()
        mx, mn = g.max(float(rank + 1)), g.min(float(rank + 1))
        tot = g.sum_int([rank, 1, 10 * rank])
        arr = g.allgather_array(np.full((2, 3), rank, np.int32))
        objs = g.allgather_object({"rank": rank, "text": "x" * (rank * 1000)})
        big = g.allgather_array(np.arange(300_000, dtype=np.float64) + rank)   # multi-MB frames through the star
        ok = (uid == bytes(range(128)) and mx == world and mn == 1.0 and tot.tolist() == [sum(range(world)), world, 10 * sum(range(world))]
              and arr.shape == (2 * world, 3) and [int(v) for v in arr[::2, 0]] == list(range(world))
              and [o["rank"] for o in objs] == list(range(world)) and len(objs[-1]["text"]) == (world - 1) * 1000
              and big.shape == (300_000 * world,) and big[-1] == 299_999 + world - 1)
        g.barrier()
    open(os.path.join(out_dir, f"r{rank}.txt"), "w").write(str(ok))


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_rendezvous_collectives(tmp_path, world):
    _run_ranks(_rdv_worker, world, str(tmp_path))
    assert [(tmp_path / f"r{r}.txt").read_text() for r in range(world)] == ["True"] * world


def test_rendezvous_refuses_a_directory_it_does_not_own_outright(tmp_path):
    """The rendezvous directory holds the job's token: a reader refuses one that other users can enter, a symlink, a hub file
    with group / world bits, and (when the test can chown) one that belongs to someone else -- instead of trusting whatever
    hub.json it finds there (the default path is predictable)."""
    import json
    from same_amd.rendezvous import HostGroup, _check_private

    def reader(d):
        return HostGroup(1, 2, rdv_dir=str(d), timeout=5)

    hub = {"port": 1, "token": "0" * 32, "world": 2}
    open_dir = tmp_path / "open"
    open_dir.mkdir(mode=0o755)
    os.chmod(open_dir, 0o755)
    (open_dir / "hub.json").write_text(json.dumps(hub))
    os.chmod(open_dir / "hub.json", 0o600)
    with pytest.raises(PermissionError, match="accessible to other users"):
        reader(open_dir)
    priv = tmp_path / "priv"
    priv.mkdir(mode=0o700)
    (priv / "hub.json").write_text(json.dumps(hub))
    os.chmod(priv / "hub.json", 0o644)
    with pytest.raises(PermissionError, match="rendezvous file"):
        reader(priv)
    link = tmp_path / "link"
    os.symlink(priv, link)
    with pytest.raises(PermissionError, match="not a plain directory"):
        reader(link)
    if os.geteuid() == 0:
        other = tmp_path / "other"
        other.mkdir(mode=0o700)
        (other / "hub.json").write_text(json.dumps(hub))
        os.chmod(other / "hub.json", 0o600)
        os.chown(other, 12345, 12345)
        with pytest.raises(PermissionError, match="belongs to uid 12345"):
            reader(other)
        # rank 0 does not adopt such a directory either
        with pytest.raises(PermissionError):
            HostGroup(0, 2, rdv_dir=str(other), timeout=2)
    # a directory of ours that was merely created too open is made private by rank 0, not refused
    _check_private(str(priv), "dir", want_dir=True)


def test_allgather_object_carries_frames_without_pickle(tmp_path):
    """The object exchange is JSON + .npy bytes (allow_pickle=False): dicts with non-string keys, tuples, sets, numpy arrays and
    the per-window match tables (pandas frames with string / float / bool columns and a non-default index) come back equal,
    and something that is not data is refused at the sender."""
    import json
    import pandas as pd
    from same_amd import rendezvous as rdv

    df = pd.DataFrame({"aligned_idx": np.arange(4), "cell_type": ["a", "b", None, "d"], "cost": [1.5, np.nan, 2.0, 3.0],
                       "flag": np.array([True, False, True, True])}, index=[3, 5, 7, 9])
    obj = {"k": (1, 2.5, "s"), 3: [df, None, {1, 2}], "arr": np.arange(6, dtype=np.int32).reshape(2, 3), "empty": pd.DataFrame(),
           "raw": bytes(range(5)), "np": np.float32(0.25)}
    wire = rdv.pack_object(obj)
    assert b"pickle" not in wire and b"__reduce__" not in wire
    back = rdv.unpack_object(wire)
    assert back["k"] == (1, 2.5, "s") and back[3][1] is None and back[3][2] == {1, 2} and back["raw"] == bytes(range(5)) and back["np"] == 0.25
    assert back["arr"].dtype == np.int32 and np.array_equal(back["arr"], obj["arr"]) and back["empty"].shape == (0, 0)
    pd.testing.assert_frame_equal(back[3][0], df)
    with pytest.raises(TypeError):
        rdv.pack_object({"f": open})
    with pytest.raises(ValueError):
        rdv.unpack_object(wire[:-5])                       # a truncated frame is refused, not half-read
    assert "import pickle" not in open(rdv.__file__).read()
    with rdv.HostGroup(0, 1) as g:
        assert g.allgather_object(obj)[0]["k"] == (1, 2.5, "s")


def test_rendezvous_dir_is_keyed_by_launcher(monkeypatch):
    """Without SAME_RDV_DIR (the torch.distributed.run case) every worker of one agent derives the same directory from
    MASTER_PORT + the parent's pid and start time; another port gives another directory."""
    from same_amd import rendezvous

    monkeypatch.delenv("SAME_RDV_DIR", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29500")
    a = rendezvous.default_rdv_dir()
    assert a == rendezvous.default_rdv_dir() and str(os.getppid()) in a and "29500" in a
    monkeypatch.setenv("MASTER_PORT", "29501")
    assert rendezvous.default_rdv_dir() != a
    monkeypatch.setenv("SAME_RDV_DIR", "/tmp/given")
    assert rendezvous.default_rdv_dir() == "/tmp/given"


def test_rendezvous_rejects_strangers(tmp_path):
    """A connection that does not present the job's token is dropped and does not take a rank's place."""
    import socket
    import threading
    import time

    from same_amd.rendezvous import HostGroup

    rdv = str(tmp_path)
    res = {}

    def r0():
        with HostGroup(0, 2, rdv_dir=rdv, timeout=60) as g:
            res["got"] = g.allgather_bytes(b"zero")

    t = threading.Thread(target=r0)
    t.start()
    hub = os.path.join(rdv, "hub.json")
    for _ in range(2000):
        if os.path.exists(hub):
            break
        time.sleep(0.005)
    port = json.load(open(hub))["port"]
    with socket.create_connection(("127.0.0.1", port)) as s:   # stranger: wrong token
        s.sendall(b"0" * 32 + (1).to_bytes(4, "little"))
        time.sleep(0.1)
    with HostGroup(1, 2, rdv_dir=rdv, timeout=60) as g1:
        got1 = g1.allgather_bytes(b"one")
    t.join(60)
    assert res["got"] == [b"zero", b"one"] == got1


def _table_worker(rank, world, rdv, out_dir):
    _setup_paths()
    from same_amd.dist import allgather_table
    from same_amd.rendezvous import HostGroup

    with HostGroup(rank, world, rdv_dir=rdv, timeout=120) as g:
        rng = np.random.default_rng(100 + rank)
        n = [0, 7, 1000, 33][rank % 4] + 100 * (rank == world - 1)           # ragged: an empty table and different lengths
        mine = {"Aligned": rng.integers(0, 2 ** 62, n), "X": rng.random(n), "viol": (rng.random(n) < 0.5).astype(np.uint8),
                "window_id": np.full(n, rank, np.int64)}
        every = allgather_table(None, None, g, mine)                            # ctx None: the same blocks over the host group
        ok = len(every) == world
        for r, t in enumerate(every):
            want_rng = np.random.default_rng(100 + r)
            m = [0, 7, 1000, 33][r % 4] + 100 * (r == world - 1)
            ok = ok and list(t) == list(mine) and np.array_equal(t["Aligned"], want_rng.integers(0, 2 ** 62, m)) and len(t["X"]) == m \
                and (t["window_id"] == r).all() and t["viol"].dtype == np.uint8
        g.barrier()
    open(os.path.join(out_dir, f"t{rank}.txt"), "w").write(str(bool(ok)))


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_allgather_table_blocks_over_the_host_group(tmp_path, world):
    """The packing of dist.allgather_table (the cfg 5 exchange of the ranks' match tables) at several world sizes, ragged and empty
    tables included -- on the GPU the same blocks travel by ncclAllGather (tests/test_gpu_run_same.py)."""
    _run_ranks(_table_worker, world, str(tmp_path))
    assert [(tmp_path / f"t{r}.txt").read_text() for r in range(world)] == ["True"] * world


# ---------------------------------------------------------------------------------------------- bench.py launcher
@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_its_own_ranks(world):
    """`python3 bench.py --gpus N` (how the driver calls it): the parent spawns N ranks, they rendezvous, exchange the id,
    barrier and max-reduce, and exactly one JSON line comes back on stdout.  --dry-launch stops before any GPU work."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SAME_RDV_DIR")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["world"] == world and d["max_of_rank_plus_1"] == float(world)
    assert [r["rank"] for r in d["ranks"]] == list(range(world)) and all(r["id_ok"] for r in d["ranks"])
    assert len({r["pid"] for r in d["ranks"]}) == world            # fresh processes, one per rank
    # the contract of the real N > 1 line (bench.py refuses to write a line that lacks one of these; the GPU suite checks their contents)
    assert set(d["line_keys_at_n_gt_1"]) == {"rccl", "gather", "gather_hidden_ms", "per_rank_dense_ms", "per_rank", "strong_cfg4", "cfg5"}
    # every rank was told how many share the host: the Qhull helper budget is divided by it (same_amd/qhull_pool.py)
    assert [r["local_world"] for r in d["ranks"]] == [world] * world and sum(r["qhull_helpers"] for r in d["ranks"]) <= max(24, world)
    assert {"value", "ms_per_step", "gather", "gather_hidden_ms", "per_rank_dense_ms", "parity_spot_check"} <= set(d["strong_record_keys"])


@pytest.mark.parametrize("world", [2, 8])
def test_bench_as_ranks_of_torch_distributed_run(world):
    """The driver's N > 1 command, verbatim: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- every process is a rank (RANK / LOCAL_RANK / WORLD_SIZE from the agent), the ranks
    find each other through the directory derived from MASTER_PORT and the agent's pid (no SAME_RDV_DIR), pass its ownership
    checks, and rank 0 prints the one line.  --dry-launch stops before any GPU work.  (torch is the LAUNCHER here, nothing more:
    bench.py and the package import none of it.)"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SAME_RDV_DIR", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
                        "--dry-launch"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["world"] == world and d["max_of_rank_plus_1"] == float(world) and all(r["id_ok"] for r in d["ranks"])
    assert [r["rank"] for r in d["ranks"]] == list(range(world)) and len({r["pid"] for r in d["ranks"]}) == world


def test_bench_launcher_reports_a_failed_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SAME_RDV_DIR")}
    # without --dry-launch the ranks need a GPU: here every rank exits non-zero, and so must the launcher (no line)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--no-cpu-baseline"],
                       env=dict(env, SAME_BENCH_RDV_TIMEOUT="60"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_bench_is_torch_free():
    src = "".join(open(os.path.join(ROOT, *parts)).read() for parts in (("bench.py",), ("same_amd", "dist.py"), ("same_amd", "rendezvous.py"),
                                                                        ("same_amd", "bench_common.py"), ("same_amd", "bench_launch.py"),
                                                                        ("same_amd", "bench_cfg5.py"), ("same_amd", "bench_problem.py"), ("same_amd", "qhull_pool.py")))
    assert "import torch" not in src and "from torch" not in src


# ---------------------------------------------------------------------------------------------- sharded cost build
def _knn_worker(rank, world, rdv, kind, case, out_dir):
    _setup_paths()
    if kind == "gloo" and rank:
        _wait_port(rdv)
    from conftest import frames_from_golden, load_golden
    from oracle import same_oracle as orc
    from same_amd.dist import pairs_and_costs, sharded_knn_cost_host

    group = _make_group(kind, rank, world, rdv)
    try:
        g = load_golden(case)
        a_df, r_df, cols = frames_from_golden(g)
        radius, k, w = g["params"][0], int(g["params"][1]), g["params"][3]
        A, R = a_df[cols].to_numpy(), r_df[cols].to_numpy()
        axy, rxy = a_df[["X", "Y"]].to_numpy(), r_df[["X", "Y"]].to_numpy()

        def compute_block(b, e):
            idx, _, _ = orc.knn_prune(axy, rxy, radius, k, b, e)
            cost = np.full(idx.shape, np.inf)
            rr, cc = np.nonzero(idx >= 0)
            cost[rr, cc] = orc.pair_cost_arrays(A, R, axy, rxy, np.column_stack((rr + b, idx[rr, cc])), w)
            return idx, cost

        idx, cost = sharded_knn_cost_host(compute_block, len(a_df), k, group)
        na, nr, pairs, c = pairs_and_costs(a_df, r_df, idx, cost)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pairs=np.asarray(pairs, dtype=np.int64), costs=np.array(c),
                 kept_a=na["__row"].to_numpy(), kept_r=nr["__row"].to_numpy())
    finally:
        group.close()


@pytest.mark.parametrize("kind,world,case", [("tcp", 2, "cfg1_500"), ("tcp", 3, "cfg1_500"), ("tcp", 2, "cfg2_small"),
                                             ("tcp", 3, "cfg2_small"), ("tcp", 8, "cfg1_500"),   # G in {1 (fixtures), 2, 8}
                                             ("gloo", 2, "cfg2_small")])
def test_sharded_knn_cost(tmp_path, kind, world, case):
    from conftest import load_golden

    _run_ranks(_knn_worker, world, kind, case, str(tmp_path))
    g = load_golden(case)
    for rank in range(world):
        out = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(out["pairs"], g["pairs"])          # identical to the reference's single-process result
        assert np.array_equal(out["costs"], g["all_costs"])
        assert np.array_equal(out["kept_a"], g["kept_aligned"]) and np.array_equal(out["kept_r"], g["kept_ref"])


# ---------------------------------------------------------------------------------------------- sharded sweeps
def _sweep_worker(rank, world, rdv, kind, case, out_dir):
    _setup_paths()
    if kind == "gloo" and rank:
        _wait_port(rdv)
    import pandas as pd
    from conftest import frames_from_golden, load_golden
    from oracle import same_oracle as orc
    from same_amd import dist, triangles

    group = _make_group(kind, rank, world, rdv)
    try:
        g = load_golden(case)
        a_df, r_df, _ = frames_from_golden(g)
        na = a_df.iloc[g["kept_aligned"]].reset_index(drop=True)
        nr = r_df.iloc[g["kept_ref"]].reset_index(drop=True)
        tris, sign = g["tri_plain"], g["source_signs"].astype(np.int8)
        rxy = nr[["X", "Y"]].to_numpy()
        match, _ = orc.matching_from_x(g["x_vals"], g["pairs"], len(na))     # src/same.py:634-639
        # a10: flags per triangle block -> gathered flags -> checked + ascending flipped list
        checked, viol, flags = dist.sharded_orient_sweep_host(
            lambda t0, t1: orc.orient_sweep(tris[t0:t1], sign[t0:t1], rxy, match)[2], len(tris), group)
        # a11: the whole report, triangle loop split over the ranks
        ch = g["greedy_chosen"]
        m_df = pd.DataFrame({"aligned_idx": ch[:, 0], "ref_idx": ch[:, 1]})
        info = triangles.precompute_triangle_info(na, tris, triangles.build_simplex_map(len(na), tris))
        rep = dist.sharded_verify_spatial_preservation(na, nr, m_df, info, group, block_sweep=orc.xyorder_sweep)
        # a12
        before, after, flipped, m3 = dist.sharded_triangle_area_flips(na, nr, tris, {int(i): int(j) for i, j, _ in ch}, group,
                                                                      block_sweep=orc.area_flip)
        n = len(tris)
        s = rep["violation_summary"]
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), checked=checked, viol=viol, flags=flags,
                 vx=np.array([[v["triangle_idx"], v["point1"]["aligned_idx"], v["point2"]["aligned_idx"], v["point1"]["ref_idx"],
                               v["point2"]["ref_idx"]] for v in rep["x_order_violations"]], dtype=np.int64).reshape(-1, 5),
                 vy=np.array([[v["triangle_idx"], v["point1"]["aligned_idx"], v["point2"]["aligned_idx"], v["point1"]["ref_idx"],
                               v["point2"]["ref_idx"]] for v in rep["y_order_violations"]], dtype=np.int64).reshape(-1, 5),
                 vtris=np.sort(np.array(rep["triangles_with_violations"], dtype=np.int64)),
                 vpoints=np.sort(np.array(rep["points_with_violations"], dtype=np.int64)),
                 summary=np.array([s["total_triangles"], s["violated_triangles"], s["total_comparisons"], s["total_violations"]]),
                 before=np.array([before[t] for t in range(n)]),
                 after=np.array([np.nan if after[t] is None else after[t] for t in range(n)]),
                 flipped=np.array(flipped, dtype=np.int64), m3=np.array([m3[t] for t in range(n)], dtype=np.uint8))
    finally:
        group.close()


@pytest.mark.parametrize("kind,world,case", [("tcp", 1, "cfg2_small"), ("tcp", 2, "cfg2_small"), ("tcp", 8, "cfg2_small"),
                                             ("tcp", 3, "cfg1_500"), ("tcp", 8, "synthetic_example"), ("gloo", 2, "cfg2_small")])
def test_sharded_sweeps_equal_single_process_reference(tmp_path, kind, world, case):
    """Triangle-block sharding of the three sweeps (the loops at src/same.py:645-669, src/violationhelper.py:53-117,
    src/same.py:1362-1402) reproduces the reference's single-process outputs on every rank, for G in {1, 2, 3, 8}."""
    from conftest import load_golden

    _run_ranks(_sweep_worker, world, kind, case, str(tmp_path))
    g = load_golden(case)
    for rank in range(world):
        o = np.load(tmp_path / f"rank{rank}.npz")
        assert int(o["checked"]) == int(g["lazy_checked"][0])
        assert np.array_equal(o["viol"], g["lazy_violating"])            # ascending, as src/same.py:687-703 needs
        assert np.array_equal(np.flatnonzero(o["flags"] == 2), g["lazy_violating"])
        assert np.array_equal(o["vx"], g["viol_x"]) and np.array_equal(o["vy"], g["viol_y"])   # traversal order kept
        assert np.array_equal(o["vtris"], np.sort(g["viol_tris"])) and np.array_equal(o["vpoints"], np.sort(g["viol_points"]))
        assert np.array_equal(o["summary"], g["viol_summary"])
        assert np.array_equal(o["before"], g["area_before"]) and np.array_equal(o["after"], g["area_after"], equal_nan=True)
        assert np.array_equal(o["flipped"], g["area_flipped"]) and np.array_equal(o["m3"], g["area_matched3"])


def test_triangle_blocks_tile_the_list():
    from same_amd.dist import TRI_BLOCK_ALIGN, tri_block

    for n in (0, 1, 255, 256, 257, 1000, 2252, 100_000, 199_973):
        for world in (1, 2, 3, 8):
            spans = [tri_block(n, world, r) for r in range(world)]
            assert all(b % TRI_BLOCK_ALIGN == 0 for _, _, b in spans) and len({b for _, _, b in spans}) == 1
            covered = [t for b, e, _ in spans for t in range(b, e)] if n <= 3000 else None
            if covered is not None:
                assert covered == list(range(n))
            assert sum(e - b for b, e, _ in spans) == n
            for r, (b, e, blk) in enumerate(spans):
                assert e == b or b == r * blk          # a non-empty block starts at rank*block: position in the gather == index


# ---------------------------------------------------------------------------------------------- windows
def test_window_round_robin():
    from same_amd.windows import assign_windows

    plan = [{"n_ref": r, "n_mov": m} for r, m in ((10, 10), (50, 40), (5, 5), (30, 30), (20, 80), (1, 1), (60, 60))]
    shards = assign_windows(plan, 3)
    assert sorted(w for s in shards for w in s) == list(range(7))
    loads = [sum(plan[w]["n_ref"] * plan[w]["n_mov"] for w in s) for s in shards]
    assert max(loads) <= 3600 + 2000  # heaviest-first keeps the big windows apart
    assert assign_windows([], 4) == [[], [], [], []]


def _two_groups_worker(rank, rdv, out_dir):
    _setup_paths()
    import time

    from same_amd.rendezvous import HostGroup

    got = []
    for k in range(4):           # four groups one after the other in ONE rendezvous directory, the ranks out of step between them
        with HostGroup(rank, 2, rdv_dir=rdv, timeout=60) as g:
            got.append(g.allgather_object({"k": k, "rank": rank}))
            if rank == 0:
                time.sleep(0.3)  # rank 1 is already making its next group while this one's rendezvous file still stands
    ok = all(got[k] == [{"k": k, "rank": 0}, {"k": k, "rank": 1}] for k in range(4))
    open(os.path.join(out_dir, f"seq{rank}.txt"), "w").write(str(ok))


def test_groups_made_one_after_the_other_do_not_meet_each_others_files(tmp_path):
    """A rank that starts its next HostGroup while rank 0 has not yet closed the last one must not read the LAST group's rendezvous file
    (dist's wrappers make a group per call when none is passed): the files are numbered per group."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_two_groups_worker, args=(r, str(tmp_path / "rdv"), str(tmp_path))) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert [open(tmp_path / f"seq{r}.txt").read() for r in range(2)] == ["True", "True"]
