"""The main workload of bench.py resident on one rank -- inputs, output buffers, the caller-held prune index, the bound sweep -- and its
step (dense build + prune + candidate costs (+ gather) + triangle maps + the three sweeps).  bench.py holds the line and the
cpu_baseline leg; nothing here touches the oracle."""
import time

from .bench_common import stats3

WORKLOADS = {
    # name: (n_ref, aligned rows [per rank if weak, in total if strong], T, k, radius)
    "dense100k": (100_000, 100_000, 20, 32, 25.0),
    "cfg4": (200_000, 200_000, 20, 32, 25.0),
    "cfg2": (10_000, 10_000, 20, 32, 25.0),
    "tiny": (4_000, 4_000, 20, 32, 25.0),
    "cfg5": None,              # sliding windows, see run_cfg5
}
STRONG_OF = {"dense100k": "cfg4"}   # the ONE-problem configuration embedded after a weak run of the key (else: the same shape)
STRONG_CHUNK_BYTES = 40e9  # dense buffer of the strong mode (25k rows x 200k refs x 8 B)
SPREAD_MAX_T = 12          # --spread auto: type counts up to this are bound by the store stream (placement over the HBM regions pays)


def dense_kernel_label(dtype, T):
    """The kernel csrc/cost.hip dispatches to for this dtype and type count."""
    name = "double" if dtype == "f64" else "float"
    if T >= (48 if dtype == "f64" else 49):
        return f"dense_cost_rowblock_kernel<{name},32>"
    cpl = (2 if dtype == "f64" else 4)
    if T * cpl * (2 if dtype == "f64" else 1) > 160:
        cpl = 1
    return f"dense_cost_kernel<{name},{T},{cpl}>"


class Problem:
    """One workload resident on this rank: inputs, output buffers, the caller-held KNN index, the bound sweep -- and the step.

    weak: this rank owns its own section of `rows_cfg` aligned cells against the replicated refs.
    strong: ONE problem; every rank holds all aligned cells (XY / types are a few MB) and owns a row block of the work and a
    triangle block of the sweeps."""

    def __init__(self, env, name, strong, dense_buf=None):
        import ctypes

        import numpy as np
        from scipy.spatial import Delaunay

        from same_amd import synth
        from same_amd.dist import ShardedSweeps, row_block
        from same_amd.triangles import cos_threshold

        self.env, self.name, self.strong = env, name, strong
        args, group, ctx, tctx, comm = env.args, env.group, env.ctx, env.tctx, env.comm
        L, TH, chk = env.L, env.TH, env.chk
        self.np, self.ctypes = np, ctypes
        self.n_ref, self.rows_cfg, self.T, self.k, self.radius = WORKLOADS[name]
        n_ref, rows_cfg, T, k, radius = self.n_ref, self.rows_cfg, self.T, self.k, self.radius
        self.ref = ref = synth.make_cells(n_ref, T, seed=0)
        if strong:
            self.mov = mov = synth.make_cells(rows_cfg, T, seed=1, side=ref["side"])
            self.rb, self.re, self.block = row_block(rows_cfg, group.world, group.rank)
            self.n_mov = rows_cfg
        else:
            self.mov = mov = synth.make_cells(rows_cfg, T, seed=1 + group.rank, side=ref["side"])
            self.rb, self.re, self.block, self.n_mov = 0, rows_cfg, rows_cfg, rows_cfg
        rb, re, block, n_mov = self.rb, self.re, self.block, self.n_mov
        self.rows = rows = re - rb
        self.tris = tris = np.ascontiguousarray(Delaunay(mov["xy"]).simplices, dtype=np.int32)  # host input (Qhull), as in the reference
        self.Tr = Tr = len(tris)
        self.bufs = []   # everything allocated here, for close()

        def keep(b):
            self.bufs.append(b)
            return b

        self.dA, self.dR = keep(ctx.to_device(mov["types"])), keep(ctx.to_device(ref["types"]))
        self.dax, self.drx = keep(ctx.to_device(mov["xy"])), keep(ctx.to_device(ref["xy"]))
        self.dsize, self.dtype_id = keep(ctx.to_device(mov["size"])), keep(ctx.to_device(mov["cell_type"]))
        self.dtris = keep(ctx.to_device(tris))
        self.ld = ld = (n_ref + 1) & ~1
        self.chunk_rows = max(1, min(max(rows, 1), int(STRONG_CHUNK_BYTES // (ld * 8)))) if strong else rows
        need = max(self.chunk_rows, 1) * ld * 8
        self.spread = False
        if dense_buf is not None and dense_buf.nbytes >= need:
            self.dD, self.own_dense = dense_buf, False      # the resident block of the run's main problem, reused
        else:
            # the dense cost block (80 GB at dense100k).  A streaming store confined to one of the card's three HBM regions runs ~20 %
            # below one spread over them, so STORE-BOUND shapes (T <= SPREAD_MAX_T) take the block from same_dev_alloc_spread; where fp64
            # issue binds (the T = 20 headline: 16.88 ms plain against 17.04 ms spread) a plain hipMalloc is as fast, starts 0.4 s and
            # 404 probe launches sooner and does not depend on timed stores while other ranks start beside this one
            self.spread = args.spread == "on" or (args.spread == "auto" and T <= SPREAD_MAX_T)
            self.dD, self.own_dense = (ctx.alloc_spread(need) if self.spread else ctx.alloc(need)), True
        ta = lambda n: keep(tctx.alloc(n))
        self.didx, self.dcost, self.dcnt = ta(block * k * 4), ta(block * k * 8), ta(max(block, 1) * 4)
        chk(L.same_dev_memset(TH, self.didx.ptr, 0xFF, self.didx.nbytes), "memset")   # rows past a short last block stay -1
        self.knn_index = ctypes.c_void_p()   # caller-held grid index of the reference cells: built once, reused by every prune
        chk(L.same_knn_index_build(TH, self.drx.ptr, n_ref, radius, ctypes.byref(self.knn_index)), "same_knn_index_build")
        self.gidx = self.gcost = None
        if comm is not None:
            self.gidx, self.gcost = ta(block * k * 4 * group.world), ta(block * k * 8 * group.world)
        self.dcls, self.dperim, self.dmaxcos = ta(Tr), ta(Tr * 8), ta(Tr * 8)
        self.dsign, self.dweight = ta(Tr), ta(Tr * 8)
        self.dedge, self.dtflag, self.dpflag, self.dcounts = ta(Tr * 3), ta(Tr), ta(n_mov), ta(32)
        self.dbefore, self.dafter, self.dm3, self.dflip = ta(Tr * 8), ta(Tr * 8), ta(Tr * 3), ta(Tr)
        self.dmatch = ta(n_mov * 4)
        self.en, self.thr = cos_threshold(15)
        # source signs + the resident sweep state (one untimed pass)
        ctx.sync()   # the uploads above went through the dense context's stream
        chk(L.same_tri_sign_weight_dev(TH, self.dax.ptr, self.dsize.ptr, self.dtris.ptr, Tr, self.dsign.ptr, self.dweight.ptr), "sign")
        self.sign0 = self.dsign.download((Tr,), np.int8)
        self.sweep = ctypes.c_void_p()
        chk(L.same_sweep_bind(TH, tris.ctypes.data, Tr, self.sign0.ctypes.data, ref["xy"].ctypes.data, n_ref, n_mov, None, 0,
                              ctypes.byref(self.sweep)), "bind")
        self.sharded = ShardedSweeps(tctx, comm, self.sweep, self.dax, self.drx, self.dtris, Tr,
                                     n_mov) if (strong and comm is not None) else None
        self.checked, self.nviol = ctypes.c_int64(0), ctypes.c_int64(0)
        self.viol = np.empty(max(Tr, 1), np.int32)
        self.last = {"checked": 0, "viol": self.viol[:0]}
        self.n_chunks = len(range(rb, re, self.chunk_rows))
        self.gather_ms = []     # per step: device (or, host transport, wall) time of the candidate-list all-gather
        self.gather_bytes = 0
        self.use_q32 = args.dense == "q32"
        if self.use_q32:
            from same_amd import ops

            if T > 32:
                raise SystemExit("--dense q32 supports T <= 32")
            self.q_off, self.q_l2 = ops.quantize_types(mov["types"], ref["types"])
            self.dAq, self.dRq = keep(ctx.alloc(mov["types"].size * 4)), keep(ctx.alloc(ref["types"].size * 4))
            chk(L.same_quantize_u32_dev(env.H, self.dA.ptr, mov["types"].size, self.q_off, 2.0 ** self.q_l2, self.dAq.ptr), "quantize")
            chk(L.same_quantize_u32_dev(env.H, self.dR.ptr, ref["types"].size, self.q_off, 2.0 ** self.q_l2, self.dRq.ptr), "quantize")
        if not strong:   # candidate matching for the sweeps: nearest reference within the radius (from one untimed prune)
            self.prune_and_costs()
            chk(L.same_first_candidate_dev(TH, self.didx.ptr, n_mov, k, self.dmatch.ptr), "match")

    # ---- the pieces of a step --------------------------------------------------------------------------------------
    def dense_launch(self, c0, c1):
        e, L, H = self.env, self.env.L, self.env.H
        if self.use_q32:
            return L.same_dense_cost_q32_dev(H, self.dAq.ptr, self.dRq.ptr, self.dA.ptr, self.dR.ptr, self.T, self.dax.ptr, self.drx.ptr,
                                             self.n_ref, c0, c1, 1.0, 2.0 ** -self.q_l2, 1e-6, self.dD.ptr, self.ld)
        return L.same_dense_cost_f64_dev(H, self.dA.ptr, self.dR.ptr, self.T, self.dax.ptr, self.drx.ptr, self.n_ref, c0, c1, 1.0,
                                         self.dD.ptr, self.ld)

    def dense_all(self, timed=None):
        """Enqueue the dense build of this rank's rows (strong mode: in chunks through the one buffer).  With `timed`, HIP
        events on the dense stream bracket the launch(es); dense_time() reads them after the rest of the step was issued."""
        L, H, chk = self.env.L, self.env.H, self.env.chk
        if timed is not None:
            chk(L.same_timer_start(H), "timer")
        for c0 in range(self.rb, self.re, self.chunk_rows):
            chk(self.dense_launch(c0, min(c0 + self.chunk_rows, self.re)), "dense")
        if timed is not None:
            chk(L.same_timer_mark(H), "timer")

    def dense_time(self, timed):
        if timed is not None:
            ms = self.ctypes.c_float(0)
            self.env.chk(self.env.L.same_timer_read(self.env.H, self.ctypes.byref(ms)), "timer")
            timed.append((ms.value / max(self.n_chunks, 1), self.rows / max(self.n_chunks, 1)))   # per launch

    def prune_and_costs(self):
        L, TH, chk = self.env.L, self.env.TH, self.env.chk
        chk(L.same_knn_prune_indexed_dev(TH, self.knn_index, self.dax.ptr, self.rb, self.re, self.k, self.didx.ptr, None, self.dcnt.ptr),
            "knn")
        chk(L.same_padded_cost_f64_dev(TH, self.dA.ptr, self.dR.ptr, self.T, self.dax.ptr, self.drx.ptr, self.rb, self.re, self.k,
                                       self.didx.ptr, 1.0, self.dcost.ptr), "padded")

    def tri_maps(self):
        L, TH, chk = self.env.L, self.env.TH, self.env.chk
        chk(L.same_tri_classify_dev(TH, self.dax.ptr, self.dtris.ptr, self.Tr, self.radius, self.en, self.thr, self.dtype_id.ptr,
                                    self.dcls.ptr, self.dperim.ptr, self.dmaxcos.ptr), "cls")
        chk(L.same_tri_sign_weight_dev(TH, self.dax.ptr, self.dsize.ptr, self.dtris.ptr, self.Tr, self.dsign.ptr, self.dweight.ptr), "sign")

    def local_sweeps(self):
        L, TH, chk, c = self.env.L, self.env.TH, self.env.chk, self.ctypes
        chk(L.same_xyorder_sweep_dev(TH, self.dax.ptr, self.n_mov, self.drx.ptr, self.dtris.ptr, self.Tr, self.dmatch.ptr, self.dedge.ptr,
                                     self.dtflag.ptr, self.dpflag.ptr, self.dcounts.ptr), "xy")
        chk(L.same_area_flip_dev(TH, self.dax.ptr, self.drx.ptr, self.dtris.ptr, self.Tr, self.dmatch.ptr, self.dbefore.ptr,
                                 self.dafter.ptr,
                                 self.dm3.ptr, self.dflip.ptr), "area")
        chk(L.same_orient_sweep_dev(self.sweep, self.dmatch.ptr, c.byref(self.checked), self.viol.ctypes.data, c.byref(self.nviol)),
            "orient")
        self.last["checked"], self.last["viol"] = self.checked.value, self.viol[: self.nviol.value]

    def _gather_read(self):
        ms, nb = self.env.comm.gather_time()
        self.gather_ms.append(ms)
        self.gather_bytes = nb

    def step(self, timed=None, gather=True):
        comm, k, block = self.env.comm, self.k, self.block
        gather = gather and comm is not None
        self.dense_all(timed=timed)
        if self.strong:
            self.prune_and_costs()
            if gather:
                comm.wait()                                   # closes the previous step's timing batch
                comm.allgather_dev(self.didx, self.gidx, block * k * 4)
                comm.allgather_dev(self.dcost, self.gcost, block * k * 8)
                if comm.synchronous:
                    self._gather_read()
            # the common matching: nearest reference of every aligned cell, from the gathered lists (identical on every rank;
            # a loop run without the gather reuses the lists of the last gather)
            src = self.gidx if comm is not None else self.didx
            self.env.chk(self.env.L.same_first_candidate_dev(self.env.TH, src.ptr, self.n_mov, k, self.dmatch.ptr), "match")
            self.tri_maps()
            if self.sharded is not None:
                self.last["checked"], self.last["viol"] = self.sharded.run(self.dmatch)
            else:
                self.local_sweeps()
            self.dense_time(timed)
            if gather and not comm.synchronous:
                self._gather_read()
        else:
            if gather:
                comm.wait()   # the previous step's gather (still reading didx/dcost) overlapped the dense build above
            self.prune_and_costs()
            if gather:  # on the communication stream: overlaps the sweeps below and the next step's dense build
                comm.allgather_dev_async(self.didx, self.gidx, block * k * 4)
                comm.allgather_dev_async(self.dcost, self.gcost, block * k * 8)
            self.tri_maps()
            self.local_sweeps()
            self.dense_time(timed)       # waits for the dense kernel: the gather issued above has long finished by then
            if gather:
                self._gather_read()

    def timed_loop(self, steps, warmup, gather=True):
        """warm-up, barrier + sync, exactly `steps` steps, sync + barrier; -> (seconds, max over ranks; [(dense ms, rows)] of this rank)."""
        env = self.env
        for _ in range(warmup):
            self.step(gather=gather)
        env.ctx.sync()
        env.tctx.sync()
        env.group.barrier()
        self.gather_ms = []
        dense_ms = []
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(timed=dense_ms, gather=gather)
        env.ctx.sync()
        env.tctx.sync()
        env.group.barrier()
        return env.group.max(time.perf_counter() - t0), dense_ms

    def transport_check(self):
        """Did the exchange deliver the right rows to the right place?  Rank 0 recomputes the first rows of the LAST rank's
        block on its own GPU (weak mode: from that rank's seed) and compares them with what the gather put into its own copy of
        the gathered lists, bit for bit.  (A transport check; the arithmetic itself is checked at N=1.)  -> description."""
        from same_amd import synth

        env, np = self.env, self.np
        L, TH, chk, tctx, group = env.L, env.TH, env.chk, env.tctx, env.group
        tctx.sync()
        peer, k = group.world - 1, self.k
        S = min(2000, self.block)
        if self.strong:
            peer_mov, p0 = self.mov, peer * self.block
            S = max(0, min(S, self.n_mov - p0))
        else:
            peer_mov, p0 = synth.make_cells(self.rows_cfg, self.T, seed=1 + peer, side=self.ref["side"]), 0
        if S <= 0:
            return "transport: the last rank's block is empty, nothing to compare"
        pA, pxy = tctx.to_device(peer_mov["types"]), tctx.to_device(peer_mov["xy"])
        pidx, pcost, pcnt = tctx.alloc(S * k * 4), tctx.alloc(S * k * 8), tctx.alloc(S * 4)
        chk(L.same_knn_prune_indexed_dev(TH, self.knn_index, pxy.ptr, p0, p0 + S, k, pidx.ptr, None, pcnt.ptr), "knn")
        chk(L.same_padded_cost_f64_dev(TH, pA.ptr, self.dR.ptr, self.T, pxy.ptr, self.drx.ptr, p0, p0 + S, k, pidx.ptr, 1.0, pcost.ptr),
            "padded")
        want_i, want_c = pidx.download((S, k), np.int32), pcost.download((S, k), np.float64)
        got_i = self.gidx.download((S, k), np.int32, offset_bytes=peer * self.block * k * 4)
        got_c = self.gcost.download((S, k), np.float64, offset_bytes=peer * self.block * k * 8)
        for b in (pA, pxy, pidx, pcost, pcnt):
            b.free()
        if not (np.array_equal(got_i, want_i) and np.array_equal(got_c, want_c)):
            raise SystemExit(f"gathered candidate lists of rank {peer} differ from a local recomputation: refusing to report a number")
        return (f"transport: rows [0,{S}) of rank {peer}'s block in rank 0's gathered lists (idx + cost) equal a local recomputation "
                "bit for bit; arithmetic parity is the N=1 run's check")

    def workload_text(self):
        w = self.env.group.world
        shape = (f"ONE problem of {self.n_mov} aligned x {self.n_ref} ref cells, aligned-row blocks and triangle blocks over {w} rank(s), "
                 f"dense build in {self.chunk_rows}-row chunks" if self.strong else f"{self.rows} aligned x {self.n_ref} ref cells per GPU")
        arith = ("fixed-point (2^-%d grid, every output within 1e-6 relative of the fp64 one) " % self.q_l2) if self.use_q32 else "fp64 "
        return (f"{self.name}: {shape}, T={self.T} type cols, {arith}dense L1 cost + r={self.radius:g}/k={self.k} KNN prune + pair costs + "
                f"{self.Tr} Delaunay triangles classify/sign + orientation / XY-order / area-flip sweeps")

    def respread(self):
        """Replace a plain cost block of this problem by one laid over the HBM regions (the store-bound entries of the T sweep are
        measured on such a block); the contents are lost.  No-op when the block is spread already or is not this problem's own."""
        if self.spread or not self.own_dense:
            return self.dD
        self.env.ctx.sync()
        nbytes = self.dD.nbytes
        self.dD.free()
        self.dD, self.spread = self.env.ctx.alloc_spread(nbytes), True
        return self.dD

    def close(self, keep_dense=False):
        L = self.env.L
        self.env.ctx.sync()
        self.env.tctx.sync()
        L.same_sweep_unbind(self.sweep)
        L.same_knn_index_destroy(self.knn_index)
        if self.sharded is not None:
            self.sharded.close()
        for b in self.bufs:
            b.free()
        if self.own_dense and not keep_dense:
            self.dD.free()


def gather_report(prob, dt_with, steps_with, dt_without, steps_without):
    """`gather`: the candidate-list all-gather by itself (events on its stream) and what it costs the step."""
    env = prob.env
    ms = stats3(prob.gather_ms)
    world = env.group.world
    with_ms, without_ms = dt_with / steps_with * 1e3, (dt_without / steps_without * 1e3 if dt_without is not None else None)
    out = {"bytes_per_rank": int(prob.gather_bytes), "bytes_total": int(prob.gather_bytes) * world,
           "ms": ms[1] if ms else None, "ms_min_mean_max": ms,
           "GBs": (prob.gather_bytes * world / (ms[1] * 1e-3) / 1e9) if ms and ms[1] > 0 else None,
           "GBs_means": "bytes every rank ends up holding (nranks x bytes_per_rank) over the gather's time on this rank",
           "timed_with": ("HIP events on the stream the all-gathers run on (same_comm_gather_time), rank 0" if not env.comm.synchronous
                          else "host wall time around the synchronous host-transport exchange, rank 0"),
           "step_ms_with_gather": with_ms, "step_ms_without_gather": without_ms,
           "steps_without_gather": steps_without if dt_without is not None else 0}
    return out, (with_ms - without_ms if without_ms is not None else None)
