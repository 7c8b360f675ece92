"""Row-stripe sharding of one large grid over the GPUs of a node (new-framework feature; the
reference only has independent clamped tiles, Scripts/MeshTileGenerator.cs:166-192).

Rank r of P owns rows [r*R/P, (r+1)*R/P) x all columns of a (grows x cols) grid.  The
noise stage needs no communication (world offsets, Noise/Fractal/Fractal.cs:109-116); every other
stage is a small-radius stencil.  Two ways of feeding the stencils their ghost rows:

  * haloMode "recompute" (default for the metric pipeline, whose source is closed-form noise): each
    rank evaluates the noise on its owned rows plus the total stencil radius of the pipeline
    (Gauss 17x2 + flow 5x2 + erosion 5 = 49 rows) and every launch produces a window that shrinks by
    the radius it consumed.  No data-path communication at all; ~2.4 % redundant rows per side at
    2048-row stripes.
  * haloMode "exchange": before each launch the ranks exchange the ghost rows that launch consumes
    with their two neighbours (rank-1 / rank+1 only) -- the form a pipeline needs whose source plane
    is not recomputable (an uploaded height map).
  * haloMode "exchange_once": the source plane is produced on the owned rows only, its ghost rows for
    the WHOLE pipeline (49 rows) are exchanged once, and everything after runs like "recompute" -- one
    latency-bound exchange instead of one per launch, for the price of the same redundant rows.

Because the kernels compute each cell with the same operations wherever it sits, either way the
sharded result equals the single-GPU monolithic result bit for bit; the only clamps are at the
global border.

The schedule below is host logic shared by two compute back ends:
  * HipStripeOps  -- the product path: C-ABI stripe entry points on device buffers (torch CUDA tensors
    used as plain HBM allocations), halo exchange by torch.distributed P2P on the `nccl` backend
    (= RCCL over xGMI);
  * any object with the same methods (tests/ supplies one built on the CPU oracle, driven over `gloo`).
"""
import copy
import ctypes as C

from . import _native as N


class StripePlan:
    """Geometry of one rank's stripe.  Every plane buffer has `halo` ghost rows above and below the
    owned rows (the ones hanging over the global border are never read)."""

    def __init__(self, rank, world, grows, cols, halo, neighbours_own_halo=True):
        assert 0 <= rank < world and grows >= world
        base, rem = divmod(grows, world)
        self.rank, self.world, self.grows, self.cols, self.halo = rank, world, grows, cols, halo
        self.g0 = rank * base + min(rank, rem)           # first owned global row
        self.nown = base + (1 if rank < rem else 0)
        self.rows = self.nown + 2 * halo                 # rows in every buffer
        self.own0, self.own1 = halo, halo + self.nown
        self.grow0 = self.g0 - halo
        # an exchange takes ghost rows from the adjacent rank only; recomputed ghost rows may reach further
        assert self.nown >= halo or not neighbours_own_halo, "stripe thinner than the halo"

    def stripe(self, pitch=0):
        return N.Stripe(self.cols, self.rows, self.grow0, self.grows, self.own0, self.own1, pitch)

    def widened(self, up, down):
        """The same buffer with the produced rows grown by `up` rows above and `down` rows below the
        owned ones, clipped to the global grid: what a launch has to produce so that later launches
        find their ghost rows without an exchange."""
        if up == 0 and down == 0:
            return self
        assert up <= self.halo and down <= self.halo
        v = copy.copy(self)
        v.own0 = max(self.own0 - up, -self.grow0)
        v.own1 = min(self.own1 + down, self.grows - self.grow0)
        v.g0, v.nown = self.grow0 + v.own0, v.own1 - v.own0
        return v

    def rows_window(self, a, b):
        """The same buffer producing only buffer rows [a, b) of the owned ones (a launch split into an interior part,
        which reads no ghost row, and the border parts that wait for the halo exchange)."""
        v = copy.copy(self)
        v.own0, v.own1 = max(a, self.own0), min(b, self.own1)
        v.g0, v.nown = self.grow0 + v.own0, max(0, v.own1 - v.own0)
        return v

    @property
    def up(self):
        return self.rank - 1 if self.rank > 0 else None

    @property
    def down(self):
        return self.rank + 1 if self.rank + 1 < self.world else None


def split_iterations(n, cap):
    """n applications in launches of at most `cap` (any count: stripes ping-pong between two buffers)."""
    launches = (n + cap - 1) // cap
    base, rem = divmod(n, launches)
    return [base + (1 if i < rem else 0) for i in range(launches)]


class PipelineParams:
    """The metric pipeline's stage parameters."""

    def __init__(self, noiseType=3, hurst=0.4, startingAmplitude=1.0, stepdown=2.0, detuneRate=0.0, octaves=13,
                 xpos=0, zpos=0, noiseSize=1700, filter=2, gaussIterations=17, flowIterations=5, normMin=0.0,
                 normMax=0.005, erosionIterations=5, haloMode="exchange"):
        assert haloMode in ("exchange", "recompute", "exchange_once")
        self.__dict__.update(locals())
        del self.__dict__["self"]


FLOW_PLANES = 5  # water, fN, fS, fE, fW: state buffers are [5, rows, cols]


def _launch_radii(ops, p):
    """Per launch of the schedule: (stage, fused applications, rows consumed above, rows consumed below)."""
    out = []
    if p.gaussIterations > 0:
        k_off = ops.kernel_filter_halo_rows(p.filter, 1)
        cap = max(1, ops.kernel_filter_max_fused(p.filter))
        out += [("gauss", T, T * k_off, T * k_off) for T in split_iterations(p.gaussIterations, cap)]
    if p.flowIterations > 0:
        # whole iterations fused on chip, <= flow_fused_max() per launch; a launch of n iterations reads
        # height and state 2n rows beyond the rows it produces
        out += [("flow", n, 2 * n, 2 * n) for n in split_iterations(p.flowIterations, ops.flow_fused_max())]
    left = p.erosionIterations
    while left > 0:
        E = min(left, ops.erosion_max_fused())
        out.append(("erosion", E, E, 0))                # the min window reaches upwards only
        left -= E
    return out


def halo_rows_needed(ops, p):
    """Ghost rows every plane buffer needs on each side: the widest single launch when ghost rows are
    exchanged before each launch, the whole pipeline's radius when they are recomputed."""
    radii = _launch_radii(ops, p)
    if p.haloMode in ("recompute", "exchange_once"):
        return max(sum(r[2] for r in radii), sum(r[3] for r in radii), 1)
    return max([max(r[2], r[3]) for r in radii] + [1])


def pipeline_steps(ops, plan, p, bufs, result, on_stage=None):
    """The sharded metric pipeline as a generator: yields (planes, up_rows, down_rows) wherever the
    ranks must exchange ghost rows (never in haloMode "recompute", once in "exchange_once"), runs the stripe
    kernels in between.
    bufs = (A, B, S0, S1): two height planes [plan.rows, cols] and two flow-state buffers
    [5, plan.rows, cols].  The plane whose owned rows hold the result is appended to `result`.
    `on_stage(name)` (optional) is called where a stage begins ("noise", "gauss", "flow", "erosion") and
    once at the end ("end"): the hook bench.py hangs its stream markers on."""
    A, B, S0, S1 = bufs
    mark = on_stage if on_stage is not None else (lambda name: None)
    current = "noise"
    mark(current)
    cur, nxt = A, B
    s_cur, s_nxt = S0, S1
    radii = _launch_radii(ops, p)
    recompute = p.haloMode in ("recompute", "exchange_once")   # launches produce shrinking windows, no per-launch exchange
    # rows beyond the owned ones that the remaining launches will still consume
    need_up = sum(r[2] for r in radii) if recompute else 0
    need_down = sum(r[3] for r in radii) if recompute else 0
    if p.haloMode == "exchange_once":
        ops.fractal(cur, plan, p)                 # stands for any source plane, e.g. an uploaded height map
        answer = yield [cur], need_up, need_down  # its ghost rows for the whole pipeline, once
        if answer == "async":
            yield ("finish",)                     # everything that follows reads them
    else:
        ops.fractal(cur, plan.widened(need_up, need_down), p)
    flow_launches = [i for i, r in enumerate(radii) if r[0] == "flow"]
    def split_launch(launch, win, up, down, answer):
        """Runs `launch(window)` once -- or, when the exchange that was just requested is asynchronous (the runner
        answered "async"), first on the interior rows, which read no ghost row, while the halos are in flight, then
        (after the runner has finished the exchange) on the border rows."""
        lo, hi = win.own0 + up, win.own1 - down
        if answer != "async" or hi <= lo:
            if answer == "async":
                yield ("finish",)
            launch(win)
            return
        launch(win.rows_window(lo, hi))
        yield ("finish",)
        if up > 0:
            launch(win.rows_window(win.own0, lo))
        if down > 0:
            launch(win.rows_window(hi, win.own1))

    for i, (stage, n, up, down) in enumerate(radii):
        if stage != current:
            current = stage
            mark(current)
        if recompute:
            need_up, need_down = need_up - up, need_down - down
        win = plan.widened(need_up, need_down)
        answer = None
        if stage == "gauss":
            if not recompute:
                answer = yield [cur], up, down
            yield from split_launch(lambda w: ops.kernel_filter(cur, nxt, w, p.filter, n), win, up, down, answer)
            cur, nxt = nxt, cur
        elif stage == "flow":
            first, last = i == flow_launches[0], i == flow_launches[-1]
            if not recompute:
                if first:  # height: exchanged once, read by every launch
                    widest = max(radii[j][2] for j in flow_launches)
                    answer = yield [cur], widest, widest
                else:
                    answer = yield [s_cur[k] for k in range(FLOW_PLANES)], up, down
            yield from split_launch(lambda w: ops.flow_fused(cur, s_cur, s_nxt, nxt, w, n, first, last, p.normMin, p.normMax),
                                    win, up, down, answer)
            s_cur, s_nxt = s_nxt, s_cur
            if last:
                cur, nxt = nxt, cur
        else:
            if not recompute:
                answer = yield [cur], up, down
            yield from split_launch(lambda w: ops.erosion(cur, nxt, w, n), win, up, down, answer)
            cur, nxt = nxt, cur
    mark("end")
    result.append(cur)


def run_pipeline(ops, comm, plan, p, bufs, on_stage=None):
    """One pass of the sharded metric pipeline on this rank; returns the plane holding the result.
    A comm with begin() / finish() exchanges asynchronously: the launch that needs the ghost rows runs its interior
    rows while they travel (RCCL P2P on the process group's own stream) and its border rows after finish()."""
    result = []
    gen = pipeline_steps(ops, plan, p, bufs, result, on_stage)
    overlap = getattr(comm, "overlap", False)
    pending = None
    try:
        req = next(gen)
        while True:
            if req == ("finish",):
                comm.finish(pending)
                pending = None
                req = next(gen)
                continue
            planes, up_rows, down_rows = req
            if overlap:
                pending = comm.begin(planes, plan, up_rows, down_rows)
                req = gen.send("async")
            else:
                comm.exchange(planes, plan, up_rows, down_rows)
                req = next(gen)
    except StopIteration:
        pass
    return result[0]


def run_pipeline_lockstep(ops_list, plans, p, bufs_list, copy_rows):
    """All ranks of a grid inside ONE process (tests, single-GPU rehearsal): the per-rank generators
    advance in lockstep and ghost rows are copied directly, copy_rows(dst_plane, d0, src_plane, s0, n)."""
    results = [[] for _ in plans]
    gens = [pipeline_steps(o, pl, p, b, r) for o, pl, b, r in zip(ops_list, plans, bufs_list, results)]
    while True:
        reqs = []
        for g in gens:
            try:
                reqs.append(next(g))
            except StopIteration:
                reqs.append(None)
        if all(r is None for r in reqs):
            break
        assert all(r is not None for r in reqs), "ranks left the schedule at different points"
        for r, (planes, up_rows, down_rows) in enumerate(reqs):
            pl = plans[r]
            for i, t in enumerate(planes):
                if up_rows > 0 and pl.up is not None:
                    q = plans[pl.up]
                    copy_rows(t, pl.own0 - up_rows, reqs[pl.up][0][i], q.own1 - up_rows, up_rows)
                if down_rows > 0 and pl.down is not None:
                    q = plans[pl.down]
                    copy_rows(t, pl.own1, reqs[pl.down][0][i], q.own0, down_rows)
    return [r[0] for r in results]


class HipStripeOps:
    """Stripe operations on device memory through the C ABI.  Buffers are objects with `.data_ptr()`
    (torch CUDA tensors used as plain HBM allocations); state buffers are indexable by plane."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.lib = N.lib

    def kernel_filter_halo_rows(self, filter, iterations):
        return self.lib.nz_kernel_filter_halo_rows(filter, iterations)

    def kernel_filter_max_fused(self, filter):
        return self.lib.nz_kernel_filter_max_fused(filter)

    def erosion_max_fused(self):
        return self.lib.nz_erosion_max_fused_iterations()

    def _call(self, name, *args):
        N.check(getattr(self.lib, name)(self.ctx._h, *args, 0, None), name)

    def fractal(self, buf, plan, p):
        st = plan.stripe()
        self._call("nz_fractal_stripe", p.noiseType, buf.data_ptr(), C.byref(st), p.hurst, p.startingAmplitude,
                   p.stepdown, p.detuneRate, p.octaves, p.xpos, p.zpos, p.noiseSize)

    def kernel_filter(self, src, dst, plan, filter, T):
        st = plan.stripe()
        self._call("nz_kernel_filter_stripe", src.data_ptr(), dst.data_ptr(), C.byref(st), filter, T)

    def erosion(self, src, dst, plan, E):
        st = plan.stripe()
        self._call("nz_erosion_stripe", src.data_ptr(), dst.data_ptr(), C.byref(st), E)

    def flow_fused_max(self):
        return self.lib.nz_flow_fused_max_iterations()

    def flow_fused(self, h, S_in, S_out, dst, plan, n, first, last, normMin, normMax):
        st = plan.stripe()
        arr = N.dev_ptr * FLOW_PLANES
        pin = arr(*[S_in[i].data_ptr() for i in range(FLOW_PLANES)])
        pout = arr(*[S_out[i].data_ptr() for i in range(FLOW_PLANES)])
        self._call("nz_flow_fused_stripe", h.data_ptr(), pin, pout, dst.data_ptr(), C.byref(st), n, int(first),
                   int(last), normMin, normMax)


    def map_range(self, buf, n, res, lim_min=float("inf"), lim_max=float("-inf")):
        """GetMapRangeJob over the first n floats of `buf` into the 3-float device buffer `res`."""
        self._call("nz_get_map_range", buf.data_ptr(), n, res.data_ptr(), lim_min, lim_max)

    def normalize_args(self, buf, plan, args):
        """NormalizeMap on the stripe's owned rows with args = device {min, max, range}."""
        owned = buf.data_ptr() + plan.own0 * plan.cols * 4
        self._call("nz_normalize_cells_dev", owned, plan.nown * plan.cols, args.data_ptr())


def map_range_work_floats(world):
    return 5 * world + 6


def global_map_range(ops, dist, plane, plan, res, work, lim_min=float("inf"), lim_max=float("-inf")):
    """{min, max, range} of the WHOLE grid in every rank's `res` (3 floats): GetMapRangeJob (Filter/NormalizeJob.cs:17-55)
    over the owned rows of `plane`, one all-gather of the per-rank {min, max, range}, and the same fold over the
    gathered minima and maxima in rank order -- which is the order the monolithic job walks the grid in, so the result
    (down to the sign of a zero extreme) is the monolithic one.  The one collective of the path: data-dependent
    normalisation needs every rank's extremes.  `work` = map_range_work_floats(world) floats on the device of `res`.
    The limits must not be NaN."""
    w = plan.world
    gathered, mins, maxs = work[:3 * w], work[3 * w:4 * w], work[4 * w:5 * w]
    lo, hi = work[5 * w:5 * w + 3], work[5 * w + 3:5 * w + 6]
    ops.map_range(plane[plan.own0:plan.own1], plan.nown * plan.cols, res)
    if dist is not None:  # one rank included: the collective is the same call at every world size
        dist.all_gather_into_tensor(gathered, res)
    else:
        gathered.copy_(res)
    mins.copy_(gathered.view(w, 3)[:, 0])
    maxs.copy_(gathered.view(w, 3)[:, 1])
    ops.map_range(mins, w, lo, lim_min, float("-inf"))
    ops.map_range(maxs, w, hi, float("inf"), lim_max)
    res[0:1].copy_(lo[0:1])
    res[1:2].copy_(hi[1:2])
    res[2:3].copy_(res[1:2] - res[0:1])  # IEEE fp32 subtraction, as `max_ - min_` in the job
    return res


class TorchComm:
    """Neighbour halo exchange with torch.distributed P2P (backend `nccl` = RCCL over xGMI on the GPU
    box, `gloo` in the CPU tests).  Each exchange is one grouped batch: at most two neighbours."""

    def __init__(self, dist, overlap=True):
        self.dist = dist
        self.overlap = overlap  # False: every exchange completes before the launch that follows is enqueued

    def begin(self, planes, plan, up_rows, down_rows):
        """Posts the batch; returns the requests.  On `nccl` the transfers run on the process group's stream behind
        everything already enqueued on the current stream, and finish() makes the current stream wait for them."""
        d = self.dist
        ops = []
        for t in planes:
            if up_rows > 0:  # my top ghost rows <- the rows just above, owned by rank-1
                if plan.down is not None:
                    ops.append(d.P2POp(d.isend, t[plan.own1 - up_rows:plan.own1], plan.down))
                if plan.up is not None:
                    ops.append(d.P2POp(d.irecv, t[plan.own0 - up_rows:plan.own0], plan.up))
            if down_rows > 0:  # my bottom ghost rows <- the rows just below, owned by rank+1
                if plan.up is not None:
                    ops.append(d.P2POp(d.isend, t[plan.own0:plan.own0 + down_rows], plan.up))
                if plan.down is not None:
                    ops.append(d.P2POp(d.irecv, t[plan.own1:plan.own1 + down_rows], plan.down))
        return d.batch_isend_irecv(ops) if ops else []

    def exchange(self, planes, plan, up_rows, down_rows):
        self.finish(self.begin(planes, plan, up_rows, down_rows))

    def finish(self, reqs):
        for req in reqs:
            req.wait()


class NoComm:
    """world == 1: nothing to exchange (clamp-to-edge at both borders)."""

    def exchange(self, planes, plan, up_rows, down_rows):
        pass


# ---- the same schedule behind the C ABI (nz_comm.cpp): what a C# / C++ host calls ----------------------------------------
HALO_MODES = {"recompute": N.NZ_HALO_RECOMPUTE, "exchange": N.NZ_HALO_EXCHANGE, "exchange_once": N.NZ_HALO_EXCHANGE_ONCE}
# record layout of nz_sharded_plan
OP_NOISE, OP_XBEGIN, OP_XFINISH, OP_FILTER, OP_FLOW, OP_EROSION, OP_MARK = 1, 2, 3, 4, 5, 6, 7


def rccl_version():
    v = C.c_int32(0)
    N.check(N.lib.nz_comm_rccl_version(C.byref(v)), "nz_comm_rccl_version")
    return v.value


class NativeComm:
    """nz_comm: an RCCL communicator (ncclCommInitRank) on the context's device plus its own stream.  `unique_id()` is
    made by ONE rank and handed to the others out of band (bench.py broadcasts it through torch.distributed)."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * N.NZ_COMM_ID_BYTES)()
        N.check(N.lib.nz_comm_unique_id(buf), "nz_comm_unique_id")
        return bytes(buf)

    def __init__(self, ctx, uid, rank, world):
        assert len(uid) == N.NZ_COMM_ID_BYTES
        self.ctx = ctx
        h = C.c_void_p()
        buf = (C.c_uint8 * N.NZ_COMM_ID_BYTES).from_buffer_copy(uid)
        N.check(N.lib.nz_comm_init(ctx._h, buf, rank, world, C.byref(h)), "nz_comm_init")
        self._h = h
        self.rank, self.world = rank, world

    def close(self):
        """nz_comm_destroy.  The library refuses while nz_sharded grids still hold the communicator (close them first):
        the status is raised and the handle KEPT, so that a later close() can still reach ncclCommDestroy."""
        if self._h:
            N.check(N.lib.nz_comm_destroy(self._h), "nz_comm_destroy")
            self._h = None

    # the TorchComm interface on top of nz_halo_exchange (one stripe per rank): lets run_pipeline drive the Python schedule
    # over the native exchange
    overlap = True

    def begin(self, planes, plan, up_rows, down_rows):
        arr = (N.dev_ptr * len(planes))(*[t.data_ptr() for t in planes])
        st = plan.stripe()
        N.check(N.lib.nz_halo_exchange_begin(self.ctx._h, self._h, arr, len(planes), C.byref(st), up_rows, down_rows, 0),
                "nz_halo_exchange_begin")
        return True

    def finish(self, reqs):
        N.check(N.lib.nz_halo_exchange_finish(self.ctx._h, self._h, None), "nz_halo_exchange_finish")

    def exchange(self, planes, plan, up_rows, down_rows):
        self.finish(self.begin(planes, plan, up_rows, down_rows))

    def allgather_range(self, map_ptr, n_floats, res_ptr, lim_min=float("inf"), lim_max=float("-inf")):
        N.check(N.lib.nz_comm_allgather_range(self.ctx._h, self._h, map_ptr, n_floats, res_ptr, lim_min, lim_max, 0, None),
                "nz_comm_allgather_range")


def terrain_params(p):
    """PipelineParams -> nz_terrain_params."""
    return N.TerrainParams(p.noiseType, p.hurst, p.startingAmplitude, p.stepdown, p.detuneRate, p.octaves, p.noiseSize,
                           p.filter, p.gaussIterations, p.flowIterations, p.normMin, p.normMax, p.erosionIterations)


class ShardedGrid:
    """nz_sharded: the stock stage list on a grows x cols grid cut into `stripes` row stripes over all ranks of `comm`
    (None: one rank, device copies instead of RCCL).  The library owns the planes and the launch plan; `run()` replays
    it (one C call per pass)."""

    def __init__(self, ctx, comm, grows, cols, p, stripes=None, overlap=0, external_source=False, as_rank=None):
        world = comm.world if comm is not None else 1
        if as_rank is not None:
            world = as_rank[1]
        self.ctx, self.comm, self.p = ctx, comm, p
        # overlap: 0 = exchanges on the compute stream itself; 1 / True = interior rows first, border rows after the wait;
        # 2 = border rows first, the exchange for the next launch travels while the interior runs
        self.desc = N.ShardedDesc(grows, cols, stripes if stripes is not None else world, HALO_MODES[p.haloMode],
                                  int(overlap), p.xpos, p.zpos, int(bool(external_source)),
                                  as_rank[0] if as_rank is not None else 0, as_rank[1] if as_rank is not None else 0)
        self.tp = terrain_params(p)
        h = C.c_void_p()
        # ctx None: a plan-only object (no planes, cannot run): the launch plan of rank as_rank[0] of as_rank[1]
        N.check(N.lib.nz_sharded_create(ctx._h if ctx is not None else None, comm._h if comm is not None else None,
                                        C.byref(self.desc), C.byref(self.tp), C.byref(h)), "nz_sharded_create")
        self._h = h
        self.local_stripes = N.lib.nz_sharded_local_stripes(h)

    def close(self):
        if self._h:
            N.lib.nz_sharded_destroy(self._h)
            self._h = None

    def stripe(self, i):
        """(nz_stripe, source plane address, result plane address) of local stripe i."""
        st, src, res = N.Stripe(), N.dev_ptr(), N.dev_ptr()
        N.check(N.lib.nz_sharded_stripe(self._h, i, C.byref(st), C.byref(src), C.byref(res)), "nz_sharded_stripe")
        return st, src.value, res.value

    def plan(self):
        """The compiled plan as tuples (op, stripe, n, a, b, own0, own1, planes) -- see include/noize_hip.h."""
        n = C.c_int32(0)
        N.check(N.lib.nz_sharded_plan(self._h, None, 0, C.byref(n)), "nz_sharded_plan")
        rec = (C.c_int32 * (8 * n.value))()
        N.check(N.lib.nz_sharded_plan(self._h, rec, n.value, C.byref(n)), "nz_sharded_plan")
        return [tuple(rec[8 * i:8 * i + 8]) for i in range(n.value)]

    def transfers(self):
        """The exchanges' transfers in the order this rank posts them: tuples (exchange, source rank, destination rank,
        floats).  On a plan-only object: the lists of rank as_rank[0] of the real as_rank[1]-rank job."""
        n = C.c_int32(0)
        N.check(N.lib.nz_sharded_transfers(self._h, None, 0, C.byref(n)), "nz_sharded_transfers")
        rec = (C.c_int32 * (4 * max(1, n.value)))()
        N.check(N.lib.nz_sharded_transfers(self._h, rec, n.value, C.byref(n)), "nz_sharded_transfers")
        return [tuple(rec[4 * i:4 * i + 4]) for i in range(n.value)]

    def run(self, marks=False, dep=0):
        """One pass (enqueue only).  marks: also return the five stage-boundary handles."""
        from .runtime import JobHandle
        out = N.handle_t(0)
        if marks:
            m = (N.handle_t * 5)()
            N.check(N.lib.nz_sharded_pipeline(self.ctx._h, self._h, m, dep, C.byref(out)), "nz_sharded_pipeline")
            return JobHandle(self.ctx, out.value), [JobHandle(self.ctx, v) for v in m]
        N.check(N.lib.nz_sharded_pipeline(self.ctx._h, self._h, None, dep, C.byref(out)), "nz_sharded_pipeline")
        return JobHandle(self.ctx, out.value)

    def traffic(self):
        n, b = C.c_int32(0), C.c_size_t(0)
        N.check(N.lib.nz_sharded_traffic(self._h, C.byref(n), C.byref(b)), "nz_sharded_traffic")
        return n.value, b.value

    def set_timing(self, on):
        N.check(N.lib.nz_sharded_set_timing(self._h, int(bool(on))), "nz_sharded_set_timing")

    def exchange_ms(self):
        ms = C.c_float(0)
        N.check(N.lib.nz_sharded_exchange_ms(self._h, C.byref(ms)), "nz_sharded_exchange_ms")
        return ms.value

    def map_range(self, res_ptr, lim_min=float("inf"), lim_max=float("-inf")):
        N.check(N.lib.nz_sharded_map_range(self.ctx._h, self._h, res_ptr, lim_min, lim_max, 0, None), "nz_sharded_map_range")

    def normalize(self, args_ptr):
        N.check(N.lib.nz_sharded_normalize(self.ctx._h, self._h, args_ptr, 0, None), "nz_sharded_normalize")

    def owned_rows(self, i):
        """Host copy of local stripe i's result rows: (first global row, array [rows, cols])."""
        import numpy as np
        st, _, res = self.stripe(i)
        n = (st.own1 - st.own0) * st.cols
        out = np.empty(n, np.float32)
        N.check(N.lib.nz_tile_download(self.ctx._h, res + st.own0 * st.cols * 4, out.ctypes.data, n, 0, None), "download")
        self.ctx.synchronize()
        return st.grow0 + st.own0, out.reshape(st.own1 - st.own0, st.cols)


class _PlanRecorder:
    """A stripe-ops object that launches nothing: it writes down what pipeline_steps asks for, in the record layout of
    nz_sharded_plan -- the Python schedule as the specification the native plan is checked against."""

    def __init__(self, lib=None):
        self.lib = lib or N.lib
        self.records = []
        self.ids = {}

    def kernel_filter_halo_rows(self, filter, iterations):
        return self.lib.nz_kernel_filter_halo_rows(filter, iterations)

    def kernel_filter_max_fused(self, filter):
        return self.lib.nz_kernel_filter_max_fused(filter)

    def erosion_max_fused(self):
        return self.lib.nz_erosion_max_fused_iterations()

    def flow_fused_max(self):
        return self.lib.nz_flow_fused_max_iterations()

    def _id(self, buf):
        return self.ids[id(buf)]

    def fractal(self, buf, plan, p):
        if plan.own1 > plan.own0:
            self.records.append((OP_NOISE, 0, 0, 0, plan.own0, plan.own1, self._id(buf) | (self._id(buf) << 8)))

    def kernel_filter(self, src, dst, plan, filter, T):
        if plan.own1 > plan.own0:
            self.records.append((OP_FILTER, T, 0, 0, plan.own0, plan.own1, self._id(src) | (self._id(dst) << 8)))

    def erosion(self, src, dst, plan, E):
        if plan.own1 > plan.own0:
            self.records.append((OP_EROSION, E, 0, 0, plan.own0, plan.own1, self._id(src) | (self._id(dst) << 8)))

    def flow_fused(self, h, S_in, S_out, dst, plan, n, first, last, normMin, normMax):
        if plan.own1 > plan.own0:
            self.records.append((OP_FLOW, n, int(first), int(last), plan.own0, plan.own1,
                                 self._id(h) | (self._id(dst) << 8) | (self._id(S_in) << 16) | (self._id(S_out) << 24)))


class _Named:
    def __init__(self, planes=None):
        self.planes = planes

    def __getitem__(self, k):
        return self.planes[k]


def python_plan(rank, world, grows, cols, p, overlap=True):
    """What pipeline_steps does for stripe `rank` of `world`, as records (op, n, a, b, own0, own1, planes) comparable with
    ShardedGrid.plan() (whose records also carry the local stripe index)."""
    rec = _PlanRecorder()
    halo = halo_rows_needed(rec, p)
    plan = StripePlan(rank, world, grows, cols, halo, neighbours_own_halo=p.haloMode != "recompute")
    A, B = _Named(), _Named()
    S0, S1 = _Named([_Named() for _ in range(FLOW_PLANES)]), _Named([_Named() for _ in range(FLOW_PLANES)])
    rec.ids = {id(A): 0, id(B): 1, id(S0): 0, id(S1): 1}
    first_plane = {id(A): 0, id(B): 1}
    for k in range(FLOW_PLANES):
        first_plane[id(S0[k])] = 2 + k
        first_plane[id(S1[k])] = 7 + k
    marks = {"noise": 0, "gauss": 1, "flow": 2, "erosion": 3, "end": 4}
    result = []
    gen = pipeline_steps(rec, plan, p, (A, B, S0, S1), result,
                         on_stage=lambda name: rec.records.append((OP_MARK, marks[name], 0, 0, 0, 0, 0)))
    try:
        req = next(gen)
        while True:
            if req == ("finish",):
                rec.records.append((OP_XFINISH, 0, 0, 0, 0, 0, 0))
                req = next(gen)
                continue
            planes, up_rows, down_rows = req
            rec.records.append((OP_XBEGIN, len(planes), up_rows, down_rows, 0, 0, first_plane[id(planes[0])]))
            if overlap:
                req = gen.send("async")
            else:
                rec.records.append((OP_XFINISH, 0, 0, 0, 0, 0, 0))
                req = next(gen)
    except StopIteration:
        pass
    return rec.records, rec.ids[id(result[0])], plan
