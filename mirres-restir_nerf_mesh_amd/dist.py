"""Multi-GPU sharding of one frame. One process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU tests).

Scheme used here: the spp range of the frame is split into contiguous slices, one per rank (disjoint frame-index ranges, so every rank
draws the samples a single GPU would have drawn for those indices); mesh/BVH/env/material are replicated. The ONLY data-path exchange is
one all-reduce(sum) of the six [N,3] accumulators after the loop, followed by the (cheap, replicated) average + EAW + composite.
Temporal reuse restarts at the first sample of a slice, i.e. the result is statistically equivalent, not bit-identical, to 1 GPU.

Exact scheme (`render_strips`, SURVEY §8e primary): the frame is cut into horizontal strips, one per rank. Everything on the path is per pixel
except the spatial reuse pass, which reads the pre-spatial reservoirs and the G-buffer of neighbours within 30 px: a rank's LOCAL frame is
its own rows plus 30 halo rows on either side (static G-buffer halo), and once per sample — between temporal and spatial reuse — the ranks
swap the border rows of their packed reservoirs (30 rows x fx x 32 B per direction: 1.5 MB at 1600 px, point to point with the two
neighbours). RNG streams are seeded with global pixel coordinates, so the rows a rank owns are bit-identical to the single-GPU frame; the six
raw sums are all-gathered by rows and the (cheap) average + EAW + composite runs replicated on the whole frame."""
import ctypes as C
import os

import torch


def _forced():
    """MIRRES_DIST_FORCE=1: run every collective even in a group of one rank (they degenerate to copies, but RCCL is loaded, the communicator is
    built and its kernels are enqueued on the render's streams) — how the RCCL code path is exercised on a single-GPU box (tests/test_gpu_rccl.py)."""
    return os.environ.get("MIRRES_DIST_FORCE", "0") == "1"


def spp_slice(spp, rank, world):
    """Contiguous [begin, end) of sample indices for `rank`; sizes differ by at most one; empty slices allowed when spp < world."""
    base, rem = divmod(int(spp), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


USED_SUMS = (1, 2, 4, 5)   # the accumulators the finish reads (diffuse / specular light, indirect diffuse / specular: renderer_restir.py:507-549); 0 and 3 (the colour totals) are averaged and dropped


def _sum_algo():
    a = os.environ.get("MIRRES_SUM_EXCHANGE", "direct")
    if a not in ("direct", "allreduce"):
        raise ValueError("MIRRES_SUM_EXCHANGE must be 'direct' or 'allreduce', not %r" % a)
    return a


def sum_over_ranks(flat, group=None, algo=None):
    """The element-wise sum of a flat fp32 bucket over the ranks, the same bits on every rank.

    algo 'direct' (default; MIRRES_SUM_EXCHANGE): xGMI is point-to-point with a link to every peer of the node, so the sum is a reduce-scatter issued as ONE
    all-to-all — rank r receives slice r of every rank's bucket over its seven links at once —, a local sum of the received slices IN RANK ORDER, and ONE
    all-gather of the reduced slices: two collectives of (N - 1) / N of the bucket each with every link busy in both, and the order of the additions is fixed
    here, not by whichever algorithm RCCL picks for an all-reduce (the same bits from run to run). algo 'allreduce': one RCCL all-reduce (rounds 1-4)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    algo = algo or _sum_algo()
    if algo == "allreduce":
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return flat
    n = flat.numel()
    if n == 0:                                                      # zero-size collectives are backend dependent: the sum of nothing is nothing
        return flat
    per = -(-n // world)
    per = -(-per // 64) * 64                                       # 256-byte slices
    if per * world != n:
        padded = torch.zeros(per * world, dtype=flat.dtype, device=flat.device); padded[:n] = flat
    else:
        padded = flat
    recv = torch.empty_like(padded)
    dist.all_to_all_single(recv, padded, group=group)              # recv[i] = slice `rank` of rank i's bucket
    parts = recv.view(world, per)
    mine = parts[0].clone()
    for i in range(1, world):
        mine += parts[i]                                            # rank order: the same additions wherever and whenever this runs
    dist.all_gather_into_tensor(padded, mine, group=group)
    return padded[:n]


def allreduce_sums(sums, group=None, occ=None, used=None):
    """Sum the six accumulators over ranks as ONE flat collective. `occ` (the replicated occupancy, [N] or [N,1]): only the FOREGROUND pixels' sums travel
    (round 5) — a background pixel's sums are zero on every rank (no stage accumulates into it), so their sum is the zero that is already there; the collective
    shrinks from 184 MB to 71 MB (icosphere, 38.5 % foreground) / 93 MB (lego-like) at 1600 x 1600 and the result has the same bits. `used` (indices, e.g. USED_SUMS):
    only these accumulators are exchanged — the others come back as this rank's partial sums (the frame's finish does not read them): 47 MB / 62 MB."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not _forced()):
        return sums
    idx = None
    if occ is not None:
        idx = torch.nonzero(occ.detach().reshape(-1) > 0.5, as_tuple=False).reshape(-1)      # the same list on every rank (the G-buffer is replicated)
        if idx.numel() == sums[0].shape[0]:
            idx = None
        elif idx.numel() == 0:
            # no foreground pixel in the view (camera looking away, an empty strip): every rank holds the same (replicated) occupancy and therefore takes this
            # branch together; the sums of background pixels are zero everywhere — nothing to exchange, and no zero-size collective is issued
            return sums
    sel = list(range(len(sums))) if used is None else [k for k in used if k < len(sums)]
    flat = torch.cat([(sums[k] if idx is None else sums[k][idx]).reshape(-1) for k in sel])
    flat = sum_over_ranks(flat, group)
    out, o = list(sums), 0
    for k in sel:
        s = sums[k]
        if idx is None:
            n = s.numel()
            out[k] = flat[o:o + n].view_as(s)
        else:
            n = idx.numel() * s.shape[1]
            r = torch.zeros_like(s)
            r[idx] = flat[o:o + n].view(idx.numel(), s.shape[1])
            out[k] = r
        o += n
    return out


def render_sharded(ctx, worker, mlp_mat, env_map, g, spp, random_offset, rank, world, denoise_iter=2, step_width=2, c_phi=2.0, n_phi=0.1, p_phi=0.001,
                   use_scale=False, scale=(1.0, 1.0, 1.0), group=None):
    """Renders this rank's spp slice with the fused loop, all-reduces the raw sums and finishes (average, EAW, composite) on every rank.
    `g` is a G-buffer dict (harness.build_gbuffer). Returns the six [N,3] outputs of run_restir_di_with_pt."""
    from . import _lib
    from ._lib import lib, check, stream_ptr
    from .renderer_restir import render_fused
    b, e = spp_slice(spp, rank, world)
    occ = g["occ"].clone()
    if world == 1 and not _forced():
        outs, _, _ = render_fused(ctx, worker, mlp_mat, use_scale, scale, env_map, occ, g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp,
                                  denoise_iter, step_width, c_phi, n_phi, p_phi, random_offset)
        return outs
    # an empty slice (more ranks than samples) has begin == end > 0: the loop body never runs and the sums stay zero
    sums, a, keep = render_fused(ctx, worker, mlp_mat, use_scale, scale, env_map, occ, g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp,
                                 denoise_iter, step_width, c_phi, n_phi, p_phi, random_offset, spp_range=(b, e))
    sums = allreduce_sums(sums, group, occ=g["occ"], used=USED_SUMS)
    outs = [torch.empty_like(s) for s in sums]
    for k in range(6):
        a.outs[k] = outs[k].data_ptr()
    arr = (C.c_void_p * 6)(*[s.data_ptr() for s in sums])
    check(lib().mirres_render_finish(ctx.h, C.byref(a), arr, stream_ptr()), "mirres_render_finish")
    return outs


# ------------------------------------------------------------------------------------------------ exact strip sharding
_WARNED_OVERLAP = False
HALO_ROWS = 30   # = mirres_config_t.gather_radius (SpatialResampling.slang:33-39)
STRIP_ROW_QUANTUM = 32   # local strip frames are padded (background rows) to a multiple of this many rows: see render_strips


BG_WEIGHT = 0.03   # cost of a background pixel relative to a foreground one (measured, profiles/r05_strip_table.txt: T_strip = a + 1.7 .. 2.6 ns x foreground px + ~0 x background px)


def row_cost(fy, fx, occ, bg_weight=BG_WEIGHT):
    """Static cost model of the rows of a frame: a foreground pixel costs 1, a background pixel `bg_weight`. float64 [fy] on the CPU."""
    o = occ.detach().reshape(int(fy), int(fx)).float()
    return (bg_weight * o.shape[1] + (1.0 - bg_weight) * (o > 0.5).float().sum(dim=1)).double().cpu()


def strip_bounds(fy, world, occ=None, fx=None, halo=HALO_ROWS, bg_weight=BG_WEIGHT, cost=None):
    """Row boundaries [b_0 = 0, ..., b_world = fy] of the strips. Without `occ` / `cost`: equal heights (differing by at most one row). With the full-frame
    occupancy `occ` ([fy*fx] or [fy*fx,1], any device): cost-balanced by the static model (row_cost) — boundaries at equal cost quantiles, every strip at least
    `halo` rows high. `cost` ([fy], e.g. StripBalancer.cost) replaces the model. Deterministic in its inputs, so every rank computes the same partition from the
    (replicated) G-buffer."""
    fy, world = int(fy), int(world)
    if world > 1 and fy // world < halo:
        raise ValueError("strip sharding needs at least %d rows per rank (fy=%d, world=%d)" % (halo, fy, world))
    if (occ is None and cost is None) or world == 1:
        base, rem = divmod(fy, world)
        b = [0]
        for r in range(world):
            b.append(b[-1] + base + (1 if r < rem else 0))
        return b
    if cost is None:
        cost = row_cost(fy, fx, occ, bg_weight)
    cum = torch.cumsum(cost.double().cpu(), 0)
    total = float(cum[-1])
    b = [0]
    for r in range(1, world):
        y = int(torch.searchsorted(cum, torch.tensor(total * r / world, dtype=cum.dtype)).item()) + 1
        y = max(y, b[-1] + halo)                       # at least `halo` rows per strip ...
        y = min(y, fy - halo * (world - r))            # ... including the ones still to come
        b.append(y)
    b.append(fy)
    return b


class StripBalancer:
    """Measured load balancing of the exact strip scheme over successive frames (round 5). The static model (foreground pixel = 1) leaves the slowest of eight
    strips 15-25 % above the mean (profiles/r05_strip_table.txt): what a foreground pixel costs depends on what its rays meet. The strips' own render times say
    where: every frame each rank times its strip (stream events, read at the start of the next frame: no extra synchronisation), the times are all-gathered (one
    tiny collective per frame), and a per-row correction of the static model is scaled by (t_r / mean)^damping over strip r's rows. The next frame's boundaries are
    the cost quantiles of model x correction, computed identically on every rank. Boundaries never change a pixel (the scheme is exact for any partition), so this
    only moves time. The correction is tied to rows of the image, not to a view: it carries over to a similar next view and washes out (towards 1) otherwise."""

    def __init__(self, fy, world, damping=1.0, bg_weight=BG_WEIGHT, smooth=0.02):
        self.fy, self.world, self.damping, self.bg_weight, self.smooth = int(fy), int(world), float(damping), float(bg_weight), float(smooth)
        self.corr = torch.ones(self.fy, dtype=torch.float64)
        self.last_bounds = None
        self.pending = None        # (start event, end event) of this rank's last strip render
        self.waits = []            # (event before, event after) of the sampled halo exchanges of that render, + how many exchanges each stands for
        self.wait_stride = 1
        self.native = None         # (context, stride) of a render whose exchanges the library issued and timed itself
        self.last_total_ms = self.last_wait_ms = None
        self.last_path, self.last_spp = None, None     # how the last strip render exchanged its halos ("native": csrc/comm.hip, "callback": torch.distributed from the host callback)
        self.history = []          # (bounds, times) per update: what the table in profiles/ is printed from

    def cost(self, fx, occ):
        return row_cost(self.fy, fx, occ, self.bg_weight) * self.corr

    def bounds(self, fx, occ):
        self.last_bounds = strip_bounds(self.fy, self.world, cost=self.cost(fx, occ))
        return self.last_bounds

    def update(self, times):
        """`times`: this frame's strip render times of ALL ranks (any unit), in rank order; the same list on every rank."""
        t = torch.tensor([float(x) for x in times], dtype=torch.float64)
        if self.last_bounds is None or len(t) != self.world or not bool((t > 0).all()):
            return
        self.history.append((list(self.last_bounds), [float(x) for x in t]))
        rel = (t / t.mean()) ** self.damping
        for r in range(self.world):
            self.corr[self.last_bounds[r]:self.last_bounds[r + 1]] *= rel[r]
        self.corr /= self.corr.mean()
        if self.smooth > 0:            # relax towards 1: a correction learnt on one view fades on views that do not confirm it
            self.corr = self.corr ** (1.0 - self.smooth)

    # ---- timing of this rank's strip and the exchange of the times
    # What is exchanged is the strip's OWN busy time, not the wall time of its render: in a multi-rank run every sample's halo exchange waits for the
    # neighbours, so all ranks advance at the slowest strip's pace and their wall times are near-equal whatever the partition — a balancer fed with them never moves.
    # The time the stream spends inside the exchanges (event pairs around a sample of them, on the stream the exchange is enqueued on: with RCCL the send / recv
    # kernels sit there until the peer arrives, with a host-staged backend the stream idles while the host blocks) is subtracted.
    MAX_WAIT_SAMPLES = 32      # event pairs per frame: every (spp / 32)-th exchange is bracketed and stands for its stride (512 pairs would cost milliseconds of host time)

    def start(self, spp=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.pending = (e0, e1)
        self.waits = []
        self.wait_stride = max(1, int(spp or 1) // self.MAX_WAIT_SAMPLES)

    def bracket(self, sample):
        """Context manager for the halo callback: brackets this sample's exchange with events on the CURRENT stream when it is one of the sampled ones."""
        import contextlib
        if self.pending is None or int(sample) % self.wait_stride != 0:
            return contextlib.nullcontext()
        bal = self

        class _B:
            def __enter__(self_):
                self_.a = torch.cuda.Event(enable_timing=True); self_.b = torch.cuda.Event(enable_timing=True)
                self_.a.record(torch.cuda.current_stream())
            def __exit__(self_, *exc):
                self_.b.record(torch.cuda.current_stream())
                bal.waits.append((self_.a, self_.b))
                return False
        return _B()

    def stop(self):
        if self.pending is not None:
            self.pending[1].record()

    def abort(self):
        """The render between start() and stop() raised: forget it (an unrecorded end event would fail the next exchange() on this rank only and mismatch the collective)."""
        self.pending = None
        self.waits = []
        self.native = None

    def busy_ms(self):
        """(own busy time, total, waited) of the last timed strip render in ms; None when nothing is pending."""
        if self.pending is None:
            return None
        e0, e1 = self.pending
        e1.synchronize()
        total = e0.elapsed_time(e1)
        waited = sum(a.elapsed_time(b) for a, b in self.waits) * self.wait_stride
        if self.native is not None:       # the exchanges were issued by the library (no callback to bracket): it timed every stride-th of them itself
            from ._lib import lib, check
            ctx, stride = self.native
            ms, n = C.c_double(0.0), C.c_int(0)
            check(lib().mirres_ctx_halo_time(ctx.h, C.byref(ms), C.byref(n)), "mirres_ctx_halo_time")
            waited += float(ms.value) * max(1, stride)
            self.native = None
        waited = min(waited, 0.95 * total)          # (a sampled estimate: never let it eat the whole frame)
        return total - waited, total, waited

    def exchange(self, rank, group=None, measured=None):
        """Called at the start of a frame: last frame's own busy time of every rank -> update(). Returns the gathered times (None on the first frame).
        `measured` = (busy, total, waited) replaces the event timing (tests)."""
        m = measured if measured is not None else self.busy_ms()
        if m is None:
            return None
        mine, self.last_total_ms, self.last_wait_ms = m
        self.pending = None
        self.waits = []
        times = gather_times(mine, rank, self.world, group)
        self.update(times)
        return times


def gather_times(mine, rank, world, group=None):
    """One float per rank -> the list of all of them, in rank order, on every rank (the balancer's only collective: 8 bytes per rank and frame)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [float(mine)]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    buf = torch.zeros(int(world), dtype=torch.float64, device=dev)
    buf[rank] = float(mine)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return [float(x) for x in buf.cpu()]


def strip_rows(fy, rank, world, halo=HALO_ROWS, bounds=None):
    """Rows of `rank`'s strip of an fy-row frame: (y0, y1, lo, hi) = own rows [y0, y1) and local frame rows [lo, hi) (own + halo, clipped).
    Every strip must be at least `halo` rows high so that a halo never reaches beyond the adjacent rank. `bounds` = strip_bounds(...)."""
    b = bounds if bounds is not None else strip_bounds(fy, world, halo=halo)
    y0, y1 = int(b[rank]), int(b[rank + 1])
    return y0, y1, max(0, y0 - halo), min(int(fy), y1 + halo)


class _DevMem:
    """Aliases raw device memory as a torch tensor (through __cuda_array_interface__), for the buffers libmirres hands to callbacks."""
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def device_view(ptr, shape):
    return torch.as_tensor(_DevMem(ptr, shape), device="cuda")


def on_stream(stream):
    """Context manager that makes the raw HIP stream `stream` (an integer handle, as mirres_render's halo callback receives it) torch's current stream, so
    that copies and collectives enqueued inside are ordered on it. With strip_overlap the callback gets the engine's side stream, not the caller's."""
    if not stream or int(stream) == torch.cuda.current_stream().cuda_stream:
        import contextlib
        return contextlib.nullcontext()
    return torch.cuda.stream(torch.cuda.ExternalStream(int(stream)))


def halo_plan(fy, fx, rank, world, halo=HALO_ROWS, bounds=None):
    """What the per-sample exchange moves, in LOCAL row numbers of `rank`'s frame: a list of (peer, send_rows, recv_rows) with row ranges
    [a, b). The rank above receives our first own rows (its bottom halo) and sends its last own rows (our top halo); same below."""
    y0, y1, lo, hi = strip_rows(fy, rank, world, halo, bounds)
    plan = []
    if rank > 0:
        top = y0 - lo                                             # our top halo = the upper neighbour's last `top` own rows
        py0, py1, plo, phi = strip_rows(fy, rank - 1, world, halo, bounds)
        plan.append((rank - 1, (y0 - lo, y0 - lo + (phi - py1)), (0, top)))
    if rank < world - 1:
        bot = hi - y1
        py0, py1, plo, phi = strip_rows(fy, rank + 1, world, halo, bounds)
        plan.append((rank + 1, (y1 - lo - (py0 - plo), y1 - lo), (y1 - lo, y1 - lo + bot)))
    return plan


def exchange_halos(records, plan, group=None):
    """One exchange step on `records` ([local rows, fx, 8] packed reservoirs, a view of the engine's buffer). NCCL/RCCL: batched
    point-to-point ops enqueued on the current stream. gloo (CPU tests, or GPU tensors staged through the host): blocking."""
    import torch.distributed as dist
    if not plan:
        return
    if dist.get_backend(group) == "nccl":
        ops, keep = [], []
        for peer, (sa, sb), (ra, rb) in plan:
            ops.append(dist.P2POp(dist.isend, records[sa:sb], peer, group))
            ops.append(dist.P2POp(dist.irecv, records[ra:rb], peer, group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return
    reqs, staged = [], []
    for peer, (sa, sb), (ra, rb) in plan:
        out = records[sa:sb].detach().to("cpu").contiguous()
        buf = torch.empty((rb - ra,) + tuple(records.shape[1:]), dtype=records.dtype)
        reqs.append(dist.isend(out, peer, group)); reqs.append(dist.irecv(buf, peer, group))
        staged.append((ra, rb, buf, out))
    for r in reqs:
        r.wait()
    for ra, rb, buf, _ in staged:
        records[ra:rb].copy_(buf)


def gather_rows(own, fy, fx, world, group=None, bounds=None, used=None):
    """All-gather of row strips: `own` = list of [own rows * fx, C] tensors of this rank (same C, dtype) -> list of [fy * fx, C] tensors on every rank.
    ONE collective for all buffers (round 5; rounds 1-4: a list all_gather + cat per buffer, two extra copies of each): the rank's buffers are stacked into one
    [K, tallest strip * fx, C] block (strips differ in height: padded at the end), `all_gather_into_tensor` fills a [world, K, ...] block, and every output is
    assembled from views of it with one `cat`. `used` (indices, e.g. USED_SUMS): only these buffers travel; the others — which the finish
    does not read — come back as full-size zero frames."""
    import torch.distributed as dist
    b = bounds if bounds is not None else strip_bounds(fy, world)
    rows = [int(b[r + 1] - b[r]) for r in range(world)]
    sel = list(range(len(own))) if used is None else [k for k in used if k < len(own)]
    mx, K, c = max(rows), len(sel), own[0].shape[1]
    send = torch.zeros((K, mx * fx, c), dtype=own[0].dtype, device=own[0].device)
    for j, k in enumerate(sel):
        send[j, :own[k].shape[0]] = own[k]
    recv = torch.empty((world * K,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)      # concatenation along dim 0: rank r's block = rows [r K, (r + 1) K)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view((world,) + tuple(send.shape))
    full = [None] * len(own)
    for j, k in enumerate(sel):
        full[k] = torch.cat([recv[r, j, :rows[r] * fx] for r in range(world)], dim=0)
    for k in range(len(own)):
        if full[k] is None:
            full[k] = torch.zeros((fy * fx, c), dtype=own[0].dtype, device=own[0].device)
    return full


# ---- the library's own RCCL communicator for the native halo exchange (csrc/comm.hip): one per process group, created on first use
_NATIVE_COMMS = {}


def rccl_path():
    """The librccl the process already holds (torch's); mirres_comm_* prefers a loaded copy over loading a second one."""
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else ""


def native_comm(group=None):
    """Communicator handle (int) of mirres_comm_create for `group`: rank 0 draws the unique id, torch.distributed broadcasts its 128 bytes, every rank joins
    (collective: all ranks of the group must call this together — render_strips does on its first frame). MIRRES_NATIVE_HALO=0 or a non-RCCL backend: None."""
    import torch.distributed as dist
    from ._lib import lib, check
    if os.environ.get("MIRRES_NATIVE_HALO", "1") == "0" or dist.get_backend(group) != "nccl":
        return None
    key = id(group) if group is not None else 0
    if key in _NATIVE_COMMS:
        return _NATIVE_COMMS[key]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    path = rccl_path().encode()
    ident = (C.c_char * 128)()
    if rank == 0:
        check(lib().mirres_comm_unique_id(path, C.cast(ident, C.c_void_p)), "mirres_comm_unique_id")
    t = torch.frombuffer(bytearray(bytes(ident)), dtype=torch.uint8).clone().cuda()
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    raw = bytes(t.cpu().numpy().tobytes())
    buf = (C.c_char * 128).from_buffer_copy(raw)
    h = C.c_void_p()
    torch.cuda.synchronize()
    check(lib().mirres_comm_create(C.byref(h), path, C.cast(buf, C.c_void_p), world, rank), "mirres_comm_create")
    _NATIVE_COMMS[key] = h.value
    return h.value


def render_strips(ctx_full, worker, mlp_mat, env_map, g, spp, random_offset, rank, world, denoise_iter=2, step_width=2, c_phi=2.0, n_phi=0.1, p_phi=0.001,
                  use_scale=False, scale=(1.0, 1.0, 1.0), group=None, max_bounce=None, balanced=True, overlap=None, balancer=None):
    """Exact multi-GPU frame: this rank renders its strip (all spp) with per-sample halo exchange, the raw sums are all-gathered by rows and
    finished on every rank. `g` is the full-frame G-buffer dict (harness.build_gbuffer); `ctx_full` a context of the full frame (finish only)."""
    from . import _lib
    from ._lib import lib, check, stream_ptr
    from ._ops import get_ctx
    from .renderer_restir import render_fused
    fx, fy = int(g["fx"]), int(g["fy"])
    if world == 1 and not _forced():
        outs, _, _ = render_fused(ctx_full, worker, mlp_mat, use_scale, scale, env_map, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"],
                                  g["pos"], spp, denoise_iter, step_width, c_phi, n_phi, p_phi, random_offset)
        return outs
    if balancer is not None:      # measured balancing: last frame's strip times of all ranks move this frame's boundaries (StripBalancer)
        balancer.exchange(rank, group)
        bounds = balancer.bounds(fx, g["occ"])
    else:
        bounds = strip_bounds(fy, world, g["occ"] if balanced else None, fx)      # cost-balanced strip heights by the static model (same on every rank)
    y0, y1, lo, hi = strip_rows(fy, rank, world, bounds=bounds)
    sl = slice(lo * fx, hi * fx)
    # Balanced strip heights follow the view's occupancy, so every camera would ask for a context (and a multi-GB batch pool) of its own height.
    # The local frame is therefore padded at the bottom with background rows (occ = 0: no stage works on them, no neighbour accepts them) up to a
    # multiple of STRIP_ROW_QUANTUM rows: a multi-view run cycles through a handful of context sizes that the bounded cache (_ops.get_ctx) holds.
    rows = hi - lo
    rows_pad = -(-rows // STRIP_ROW_QUANTUM) * STRIP_ROW_QUANTUM

    def _local(t):
        t = t[sl]
        if rows_pad == rows:
            return t.contiguous()
        out = torch.zeros((rows_pad * fx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        out[:rows * fx] = t
        return out
    loc = {k: _local(g[k]) for k in ("occ", "normal", "depth", "kd", "rm", "ray_dir", "pos")}
    ctx_loc = get_ctx(fx, rows_pad, max_bounce)
    plan = halo_plan(fy, fx, rank, world, bounds=bounds)

    if overlap is None:      # MIRRES_STRIP_OVERLAP=1: exchange on a side stream behind the interior rows' spatial pass (mirres_render_args_t.strip_overlap)
        overlap = os.environ.get("MIRRES_STRIP_OVERLAP", "0") == "1"
    if overlap and world > 1:
        import torch.distributed as dist
        if dist.get_backend(group) != "nccl":
            # the exchange of any other backend is staged through the host and blocks (exchange_halos): nothing overlaps, and the split costs three more launches per
            # sample and the fused temporal merge — run the in-line exchange instead and say so once
            global _WARNED_OVERLAP
            if not _WARNED_OVERLAP:
                import warnings
                warnings.warn("strip_overlap needs stream-ordered point-to-point transfers (backend nccl = RCCL); backend %r exchanges through the host: overlap ignored" % dist.get_backend(group))
                _WARNED_OVERLAP = True
            overlap = False

    import contextlib

    def _halo(user, records, sample, stream):
        try:
            with on_stream(stream):
                with (balancer.bracket(sample) if balancer is not None else contextlib.nullcontext()):
                    exchange_halos(device_view(records, (rows_pad, fx, 8)), plan, group)
            return 0
        except Exception as e:      # surfaced by mirres_render as MIRRES_E_STATE
            import sys
            print("[mirres] halo exchange failed:", e, file=sys.stderr)
            return 1
    cb = _lib.HALO_FN(_halo)
    # over RCCL the library issues the exchange itself (csrc/comm.hip: one send / receive group per sample from C, no Python on the per-sample path); the callback stays
    # for host-staged backends (gloo: the CPU / single-GPU tests) and MIRRES_NATIVE_HALO=0
    native = None
    if world > 1 or _forced():
        import torch.distributed as dist
        comm = native_comm(group) if (dist.is_available() and dist.is_initialized()) else None
        if comm is not None:
            stride = max(1, int(spp) // StripBalancer.MAX_WAIT_SAMPLES) if balancer is not None else 0
            native = (comm, plan, stride)
            if balancer is not None:
                balancer.native = (ctx_loc, stride)
    if balancer is not None:
        balancer.last_path, balancer.last_spp = ("native" if native is not None else "callback"), int(spp)
        balancer.start(spp)
    try:
        sums, a, keep = render_fused(ctx_loc, worker, mlp_mat, use_scale, scale, env_map, loc["occ"], loc["normal"], loc["depth"], loc["kd"], loc["rm"], loc["ray_dir"],
                                     loc["pos"], spp, denoise_iter, step_width, c_phi, n_phi, p_phi, random_offset, strip=(fy, lo, y0 - lo, y1 - lo), halo=cb, strip_overlap=overlap,
                                     halo_native=native)
    except BaseException:
        if balancer is not None:
            balancer.abort()
        raise
    if balancer is not None:
        balancer.stop()
    own = [s_[(y0 - lo) * fx:(y1 - lo) * fx].contiguous() for s_ in sums]
    full = gather_rows(own, fy, fx, world, group, bounds, used=USED_SUMS)
    # replicated finish on the whole frame (average, EAW, composite)
    _, af, keepf = _finish_args(ctx_full, env_map, g, spp, denoise_iter, step_width, c_phi, n_phi, p_phi)
    outs = [torch.empty_like(s_) for s_ in full]
    for k in range(6):
        af.outs[k] = outs[k].data_ptr()
    arr = (C.c_void_p * 6)(*[s_.data_ptr() for s_ in full])
    check(lib().mirres_render_finish(ctx_full.h, C.byref(af), arr, stream_ptr()), "mirres_render_finish")
    return outs


def _finish_args(ctx, env_map, g, spp, denoise_iter, step_width, c_phi, n_phi, p_phi):
    """RenderArgs for mirres_render_finish on the full frame (only the G-buffer, spp and the denoiser parameters are read)."""
    from . import _lib
    a = _lib.RenderArgs()
    keep = []
    a.spp = int(spp)
    env = env_map.detach().contiguous().float(); keep.append(env)
    a.env_map, a.Hc, a.Wc = env.data_ptr(), env.shape[0], env.shape[1]
    occ = g["occ"].clone(); keep.append(occ)
    occ[occ <= 0.5] = 0          # what mirres_render leaves in occ (renderer_restir.py:484-485)
    a.occ = occ.data_ptr()
    for name, key in (("normal", "normal"), ("depth", "depth"), ("kd", "kd"), ("rough_metal", "rm"), ("ray_dir", "ray_dir"), ("pos", "pos")):
        t = g[key].detach().contiguous().float(); keep.append(t); setattr(a, name, t.data_ptr())
    a.denoise_iter, a.step_width, a.c_phi, a.n_phi, a.p_phi = int(denoise_iter), int(step_width), float(c_phi), float(n_phi), float(p_phi)
    return None, a, keep


# ------------------------------------------------------------------------------------------------ training: gradient exchange
def allreduce_gradients(tensors, group=None, average=True, algo=None):
    """Stage-1 data parallelism (BASELINE configs[4]: 'grads all-reduced over xGMI'): every rank renders its own views / strips and
    the parameter gradients — hash grid (50 MB fp32), MLP, environment map, vertex offsets — are summed as ONE flat bucket by sum_over_ranks (direct
    exchange over the point-to-point links by default; algo / MIRRES_SUM_EXCHANGE = 'allreduce' for the flat RCCL all-reduce). `tensors` = parameters
    (their .grad is used; missing grads count as zero) or plain gradient tensors; updated in place."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not _forced()):
        return
    grads = []
    for t in tensors:
        if t.requires_grad:         # a parameter or a plain trainable leaf (the reference's `light_base`: EnvironmentLight is not an nn.Module)
            if t.grad is None:      # e.g. a strip rank that saw only background: its contribution is zero, and it must still take part
                t.grad = torch.zeros_like(t)
            g = t.grad
        else:
            g = t                   # already a gradient tensor
        grads.append(g)
    if not grads:
        return
    with torch.no_grad():
        flat = sum_over_ranks(torch.cat([g.reshape(-1).to(torch.float32) for g in grads]), group, algo)
        if average:
            flat = flat / dist.get_world_size(group)
        o = 0
        for g in grads:
            k = g.numel()
            g.copy_(flat[o:o + k].view_as(g)); o += k
