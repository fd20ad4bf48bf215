"""Multi-GPU sharding of the grid evaluation: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests).

The reference has no multi-device code.  What shards is the list of evaluation points of one grid /
octree level, and - because the MLP's GroupNorm couples all points of one `query()` call (SURVEY C-1) -
it shards at CHUNK granularity: rank r evaluates a contiguous range of whole chunks of the C-ordered
list, then ONE all-gather of the fp32 predictions rebuilds the full list on every rank (the only
exchange step of the path; everything else - octree masks, fills, marching cubes - is recomputed or
consumed redundantly from the replicated field).  Contiguous chunk ranges of a dense grid are slabs
along array axis 0 (the "Z-slab" of the north star).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch


# Self-test switch: with a process group of ONE rank the sharded entry points normally skip their collectives; setting
# this makes them run the full multi-rank code path (padding, all-gather, re-assembly) anyway.  It exists so that the RCCL
# transport can be exercised end to end on a single-GPU box (tests/test_gpu_dist.py, bench.py --dist-selftest).
ALWAYS_COLLECTIVE = False


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collective_active():
    """True when the sharded entry points exchange data: more than one rank, or the single-rank self-test."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or ALWAYS_COLLECTIVE


# ---- the collective itself ---------------------------------------------------------------------------------------------------
# Default: torch.distributed's all_gather_into_tensor (backend nccl = RCCL on ROCm, gloo in the CPU tests).
# `use_native_comm()` switches the device-tensor all-gathers to the library's own RCCL communicator (`pifu_comm_*`,
# `pifu_allgather_*` of include/pifuhd_hip.h) - the same collective as a binding without torch.distributed would issue it;
# torch.distributed is then only the side channel that hands rank 0's communicator id to the other ranks.
_native = None


def use_native_comm(device):
    """Create the library's RCCL communicator over the ranks of the initialised DEFAULT process group (collective) and route
    the device all-gathers of this module through it.  Returns the communicator handle.  `device` is resolved to an indexed
    device once ('cuda' -> the current device), so that tensors - which always carry an index - compare equal to it."""
    import torch.distributed as dist
    from . import _lib
    global _native
    rank, ws = world()
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise _lib.PifuError(_lib.ERR_INVALID, 'use_native_comm: RCCL needs a GPU device')
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    dev = torch.device('cuda', idx)
    ident = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        _lib.call('pifu_comm_unique_id', buf)
        ident = torch.tensor(list(buf), dtype=torch.uint8)
    if ws > 1:
        carrier = ident.to(dev) if dist.get_backend() == 'nccl' else ident
        dist.broadcast(carrier, src=0)
        ident = carrier.cpu()
    raw = (C.c_uint8 * _lib.COMM_ID_BYTES)(*ident.tolist())
    handle = C.c_void_p()
    _lib.call('pifu_comm_create', idx, rank, ws, raw, C.byref(handle))
    _native = {'h': handle, 'device': dev, 'calls': 0, 'bytes': 0}
    return handle


def native_comm_stats():
    """(calls, gathered bytes) that went through the library's own communicator since `use_native_comm` (None when it is
    not active): what a test reads to prove that the C exchange - not torch.distributed - carried the data."""
    return None if _native is None else (_native['calls'], _native['bytes'])


def release_native_comm():
    from . import _lib
    global _native
    if _native is not None:
        _lib.call('pifu_comm_destroy', _native['h'])
        _native = None


def _is_default_group(group):
    import torch.distributed as dist
    return group is None or group is dist.group.WORLD


def _all_gather(full, local, group=None):
    """full [ws * n, ...] <- every rank's local [n, ...] (contiguous tensors of one dtype).  The library's communicator
    spans the default group only: a call on a sub-group always goes through torch.distributed."""
    import torch.distributed as dist
    if _native is not None and full.is_cuda and full.device == _native['device'] and _is_default_group(group):
        from . import _lib
        from .nets import _ptr, _stream_ptr
        assert full.is_contiguous() and local.is_contiguous() and full.numel() == world()[1] * local.numel()
        if full.dtype == torch.float32:
            _lib.call('pifu_allgather_f32', _native['h'], _ptr(local), _ptr(full), local.numel(), _stream_ptr(full.device))
        else:
            _lib.call('pifu_allgather_bytes', _native['h'], _ptr(local), _ptr(full), local.numel() * local.element_size(),
                      _stream_ptr(full.device))
        _native['calls'] += 1
        _native['bytes'] += full.numel() * full.element_size()
        return
    dist.all_gather_into_tensor(full, local, group=group)


def shard_chunks(total_points: int, chunk: int, world_size: int, rank: int):
    """Contiguous range of whole chunks for `rank`: returns (first_point, n_points, chunks_per_rank).
    Every rank gets `chunks_per_rank` chunks except the tail (possibly fewer or none)."""
    n_chunks = (total_points + chunk - 1) // chunk
    per = (n_chunks + world_size - 1) // world_size
    c0 = min(rank * per, n_chunks)
    c1 = min(c0 + per, n_chunks)
    first = c0 * chunk
    last = min(c1 * chunk, total_points)
    return first, max(last - first, 0), per


def sharded_eval(total_points: int, chunk: int, eval_fn, device, group=None):
    """Evaluate `total_points` list entries in chunk-granular shards and all-gather the predictions.

    eval_fn(first, count) -> float32 tensor [count] on `device` (predictions of list entries
    first .. first+count, chunks cut relative to `first`).  Returns the float32 tensor [total_points],
    identical on every rank.  `chunk` is the granularity of the deal: the reference's query() chunk when the MLP's norm
    couples it, the library's own grouping (`effective_chunk`) when every norm is pointwise."""
    import torch.distributed as dist
    rank, ws = world()
    if not collective_active():
        return eval_fn(0, total_points)
    first, count, per = shard_chunks(total_points, chunk, ws, rank)
    slot = per * chunk
    local = torch.zeros(slot, device=device, dtype=torch.float32)
    if count > 0:
        local[:count] = eval_fn(first, count)
    full = torch.empty(ws * slot, device=device, dtype=torch.float32)
    _timed(lambda: _all_gather(full, local, group), full.numel() * 4, device)
    return full[:total_points]


def effective_chunk(ctx, chunk, res2=0, flags=0):
    """The granularity at which a lattice evaluation may be dealt out without changing a single bit (`pifu_effective_chunk`):
    the reference's query() chunk while a norm of the MLPs couples the points of a call (GroupNorm, train-mode BatchNorm);
    with pointwise norms - the published pifuhd.pt has none - the library's own grouping: 16 384 rows, for a dense lattice
    whole columns, i.e. raw Z-slabs of the voxel grid with no padding to 5 000-point chunks."""
    from . import _lib
    return int(_lib.load().pifu_effective_chunk(ctx.h, int(chunk), int(res2), int(flags)))


def eval_lattice_sharded(net, calib_tensor, res, chunk, flags=0, group=None):
    """Dense lattice res[0] x res[1] x res[2] evaluated across all ranks; returns fp32 [prod(res)]."""
    from . import _lib
    from .nets import _ptr, _stream_ptr
    from .mesh_util import _ctx_of
    ctx = _ctx_of(net)
    dev = ctx.device
    from .mesh_util import calib_pair
    calib, calib_inv = calib_pair(calib_tensor)
    res_a = (C.c_int32 * 3)(*[int(r) for r in res])
    total = int(res[0]) * int(res[1]) * int(res[2])

    def eval_fn(first, count):
        out = torch.empty(count, device=dev, dtype=torch.float32)
        _lib.call('pifu_eval_lattice', ctx.h, res_a, calib_inv.ctypes.data_as(C.c_void_p),
                  calib.ctypes.data_as(C.c_void_p), None, first, count, int(chunk), flags, _ptr(out), _stream_ptr(dev))
        return out
    return sharded_eval(total, effective_chunk(ctx, chunk, int(res[2]), flags), eval_fn, dev, group)


def eval_grid_octree_sharded(net, calib_tensor, R, chunk, init_resolution=64, threshold=0.05, flags=0, group=None):
    """`eval_grid_octree` (`mesh_util.py:124-187`) with every level's network evaluations sharded by
    whole chunks across ranks and all-gathered; the bookkeeping kernels run redundantly on every rank, so
    all ranks end with the identical float64 SDF.  Returns (sdf [R,R,R] float64 device tensor, evaluated)."""
    from . import _lib
    from .nets import _ptr, _stream_ptr
    from .mesh_util import _ctx_of
    ctx = _ctx_of(net)
    dev = ctx.device
    st = _stream_ptr(dev)
    from .mesh_util import calib_pair
    calib, calib_inv = calib_pair(calib_tensor)
    res_a = (C.c_int32 * 3)(R, R, R)
    sdf = torch.empty(R, R, R, device=dev, dtype=torch.float64)
    notproc = torch.empty(R, R, R, device=dev, dtype=torch.uint8)
    _lib.call('pifu_octree_begin', ctx.h, R, _ptr(sdf), _ptr(notproc), st)
    reso = R // init_resolution
    evaluated = 0
    while reso > 0:
        cnt = C.c_int64(0)
        replicated(lambda: _lib.call('pifu_octree_select', ctx.h, R, reso, _ptr(notproc), None, 0, C.byref(cnt), st), dev)
        n = cnt.value
        idx = torch.empty(max(n, 1), device=dev, dtype=torch.int64)
        vals = None
        if n:
            replicated(lambda: _lib.call('pifu_octree_select', ctx.h, R, reso, _ptr(notproc), _ptr(idx), n, C.byref(cnt), st), dev)

            def eval_fn(first, count):
                out = torch.empty(count, device=dev, dtype=torch.float32)
                _lib.call('pifu_eval_lattice', ctx.h, res_a, calib_inv.ctypes.data_as(C.c_void_p),
                          calib.ctypes.data_as(C.c_void_p), _ptr(idx), first, count, int(chunk), flags, _ptr(out), st)
                return out
            vals = sharded_eval(n, effective_chunk(ctx, chunk, 0, flags), eval_fn, dev, group).contiguous()
            evaluated += n
        replicated(lambda: _lib.call('pifu_octree_commit', ctx.h, R, reso, float(threshold), _ptr(idx), _ptr(vals), n, _ptr(sdf),
                                     _ptr(notproc), st), dev)
        if reso <= 1:
            break
        reso //= 2
    return sdf, evaluated


def colour_rounds(n_spans, ws, per_call_cap=10):
    """The colour loop of `gen_mesh` (`reconstruction.py:58-70`) as rounds: [(first_span, spans_per_rank)].  In a round,
    rank r computes the `spans_per_rank` consecutive calc_normal calls that start at `first_span + r * spans_per_rank`, so
    the ranks' results laid side by side ARE the colour rows of spans [first_span, first_span + ws * spans_per_rank) in
    vertex order (every span but the very last has exactly `interval` rows): one all-gather per round extends the PREFIX
    of finished colours, which is what the OBJ writer's `ready(upto)` takes.  Every rank ends with ceil(n_spans / ws)
    calls or fewer.  A round is at most `per_call_cap` calls per rank (ten 50 000-vertex calls = 2.0 M point rows = one
    batch of launches of the workspace) and the rounds TAPER - each takes half of what is left, down to a single call per
    rank - because what remains to be done once the GPU has finished is the last round's exchange, host copy and vertex
    lines: [10, 10, ..., 3, 2, 1] calls per rank instead of a last round as large as the first."""
    left = -(-n_spans // ws)
    out, base = [], 0
    while left > 0 and base < n_spans:
        m = min(per_call_cap, max(1, -(-left // 2)))
        out.append((base, m))
        base += ws * m
        left -= m
    return out


def round_rows(spans, base, m, ws, rank, interval):
    """One round of `colour_rounds`: (this rank's vertex rows (left, right) or None when it has no call in the round, the
    round's vertex rows (L, R) over all ranks, rows per rank slot of the exchange buffer)."""
    lo_s, hi_s = base + rank * m, min(base + (rank + 1) * m, len(spans))
    mine = (spans[lo_s][0], spans[hi_s - 1][1]) if hi_s > lo_s else None
    return mine, (spans[base][0], spans[min(base + ws * m, len(spans)) - 1][1]), m * interval


def all_gather_rows(full, local, device=None, group=None):
    """One all-gather of equal-sized row blocks: `full` [ws * rows, ...] receives rank r's `local` [rows, ...] at rows
    [r * rows, (r + 1) * rows).  The colour loop of the sharded `gen_mesh` calls it once per round, from its copy stream
    (RCCL joins the stream that is current at the call, so the compute stream is not held up)."""
    import torch.distributed as dist
    _timed(lambda: _all_gather(full, local, group), full.numel() * full.element_size(), full.device)
    return full


# ---- what the collectives of one sharded call cost (bench.py puts it into the N > 1 line) --------------------------------
_stats = None


def stats_begin():
    """Start recording: every collective of the sharded entry points is bracketed by events on the stream it is issued
    from and its gathered bytes are counted, until `stats_end()`; so is the work every rank repeats on the replicated field
    (`replicated()`: octree bookkeeping, marching cubes below the slab threshold) - what does not shrink with N."""
    global _stats
    _stats = {'events': [], 'bytes': 0, 'calls': 0, 'host_s': 0.0, 'rep_events': [], 'rep_host_s': 0.0}


def stats_end():
    """{'collectives': n, 'collective_bytes': gathered bytes per rank, 'collective_s': time between the events around the
    collectives (device tensors; it includes waiting for the slowest rank), 'collective_host_s': host time inside the
    calls}.  Synchronises the events."""
    global _stats
    st, _stats = _stats, None
    if st is None:
        return {}
    ms = 0.0
    for (e0, e1) in st['events']:
        e1.synchronize()
        ms += e0.elapsed_time(e1)
    rep_ms = 0.0
    for (e0, e1) in st['rep_events']:
        e1.synchronize()
        rep_ms += e0.elapsed_time(e1)
    return {'collectives': st['calls'], 'collective_bytes': st['bytes'], 'collective_s': round(ms * 1e-3, 5),
            'collective_host_s': round(st['host_s'], 5),
            # octree bookkeeping + (replicated) marching cubes of this rank: device time between the events around them, and
            # the host time of those calls (the per-level count read-back waits for the level's evaluation, so the host
            # figure includes it and is not additive)
            'replicated_s': round(rep_ms * 1e-3, 5), 'replicated_host_s': round(st['rep_host_s'], 5)}


def _timed(fn, nbytes, device):
    import time
    if _stats is None:
        return fn()
    dev = torch.device(device)
    ev = None
    if dev.type == 'cuda':
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(torch.cuda.current_stream(dev))
    h0 = time.perf_counter()
    out = fn()
    _stats['host_s'] += time.perf_counter() - h0
    if ev is not None:
        ev[1].record(torch.cuda.current_stream(dev))
        _stats['events'].append(ev)
    _stats['bytes'] += int(nbytes)
    _stats['calls'] += 1
    return out


def replicated(fn, device):
    """Run `fn()` - work every rank repeats on the replicated field - and, while statistics are recorded, book the device time
    between two events around it."""
    import time
    if _stats is None:
        return fn()
    dev = torch.device(device)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if dev.type == 'cuda' else None
    if ev:
        ev[0].record(torch.cuda.current_stream(dev))
    h0 = time.perf_counter()
    out = fn()
    _stats['rep_host_s'] += time.perf_counter() - h0
    if ev:
        ev[1].record(torch.cuda.current_stream(dev))
        _stats['rep_events'].append(ev)
    return out


def _broadcast(t, src, group=None):
    import torch.distributed as dist
    _timed(lambda: dist.broadcast(t, src=src, group=group), t.numel() * t.element_size(), t.device)


def filter_sharded(net, coarse_image, fine_image, group=None):
    """`filter_global` + `filter_local` of ONE subject dealt out over the ranks (every rank calls it with the same images;
    world size >= 2, both normal nets present, netG in eval mode - the `recon()` configuration - else the plain replicated
    calls): the two normal nets (`networks.py:131-207`, 435 GFLOP each) run on ranks 0 and 1 at the same time, their maps
    are broadcast (2 x 3 MB), then rank 0 runs the fine HGFilter (920 GFLOP) while rank 1 runs the coarse one (212 GFLOP)
    and the two feature maps are broadcast (2 x 16.8 MB at 1024 px) and installed on the other ranks (`pifu_set_features`).
    The chain netF -> fine HGFilter is what remains on the critical path (~15 ms instead of the ~21 ms of a whole pass on
    every rank); same kernels on the same inputs: every rank ends with bit-identical maps."""
    rank, ws = world()
    g = net.netG
    split = (collective_active() and ws >= 2 and g.netF is not None and g.netB is not None and not g.training
             and not (net.training and int(net.opt.num_stack) > 1))
    if not split:
        net.filter_global(coarse_image)
        net.filter_local(fine_image[:, None])
        return False
    dev = net.device
    h, w = coarse_image.shape[2:]
    # (1) one normal net per rank (ranks 0 and 1), the maps to everybody
    if rank == 0:
        g.filter(coarse_image, _part='F')
    elif rank == 1:
        g.filter(coarse_image, _part='B')
    for name in ('nmlF', 'nmlB'):
        t = getattr(g, name, None)
        if t is None or tuple(t.shape) != (1, 3, h, w):
            setattr(g, name, torch.empty(1, 3, h, w, device=dev))
    _broadcast(g.nmlF, 0, group)
    _broadcast(g.nmlB, 1, group)
    # (2) the fine hourglass on rank 0, the coarse one on rank 1, the maps to everybody
    if rank == 0:
        net.filter_local(fine_image[:, None])
    if rank == 1:
        g.filter(coarse_image, _part='hourglass')
    fh, fw = fine_image.shape[2] // 2, fine_image.shape[3] // 2
    ff = net.im_feat_list[-1] if rank == 0 else torch.empty(1, int(net.opt.hg_dim), fh, fw, device=dev)
    fc = g.im_feat_list[-1] if rank == 1 else torch.empty(1, int(g.opt.hg_dim), h // 4, w // 4, device=dev)
    ff, fc = ff.contiguous(), fc.contiguous()
    _broadcast(ff, 0, group)
    _broadcast(fc, 1, group)
    if rank != 0:
        net.set_features(ff)
    if rank != 1:
        g.set_features(fc)
    return True


# `reconstruction(..., sharded=True)` runs marching cubes over one axis-0 slab per rank from this resolution on (below it
# the replicated run is a few milliseconds and cheaper than the three exchanges of the slab form)
SLAB_MC_MIN_RESOLUTION = 1024

# Size guard of `gen_mesh(..., sharded=True)` / `reconstruction(..., sharded=True)`: below this grid resolution every rank
# computes the whole call by itself (identical results on every rank, no collective).  A 64^3 call is ~15 ms of which the
# encoders are most; its octree evaluates ~0.3 M points in 2 - 3 levels, and one exchange per level + one per colour round +
# the four broadcasts of the dealt-out encoders (each a launch, a wait for the slowest rank and ~10 us of xGMI latency) cost
# what they save.  From 128^3 on (0.055 s on one GPU, ~30 % of it the encoder chain) the sharded call is expected to win;
# the threshold is a module variable so that a measured curve can move it.
SHARDED_MIN_RESOLUTION = 128


def sharded_call(resolution):
    """True when a sharded entry point called at `resolution` exchanges data: more than one rank (or the single-rank
    self-test) and a grid at or above the size guard."""
    return collective_active() and int(resolution) >= SHARDED_MIN_RESOLUTION


def marching_cubes_sharded(ctx, volume, level, group=None):
    """Lewiner marching cubes of a replicated device volume with ONE axis-0 slab per rank (the serial scan of
    `marching_cubes_lewiner`, `mesh_util.py:84`, has axis 0 outermost, so a slab owns a contiguous range of the mesh's
    vertex and face numbers).  Exchanges: the per-slab counts and value ranges (exclusive prefix = numbering offset, and
    scikit-image's two exceptions, raised on every rank alike), the vertex numbers on each slab's top node plane
    (int32 [2][n1][n2], read by the slab above), and the finished vertex / face arrays.  Returns (verts float32 [V, 3],
    faces int32 [F, 3]) on the device, identical on every rank and bit-identical to `mesh_util.marching_cubes_device`."""
    import torch.distributed as dist
    from . import _lib
    from .mesh_util import SlabMarchingCubes, check_mc_range, marching_cubes_device, slab_planes
    if not collective_active():
        return marching_cubes_device(ctx, volume, level)
    rank, ws = world()
    vol = volume.contiguous()
    if vol.dim() != 3 or min(vol.shape) < 2:
        raise ValueError('Input array must be at least 2x2x2.')
    dev = vol.device
    n1, n2 = vol.shape[1], vol.shape[2]
    mine = slab_planes(vol.shape[0], ws)[rank]
    step = SlabMarchingCubes(ctx, vol, level, *mine) if mine is not None else None
    nv, nf, lo, hi = step.count() if step else (0, 0, float('inf'), float('-inf'))
    # float64 carries the counts exactly (< 2^31 vertices by the int32 face indices) next to the range
    stat = torch.tensor([nv, nf, lo, hi], device=dev, dtype=torch.float64)
    stats = torch.empty(ws * 4, device=dev, dtype=torch.float64)      # gathered tensors are concatenated along axis 0
    _timed(lambda: _all_gather(stats, stat, group), stats.numel() * 8, dev)
    stats = stats.view(ws, 4).cpu()
    nvs = [int(x) for x in stats[:, 0]]
    nfs = [int(x) for x in stats[:, 1]]
    check_mc_range(float(level), float(stats[:, 2].min()), float(stats[:, 3].max()), sum(nvs))
    if sum(nvs) >= 2 ** 31:
        raise _lib.PifuError(_lib.ERR_INVALID, 'marching_cubes_sharded: %d vertices do not fit int32 face indices' % sum(nvs))
    if step:
        verts, halo = step.verts(sum(nvs[:rank]))
    else:
        verts = torch.empty(0, 3, device=dev, dtype=torch.float32)
        halo = torch.zeros(2, n1, n2, device=dev, dtype=torch.int32)
    halos = torch.empty(ws * 2, n1, n2, device=dev, dtype=torch.int32)
    _timed(lambda: _all_gather(halos, halo, group), halos.numel() * 4, dev)
    halos = halos.view(ws, 2, n1, n2)
    # slabs without cells are the last ranks (slab_planes deals the cell layers out from rank 0), so the slab below is rank - 1
    faces = step.faces(halos[rank - 1] if rank > 0 else None) if step else torch.empty(0, 3, device=dev, dtype=torch.int32)
    cap_v, cap_f = max(max(nvs), 1), max(max(nfs), 1)
    pad_v = torch.zeros(cap_v, 3, device=dev, dtype=torch.float32)
    pad_v[:nv] = verts
    pad_f = torch.zeros(cap_f, 3, device=dev, dtype=torch.int32)
    pad_f[:nf] = faces
    all_v = torch.empty(ws * cap_v, 3, device=dev, dtype=torch.float32)
    all_f = torch.empty(ws * cap_f, 3, device=dev, dtype=torch.int32)
    _timed(lambda: _all_gather(all_v, pad_v, group), all_v.numel() * 4, dev)
    _timed(lambda: _all_gather(all_f, pad_f, group), all_f.numel() * 4, dev)
    all_v, all_f = all_v.view(ws, cap_v, 3), all_f.view(ws, cap_f, 3)
    return (torch.cat([all_v[r, :nvs[r]] for r in range(ws)]), torch.cat([all_f[r, :nfs[r]] for r in range(ws)]))
