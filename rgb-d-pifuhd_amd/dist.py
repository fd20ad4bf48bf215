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
    identical on every rank."""
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
    dist.all_gather_into_tensor(full, local, group=group)
    return full[:total_points]


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
    return sharded_eval(total, int(chunk), eval_fn, dev, group)


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
        _lib.call('pifu_octree_select', ctx.h, R, reso, _ptr(notproc), None, 0, C.byref(cnt), st)
        n = cnt.value
        idx = torch.empty(max(n, 1), device=dev, dtype=torch.int64)
        vals = None
        if n:
            _lib.call('pifu_octree_select', ctx.h, R, reso, _ptr(notproc), _ptr(idx), n, C.byref(cnt), st)

            def eval_fn(first, count):
                out = torch.empty(count, device=dev, dtype=torch.float32)
                _lib.call('pifu_eval_lattice', ctx.h, res_a, calib_inv.ctypes.data_as(C.c_void_p),
                          calib.ctypes.data_as(C.c_void_p), _ptr(idx), first, count, int(chunk), flags, _ptr(out), st)
                return out
            vals = sharded_eval(n, int(chunk), eval_fn, dev, group).contiguous()
            evaluated += n
        _lib.call('pifu_octree_commit', ctx.h, R, reso, float(threshold), _ptr(idx), _ptr(vals), n, _ptr(sdf),
                  _ptr(notproc), st)
        if reso <= 1:
            break
        reso //= 2
    return sdf, evaluated


def shard_range(n_items: int, world_size: int, rank: int):
    """Contiguous share [i0, i1) of `n_items` for `rank`, and the per-rank capacity."""
    per = (n_items + world_size - 1) // world_size
    i0 = min(rank * per, n_items)
    return i0, min(i0 + per, n_items), per


def sharded_calls(spans, call_fn, width, device, group=None):
    """Shard a list of independent calls (each one GroupNorm batch, e.g. the 50 000-vertex `calc_normal`
    calls of `gen_mesh`, `reconstruction.py:58-70`) across ranks by contiguous call ranges.

    spans: list of (left, right) row ranges, ascending and disjoint; call_fn(left, right) -> float32 tensor
    [right-left, width] on `device`.  Returns float32 [spans[-1][1], width] (rows outside every span are 0),
    identical on every rank after ONE all-gather."""
    import torch.distributed as dist
    rank, ws = world()
    n_rows = spans[-1][1] if spans else 0
    out = torch.zeros(n_rows, width, device=device, dtype=torch.float32)
    if not collective_active():
        for (l, r) in spans:
            out[l:r] = call_fn(l, r)
        return out
    i0, i1, per = shard_range(len(spans), ws, rank)
    slot_rows = max(max((r - l) for (l, r) in spans), 1) * per
    local = torch.zeros(slot_rows, width, device=device, dtype=torch.float32)
    row = 0
    for (l, r) in spans[i0:i1]:
        local[row:row + (r - l)] = call_fn(l, r)
        row += r - l
    full = torch.empty(ws * slot_rows, width, device=device, dtype=torch.float32)
    dist.all_gather_into_tensor(full, local, group=group)
    for rk in range(ws):
        j0, j1, _ = shard_range(len(spans), ws, rk)
        row = rk * slot_rows
        for (l, r) in spans[j0:j1]:
            out[l:r] = full[row:row + (r - l)]
            row += r - l
    return out


def gather_runs(runs, mine, run_fn, width, device, group=None):
    """`runs`: one contiguous row range (left, right) per rank that has work, ascending; `mine`: this rank's run (list of
    0 or 1 entries); run_fn(left, right) -> float32 [right-left, width].  One all-gather; returns [runs[-1][1], width]."""
    import torch.distributed as dist
    rank, ws = world()
    n_rows = runs[-1][1] if runs else 0
    slot_rows = max(max((r - l) for (l, r) in runs), 1) if runs else 1
    local = torch.zeros(slot_rows, width, device=device, dtype=torch.float32)
    for (l, r) in mine:
        local[:r - l] = run_fn(l, r)
    full = torch.empty(ws * slot_rows, width, device=device, dtype=torch.float32)
    dist.all_gather_into_tensor(full, local, group=group)
    out = torch.zeros(n_rows, width, device=device, dtype=torch.float32)
    for rk, (l, r) in enumerate(runs):          # ranks with work are the first len(runs) ranks (shard_range is monotone)
        out[l:r] = full[rk * slot_rows: rk * slot_rows + (r - l)]
    return out


# `reconstruction(..., sharded=True)` runs marching cubes over one axis-0 slab per rank from this resolution on (below it
# the replicated run is a few milliseconds and cheaper than the three exchanges of the slab form)
SLAB_MC_MIN_RESOLUTION = 1024


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
    dist.all_gather_into_tensor(stats, stat, group=group)
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
    dist.all_gather_into_tensor(halos, halo, group=group)
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
    dist.all_gather_into_tensor(all_v, pad_v, group=group)
    dist.all_gather_into_tensor(all_f, pad_f, group=group)
    all_v, all_f = all_v.view(ws, cap_v, 3), all_f.view(ws, cap_f, 3)
    return (torch.cat([all_v[r, :nvs[r]] for r in range(ws)]), torch.cat([all_f[r, :nfs[r]] for r in range(ws)]))
