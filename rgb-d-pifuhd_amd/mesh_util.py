"""Grid evaluation, marching cubes and OBJ output: drop-in for the reference's `mesh_util.py`.

Same function names / arguments / return values / error behaviour as `mesh_util.py:12-198`, but the grid
never exists on the host: lattice points are generated on the device from their indices, the float64 SDF
and the octree state stay in HBM, and marching cubes (Lewiner, scikit-image 0.18.3 semantics) runs as
HIP kernels.  Everything computes through the C ABI in `include/pifuhd_hip.h`.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np
import torch
from numpy.linalg import inv

from . import _lib
from .nets import PIFuMRNet, PIFuNetwNML, _ptr, _stream_ptr


def create_grid(resX, resY, resZ, b_min=np.array([-1, -1, -1]), b_max=np.array([1, 1, 1]), transform=None):
    """`create_grid` (`mesh_util.py:12-38`).  Kept for API compatibility (small grids / custom fields);
    `reconstruction()` does not call it - points are generated on the device."""
    coords = np.mgrid[:resX, :resY, :resZ].reshape(3, -1).astype(np.float64)
    mat = np.eye(4)
    length = np.asarray(b_max, dtype=np.float64) - np.asarray(b_min, dtype=np.float64)
    mat[0, 0], mat[1, 1], mat[2, 2] = length[0] / resX, length[1] / resY, length[2] / resZ
    mat[0:3, 3] = b_min
    coords = coords * np.diag(mat)[:3, None] + mat[:3, 3:4]
    if transform is not None:
        coords = np.matmul(transform[:3, :3], coords) + transform[:3, 3:4]
        mat = np.matmul(transform, mat)
    return coords.reshape(3, resX, resY, resZ), mat


def calib_pair(calib_tensor):
    """(calib float32 [4,4], calib_inv float64 [4,4]) as the reference forms them (`mesh_util.py:61-62`): the inverse is
    taken of the FLOAT32 matrix (numpy inverts in float32) and only then promoted, when it meets the float64 grid."""
    calib = np.ascontiguousarray(calib_tensor[0].detach().cpu().numpy().astype(np.float32))
    return calib, np.ascontiguousarray(inv(calib).astype(np.float64))


def _ctx_of(net):
    if isinstance(net, (PIFuMRNet, PIFuNetwNML)):
        net._sync_weights()          # pending load_state_dict / train-eval changes reach the device before any direct call
        return net._ctx
    raise TypeError('net must be a pifuhd_amd PIFuMRNet / PIFuNetwNML')


def eval_grid_device(net, calib_tensor, resolution, num_samples, use_octree, init_resolution=64, threshold=0.05,
                     full_coarse=False):
    """Device-resident eval_grid / eval_grid_octree for one of our nets.  Returns (sdf float64 device
    tensor [R,R,R], number of evaluated points)."""
    ctx = _ctx_of(net)
    dev = ctx.device
    R = int(resolution)
    calib, calib_inv = calib_pair(calib_tensor)
    sdf = torch.empty(R, R, R, device=dev, dtype=torch.float64)
    notproc = torch.empty(R, R, R, device=dev, dtype=torch.uint8) if use_octree else None
    flags = 0
    if isinstance(net, PIFuNetwNML):
        flags |= _lib.Q_COARSE_ONLY
    elif full_coarse:
        flags |= _lib.Q_FULL_COARSE
    n_eval = C.c_int64(0)
    _lib.call('pifu_eval_grid', ctx.h, R, calib_inv.ctypes.data_as(C.c_void_p), calib.ctypes.data_as(C.c_void_p),
              int(num_samples), int(init_resolution) if use_octree else 0, float(threshold), flags, _ptr(sdf),
              _ptr(notproc), C.byref(n_eval), _stream_ptr(dev))
    return sdf, n_eval.value


def marching_cubes_device(ctx, volume, level):
    """Lewiner marching cubes on a device volume (float32 or float64 torch tensor [n0,n1,n2]); returns
    device tensors (verts float32 [V,3], faces int32 [F,3]).  Raises ValueError / RuntimeError exactly
    where scikit-image does."""
    dev = ctx.device
    vol = volume.contiguous()
    if vol.dim() != 3 or min(vol.shape) < 2:
        raise ValueError('Input array must be at least 2x2x2.')
    dtype = {torch.float32: 0, torch.float64: 1}[vol.dtype]
    nv, nf = C.c_int64(0), C.c_int64(0)
    try:
        _lib.call('pifu_mc_count', ctx.h, _ptr(vol), dtype, vol.shape[0], vol.shape[1], vol.shape[2], float(level),
                  C.byref(nv), C.byref(nf), _stream_ptr(dev))
    except _lib.PifuError as e:
        if e.code == _lib.ERR_MC_LEVEL:
            raise ValueError(e.message)
        if e.code == _lib.ERR_MC_EMPTY:
            raise RuntimeError(e.message)
        raise
    verts = torch.empty(nv.value, 3, device=dev, dtype=torch.float32)
    faces = torch.empty(nf.value, 3, device=dev, dtype=torch.int32)
    _lib.call('pifu_mc_emit', ctx.h, _ptr(verts), _ptr(faces), _stream_ptr(dev))
    return verts, faces


def slab_planes(n0, nslabs):
    """Axis-0 slabs of a volume with `n0` node planes: the n0 - 1 cell layers are dealt out in contiguous, near-equal
    runs; returns one (z_first, n_planes) per slab (neighbours share a node plane), None for slabs left without cells."""
    cells = n0 - 1
    per = (cells + nslabs - 1) // nslabs
    out = []
    for k in range(nslabs):
        c0 = min(k * per, cells)
        c1 = min(c0 + per, cells)
        out.append((c0, c1 - c0 + 1) if c1 > c0 else None)
    return out


class SlabMarchingCubes:
    """The three steps of marching cubes on ONE axis-0 slab of a device volume (`pifu_mc_slab_*`); the caller supplies
    what crosses slabs: the vertex-number offset and the halo plane of the slab below."""

    def __init__(self, ctx, volume, level, z_first, n_planes):
        if volume.dim() != 3 or min(volume.shape[1:]) < 2 or n_planes < 2:
            raise ValueError('Input array must be at least 2x2x2.')
        self.ctx, self.level, self.z_first = ctx, float(level), int(z_first)
        self.slab = volume[z_first:z_first + n_planes]              # a view: axis 0 is the slowest axis
        assert self.slab.is_contiguous()
        self.dtype = {torch.float32: 0, torch.float64: 1}[volume.dtype]
        self.nv = self.nf = 0

    def count(self):
        nv, nf, lo, hi = C.c_int64(0), C.c_int64(0), C.c_float(0), C.c_float(0)
        v = self.slab
        _lib.call('pifu_mc_slab_count', self.ctx.h, _ptr(v), self.dtype, v.shape[0], v.shape[1], v.shape[2], self.z_first,
                  self.level, C.byref(nv), C.byref(nf), C.byref(lo), C.byref(hi), _stream_ptr(self.ctx.device))
        self.nv, self.nf = nv.value, nf.value
        return self.nv, self.nf, lo.value, hi.value

    def verts(self, vert_base, want_halo=True):
        dev = self.ctx.device
        verts = torch.empty(self.nv, 3, device=dev, dtype=torch.float32)
        halo = torch.zeros(2, self.slab.shape[1], self.slab.shape[2], device=dev, dtype=torch.int32) if want_halo else None
        _lib.call('pifu_mc_slab_verts', self.ctx.h, _ptr(verts) if self.nv else None, int(vert_base),
                  _ptr(halo) if want_halo else None, _stream_ptr(dev))
        return verts, halo

    def faces(self, halo_in):
        dev = self.ctx.device
        faces = torch.empty(self.nf, 3, device=dev, dtype=torch.int32)
        _lib.call('pifu_mc_slab_faces', self.ctx.h, _ptr(faces) if self.nf else None,
                  _ptr(halo_in) if halo_in is not None else None, _stream_ptr(dev))
        return faces


def check_mc_range(level, lo, hi, nverts):
    """The two exceptions scikit-image raises (`mesh_util.py:84` -> marching_cubes_lewiner), from totals over all slabs."""
    if level < lo or level > hi:
        raise ValueError('Surface level must be within volume data range.')
    if nverts == 0:
        raise RuntimeError('No surface found at the given iso value.')


def marching_cubes_slabs(ctx, volume, level, nslabs):
    """`marching_cubes_device` computed slab by slab in one process (what `dist.marching_cubes_sharded` does with one slab
    per rank): same vertices and faces, bit for bit; the working set is that of one slab."""
    vol = volume.contiguous()
    if vol.dim() != 3 or min(vol.shape) < 2:
        raise ValueError('Input array must be at least 2x2x2.')
    runs = [r for r in slab_planes(vol.shape[0], nslabs) if r is not None]
    verts, faces, halo, base = [], [], None, 0
    lo, hi = float('inf'), float('-inf')
    for (z_first, n_planes) in runs:
        step = SlabMarchingCubes(ctx, vol, level, z_first, n_planes)
        nv, _, l, h = step.count()
        lo, hi = min(lo, l), max(hi, h)
        v, halo_out = step.verts(base)
        faces.append(step.faces(halo))
        verts.append(v)
        halo, base = halo_out, base + nv
    check_mc_range(float(level), lo, hi, base)
    return torch.cat(verts), torch.cat(faces)


def transform_verts(ctx, verts_d, trans_mat, defer_host=False, host=True):
    """`(trans_mat[:3, :3] @ verts.T + trans_mat[:3, 3:4]).T` in float64 (`mesh_util.py:87-89`) on the device
    (`pifu_transform_verts`).  Returns (host float64 [V, 3], the same array as a device tensor).  The kernel evaluates each
    coordinate as the FMA chain `fma(m2, z, fma(m1, y, m0 * x))` followed by the addition of the translation - the order in
    which an FMA-capable BLAS dgemm (what numpy's matmul calls for this shape) accumulates K = 3 - so the result is a fixed
    function of its inputs on every machine; the tests compare it with numpy's matmul on the WHOLE array
    (`tests/test_gpu_grid_mc.py`), nothing is sampled or recomputed here."""
    n = verts_d.shape[0]
    rows = np.ascontiguousarray(trans_mat[:3, :4], dtype=np.float64)
    out_d = torch.empty(n, 3, device=verts_d.device, dtype=torch.float64)
    _lib.call('pifu_transform_verts', ctx.h, _ptr(verts_d), n, rows.ctypes.data_as(C.c_void_p), _ptr(out_d),
              _stream_ptr(ctx.device))
    if not host:
        return None, out_d
    if defer_host:
        host_arr, ev = to_host_async(out_d)
        return host_arr, out_d, ev
    return to_host(out_d), out_d


def reconstruction(net, cuda, calib_tensor, resolution, b_min, b_max, thresh=0.5, use_octree=False,
                   num_samples=10000, transform=None, return_sdf=False, sharded=False, return_device_verts=False,
                   _defer_host=False, _host_result=True):
    """`reconstruction` (`mesh_util.py:40-96`): returns (verts float64 [V,3], faces int32 [F,3], normals,
    values) or -1.  Like the reference, `b_min`, `b_max` and `transform` are ignored (the box is always
    [-1,1]^3 before calib^-1, SURVEY C-5).  `normals` / `values` (discarded by gen_mesh) are returned empty.
    `sharded=True`: the grid evaluation is split over the ranks of the initialised process group
    (`pifuhd_amd.dist`); every rank gets the identical field; marching cubes runs redundantly on it, or - from
    `dist.SLAB_MC_MIN_RESOLUTION` on - one axis-0 slab per rank (`dist.marching_cubes_sharded`), with the same result.
    `return_device_verts=True` appends the float64 vertices as a device tensor (the colour loop of `gen_mesh` continues
    from it without a host round trip).  `_defer_host=True` (gen_mesh): the vertex and face arrays cross to the host on
    the copy stream while the caller goes on issuing work; a `HostCopies` is appended to the result, and the arrays' CONTENTS
    are valid only after its `wait_faces()` / `wait_verts()`.  `_host_result=False` (a rank of a sharded gen_mesh that
    returns nothing): no host copies at all - `verts` and `faces` come back as None, the device vertices are still there."""
    ctx = _ctx_of(net)
    from . import dist as pdist
    sharded = sharded and pdist.sharded_call(resolution)      # below the size guard every rank computes the call alone
    if sharded:
        flags = _lib.Q_COARSE_ONLY if isinstance(net, PIFuNetwNML) else 0
        R = int(resolution)
        if use_octree:
            sdf, _ = pdist.eval_grid_octree_sharded(net, calib_tensor, R, num_samples, flags=flags)
        else:
            sdf = pdist.eval_lattice_sharded(net, calib_tensor, (R, R, R), num_samples, flags).double().view(R, R, R)
    else:
        sdf, _ = eval_grid_device(net, calib_tensor, resolution, num_samples, use_octree)
    try:
        # only what the reference's own try block can fail on - the level outside the volume's range (ValueError) or no
        # surface (RuntimeError), mesh_util.py:84,94-96 - is turned into the printed -1; anything else (an allocation, a
        # failed copy) surfaces as the error it is
        if sharded and int(resolution) >= pdist.SLAB_MC_MIN_RESOLUTION:
            verts_d, faces_d = pdist.marching_cubes_sharded(ctx, sdf, thresh)
        elif sharded:
            verts_d, faces_d = pdist.replicated(lambda: marching_cubes_device(ctx, sdf, thresh), ctx.device)
        else:
            verts_d, faces_d = marching_cubes_device(ctx, sdf, thresh)
    except _lib.PifuError:               # (a RuntimeError too: a failed HIP call is not "no surface")
        raise
    except (ValueError, RuntimeError):
        print('error cannot marching cubes')                                             # mesh_util.py:94-96
        return -1
    pending = HostCopies()
    pending.n_verts, pending.n_faces = int(verts_d.shape[0]), int(faces_d.shape[0])
    if not _host_result:
        faces = None
    elif _defer_host:
        faces, pending.faces = to_host_async(faces_d)
        pending.keep.append(faces)
    else:
        faces = to_host(faces_d)
    _, calib_inv = calib_pair(calib_tensor)
    mat = np.eye(4)
    mat[0, 0] = mat[1, 1] = mat[2, 2] = 2.0 / resolution
    mat[0:3, 3] = -1.0
    trans_mat = np.matmul(calib_inv, mat)
    if not _host_result:
        verts, verts_dev = transform_verts(ctx, verts_d, trans_mat, host=False)
    elif _defer_host:
        verts, verts_dev, pending.verts = transform_verts(ctx, verts_d, trans_mat, defer_host=True)   # mesh_util.py:87-89
        pending.keep.append(verts)
    else:
        verts, verts_dev = transform_verts(ctx, verts_d, trans_mat)
    if faces is not None and np.linalg.det(trans_mat[:3, :3]) < 0.0:
        faces = faces[:, ::-1]                                                       # mesh_util.py:91-92
    normals = np.zeros((0, 3), np.float32)
    values = np.zeros((0,), np.float32)
    out = (verts, faces, normals, values)
    if return_sdf:
        out += (sdf,)
    elif return_device_verts:
        out += (verts_dev,)
    return out + (pending,) if _defer_host else out


class HostCopies:
    """The device-to-host copies `reconstruction(_defer_host=True)` left running on the copy stream.  It holds the arrays
    themselves until they have been waited for: they are views of pinned slabs that are handed out again once the array
    is gone, so a caller that fails between `reconstruction()` and the waits must not be able to free a slab a copy is
    still writing to (`wait_all()` in its `finally`)."""

    def __init__(self):
        self.faces = self.verts = None
        self.keep = []
        self.n_verts = self.n_faces = 0

    def wait_all(self):
        self.wait_faces()
        self.wait_verts()
        self.keep = []

    def wait_faces(self):
        if self.faces is not None:
            self.faces.synchronize()
            self.faces = None

    def wait_verts(self):
        if self.verts is not None:
            self.verts.synchronize()
            self.verts = None


_pinned = {}
_copy_streams = {}


def copy_stream(dev):
    """One side stream per device for transfers that should not queue behind (or in front of) the compute stream's kernels."""
    dev = torch.device(dev)
    side = _copy_streams.get(dev)
    if side is None:
        side = _copy_streams[dev] = torch.cuda.Stream(dev)
    return side


_slabs = []          # [pinned uint8 tensor, weak reference to the array last handed out from it]
POOL_RESULTS = os.environ.get('PIFU_POOL_RESULTS', '1') != '0'      # False: plain `.cpu().numpy()` results
POOL_MAX_BYTES = int(os.environ.get('PIFU_POOL_MAX_BYTES', str(1 << 30)))      # page-locked bytes the result pool may hold
POOL_MAX_SLABS = 6


def _pool_bytes():
    return sum(s[0].numel() for s in _slabs)


def _drop_slab(slab):
    # by identity: list.remove() would compare the entries, i.e. the tensors, element by element
    _slabs[:] = [s for s in _slabs if s is not slab]


def to_host_async(t):
    """`to_host` issued on the copy stream: returns (array, event) - the array's contents are valid once the event has
    completed (`event.synchronize()`), the compute stream is not held up meanwhile - or (array, None) where `to_host` would
    not use a pinned slab (small tensors, pool full or switched off): a finished, synchronous copy then."""
    arr, view = _to_host(t, issue=False)
    if view is None:
        return arr, None
    dev = t.device
    side, cur = copy_stream(dev), torch.cuda.current_stream(dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        view.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    t.record_stream(side)
    return arr, ev


def to_host(t):
    return _to_host(t)[0]


def _to_host(t, issue=True):
    """Device tensor -> host array OF THE CALLER'S OWN, without a fresh pageable allocation per call (a 100 MB result costs
    ~25 000 page faults plus a staged copy each time): the copy lands in a pinned slab, and the returned array is a view of
    it.  A slab is handed out again only once the array it backed is gone - arrays and every view of them are tracked
    through a weak reference, since numpy views keep their parent alive - so a caller that keeps a result keeps its memory
    (a raw address taken from the array, `arr.ctypes.data`, is NOT tracked: keep the array).  When all slabs are in use,
    when the pool would exceed POOL_MAX_BYTES (default 1 GiB, env PIFU_POOL_MAX_BYTES) or when the tensor is small this is
    the plain `.cpu().numpy()`.  A free slab more than 4x the size of the request is released instead of reused, so one
    1024^3 mesh does not pin its footprint for the life of the process; `release_pinned()` drops everything that is free,
    and PIFU_POOL_RESULTS=0 switches the pool off."""
    import weakref
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes < (4 << 20) or not POOL_RESULTS:
        return t.cpu().numpy(), None
    want = nbytes + nbytes // 4
    slot = None
    for s in list(_slabs):
        if s[1]() is not None:
            continue
        if s[0].numel() > 4 * want:                   # free and far too large: give the pages back
            _drop_slab(s)
            continue
        if s[0].numel() >= nbytes:
            slot = s
            break
    if slot is None:
        # no free slab fits: free slabs that are too small make room for one that does - smallest first, as many as the byte
        # cap or the slab count asks for (subjects of growing size would otherwise fill the pool with slabs that fit nothing
        # and leave every later result on the pageable, synchronous path for good)
        for s in sorted((s for s in _slabs if s[1]() is None), key=lambda s: s[0].numel()):
            if len(_slabs) < POOL_MAX_SLABS and _pool_bytes() + want <= POOL_MAX_BYTES:
                break
            _drop_slab(s)
        if len(_slabs) >= POOL_MAX_SLABS or _pool_bytes() + want > POOL_MAX_BYTES:
            return t.cpu().numpy(), None
        slot = [torch.empty(want, dtype=torch.uint8, pin_memory=True), lambda: None]
        _slabs.append(slot)
    view = slot[0][:nbytes].view(t.dtype).view(t.shape)
    if issue:
        view.copy_(t)
    arr = view.numpy()
    slot[1] = weakref.ref(arr)
    return arr, view            # (`issue=False`: the caller copies into `view` itself, to_host_async)


def _pin_key(tag, dev):
    """Scratch buffers belong to one (tag, device, host thread): two devices, or two threads calling gen_mesh at the same
    time, never share one."""
    dev = torch.device(dev) if dev is not None else None
    return (tag, None if dev is None else dev.index, threading.get_ident())


def pinned_flat(tag, n, dtype, dev=None):
    """Flat pinned host buffer of at least `n` elements kept (and grown) across calls, one per (tag, device, thread)."""
    key = _pin_key(tag, dev)
    buf = _pinned.get(key)
    if buf is None or buf.numel() < n or buf.dtype != dtype:
        buf = _pinned[key] = torch.empty(max(n, 1), dtype=dtype, pin_memory=True)
    return buf


def pinned_rows(tag, n, dtype, dev=None):
    """[n, 3] view of a pinned host buffer kept (and grown) across calls, one per (tag, device, thread)."""
    key = _pin_key(tag, dev)
    buf = _pinned.get(key)
    if buf is None or buf.shape[0] < n or buf.dtype != dtype:
        buf = _pinned[key] = torch.empty(max(n, 1), 3, dtype=dtype, pin_memory=True)
    return buf[:n]


def release_pinned():
    """Give back the page-locked host memory this module holds: every result slab whose array is gone and every scratch
    buffer (`pinned_flat` / `pinned_rows`).  Call it between subjects of very different size, or when the process moves on
    to other work.  Returns the number of bytes released.  Not to be called while a gen_mesh of this process is running."""
    freed = 0
    for s in list(_slabs):
        if s[1]() is None:
            freed += s[0].numel()
            _drop_slab(s)
    for key in list(_pinned):
        freed += _pinned[key].numel() * _pinned[key].element_size()
        del _pinned[key]
    return freed


# ------------------------------------------------------------------------------------------------
# Generic host-callback variants (same signatures as the reference).  The octree bookkeeping still runs
# in the HIP kernels; only `eval_func` (numpy in, numpy out) is the caller's.
# ------------------------------------------------------------------------------------------------
def batch_eval(points, eval_func, num_samples=512 * 512 * 512):
    """`batch_eval` (`mesh_util.py:98-114`)."""
    n = points.shape[1]
    sdf = np.zeros(n)
    for s in range(0, n, num_samples):
        sdf[s:s + num_samples] = eval_func(points[:, s:s + num_samples])
    return sdf


def eval_grid(coords, eval_func, num_samples=512 * 512 * 512):
    """`eval_grid` (`mesh_util.py:116-120`)."""
    resolution = coords.shape[1:4]
    return batch_eval(coords.reshape([3, -1]), eval_func, num_samples=num_samples).reshape(resolution)


def lattice_points(resolution, calib_inv, flat_index):
    """World points [3, n] (float64) of the C-ordered lattice entries `flat_index` of a resolution^3 grid over [-1, 1)^3:
    `create_grid` followed by the calib^-1 product of `reconstruction` (`mesh_util.py:28-33,59-65`) for those entries only."""
    R = int(resolution)
    idx = np.asarray(flat_index, dtype=np.int64)
    ijk = np.stack([idx // (R * R), (idx // R) % R, idx % R], 1).astype(np.float64)
    grid = np.concatenate([ijk * (2.0 / R) + (-1.0), np.ones((len(idx), 1))], 1)
    return np.matmul(grid, np.asarray(calib_inv, dtype=np.float64).T)[:, :3].T


def eval_grid_octree(coords, eval_func, init_resolution=64, threshold=0.05, num_samples=512 * 512 * 512,
                     device='cuda', resolution=None, calib_inv=None):
    """`eval_grid_octree` (`mesh_util.py:124-187`) for an arbitrary host `eval_func`; cubic grids.  The bookkeeping (test
    point selection, 8-corner min / max, block fills) runs in the HIP kernels; only `eval_func` (numpy [3, n] in, numpy
    [n] out) is the caller's.  `coords` is the reference's [3, R, R, R] host grid, or None with `resolution` and
    `calib_inv` given: the points of a level are then generated from their indices (a 512^3 float64 grid is 3.2 GB)."""
    if coords is not None:
        R = coords.shape[1]
        if coords.shape[2] != R or coords.shape[3] != R:
            raise NotImplementedError('cubic grids only')
        flat = coords.reshape(3, -1)

        def points_of(index):
            return flat[:, index]
    else:
        if resolution is None or calib_inv is None:
            raise ValueError('eval_grid_octree needs either the coords grid or resolution + calib_inv')
        R = int(resolution)

        def points_of(index):
            return lattice_points(R, calib_inv, index)
    from .nets import _Ctx
    ctx = _Ctx(device)
    dev = ctx.device
    st = _stream_ptr(dev)
    sdf = torch.empty(R, R, R, device=dev, dtype=torch.float64)
    notproc = torch.empty(R, R, R, device=dev, dtype=torch.uint8)
    _lib.call('pifu_octree_begin', ctx.h, R, _ptr(sdf), _ptr(notproc), st)
    reso = R // init_resolution
    while reso > 0:
        cnt = C.c_int64(0)
        _lib.call('pifu_octree_select', ctx.h, R, reso, _ptr(notproc), None, 0, C.byref(cnt), st)
        n = cnt.value
        idx = torch.empty(max(n, 1), device=dev, dtype=torch.int64)
        vals = torch.empty(max(n, 1), device=dev, dtype=torch.float32)
        if n:
            _lib.call('pifu_octree_select', ctx.h, R, reso, _ptr(notproc), _ptr(idx), n, C.byref(cnt), st)
            out = batch_eval(points_of(idx[:n].cpu().numpy()), eval_func, num_samples=num_samples)
            vals[:n] = torch.from_numpy(out.astype(np.float32)).to(dev)
        _lib.call('pifu_octree_commit', ctx.h, R, reso, float(threshold), _ptr(idx), _ptr(vals), n, _ptr(sdf),
                  _ptr(notproc), st)
        if reso <= 1:
            break
        reso //= 2
    return sdf.cpu().numpy()


class ObjFile:
    """`save_obj_mesh_with_color` (`mesh_util.py:189-198`) in two steps: the face lines are formatted in the background from
    the moment the triangles are known, `finish(verts, colors)` adds the vertex lines and writes the file."""

    def __init__(self, mesh_path, faces):
        f = np.asarray(faces)
        if f.dtype != np.int32 or f.ndim != 2 or f.shape[1] != 3 or f.strides[0] % 4 or f.strides[1] % 4:
            f = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        self._faces = f                                  # kept alive until finish(): the worker threads read it
        self._keep, self._streamed = [], False
        self._h = C.c_void_p()
        _lib.call('pifu_obj_begin', str(mesh_path).encode(), f.ctypes.data_as(C.c_void_p), len(f), f.strides[0] // 4,
                  f.strides[1] // 4, C.byref(self._h))

    def layout(self, verts, colors32):
        """Declare the complete vertex arrays - positions float64 [n,3] (final), colours float32 [n,3] (a buffer the caller
        still fills in; the float32 values the reference copies into its float64 colour array).  The writer derives every
        file offset from the positions, puts the face lines in place at once and writes a block of vertex lines as soon
        as `ready(upto)` has declared its colours valid."""
        if verts.dtype != np.float64 or colors32.dtype != np.float32 or not verts.flags.c_contiguous or \
                not colors32.flags.c_contiguous or verts.shape != colors32.shape or verts.ndim != 2 or verts.shape[1] != 3:
            raise ValueError('layout: verts float64 [n,3] and colours float32 [n,3], both C-contiguous (no copies are made)')
        self._keep = [verts, colors32]                   # read by the worker threads until finish()
        _lib.call('pifu_obj_layout', self._h, verts.ctypes.data_as(C.c_void_p), colors32.ctypes.data_as(C.c_void_p), len(verts))
        self._streamed = True

    def ready(self, upto):
        """The colours of vertices [0, upto) are valid."""
        _lib.call('pifu_obj_ready', self._h, int(upto))

    def finish(self, verts=None, colors=None):
        """Write the file: with the vertices given here (float64 colours, `save_obj_mesh_with_color`'s arguments), or - after
        `layout` - with the declared arrays, every colour of which is valid by now."""
        h, self._h = self._h, None
        try:
            if self._streamed:
                if verts is not None or colors is not None:
                    raise ValueError('finish(): the vertices of this file were already declared by layout()')
                _lib.call('pifu_obj_finish', h, None, None, 0)
            else:
                v = np.ascontiguousarray(verts, dtype=np.float64)
                c = np.ascontiguousarray(colors, dtype=np.float64)
                _lib.call('pifu_obj_finish', h, v.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), len(v))
        finally:
            self._faces = None
            self._keep = []

    def __del__(self):                                   # an abandoned writer (exception between the two steps)
        if getattr(self, '_h', None):
            try:
                _lib.call('pifu_obj_finish', self._h, None, None, -1)      # abandon: the file is not touched
            except Exception:
                pass
            self._h = None


def save_obj_mesh_with_color(mesh_path, verts, faces, colors):
    """`save_obj_mesh_with_color` (`mesh_util.py:189-198`), native writer."""
    v = np.ascontiguousarray(verts, dtype=np.float64)
    c = np.ascontiguousarray(colors, dtype=np.float64)
    f = np.ascontiguousarray(faces, dtype=np.int32)
    _lib.call('pifu_write_obj', str(mesh_path).encode(), v.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
              len(v), f.ctypes.data_as(C.c_void_p), len(f))
