"""`gen_mesh`: drop-in for the reference's `reconstruction.gen_mesh` (`reconstruction.py:25-72`).

Same signature, same side effects (`<save_path>` OBJ + `<save_path[:-4]>.png` with image | front normal |
back normal), same quirks (SURVEY C-6: the last vertex is never coloured because the final slice ends
at -1; a vertex count that is a multiple of 50 000 raises like the reference does).  The heavy lifting
(filters, octree grid evaluation, marching cubes, per-vertex normals) runs on the GPU through
`include/pifuhd_hip.h`.
"""
from __future__ import annotations

import os
import threading
import time

import numpy as np
import torch

from numpy.linalg import inv

from . import _lib
from .mesh_util import ObjFile, copy_stream, pinned_flat as mesh_util_pinned, pinned_rows, reconstruction, save_obj_mesh_with_color
from .nets import index


HOST_THREADS = 0          # set_host_threads(): 0 = no bound of this package's own (PNG: 8 bands, OBJ writer: the machine's, <= 32)


def set_host_threads(n):
    """Bound the host worker threads one gen_mesh call starts (OBJ formatting / writing threads of the library, PNG deflate
    bands): for several processes on one host - one per GPU - `os.cpu_count() // ranks`.  0 = default."""
    global HOST_THREADS
    HOST_THREADS = max(int(n), 0)
    _lib.call('pifu_set_host_threads', HOST_THREADS)


def write_png(path, rgb, level=1, bands=8):
    """8-bit RGB PNG of `rgb` (uint8 [H, W, 3]) with the deflate stream cut into `bands` row bands that are compressed on
    threads (zlib releases the GIL): every band is a raw deflate stream of its own ending on a byte boundary (sync flush;
    the last one finishes the stream), so their concatenation is ONE valid zlib stream - what pigz does.  Filter type 0 on
    every row, compression level 1 like `cv2.imwrite`'s default (`reconstruction.py:54`).  A 512 x 1536 strip: ~8 ms instead
    of the 50 - 130 ms of a single-threaded encoder, which was the tail of gen_mesh whenever the GPU part took less."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    raw = np.zeros((h, 1 + 3 * w), np.uint8)               # filter byte 0 + the row
    raw[:, 1:] = rgb.reshape(h, 3 * w)
    bands = max(1, min(int(bands), h, raw.size // 65536))      # a thread per 64 KB at least
    if HOST_THREADS:
        bands = min(bands, HOST_THREADS)
    cuts = [h * k // bands for k in range(bands + 1)]

    def deflate(k):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        data = raw[cuts[k]:cuts[k + 1]].tobytes()
        return c.compress(data) + c.flush(zlib.Z_FINISH if k == bands - 1 else zlib.Z_SYNC_FLUSH)
    if bands > 1:
        with ThreadPoolExecutor(bands) as pool:
            parts = list(pool.map(deflate, range(bands)))
    else:
        parts = [deflate(0)]
    idat = b'\x78\x01' + b''.join(parts) + struct.pack('>I', zlib.adler32(raw.tobytes()) & 0xffffffff)

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    with open(path, 'wb') as fh:
        fh.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) + chunk(b'IDAT', idat)
                 + chunk(b'IEND', b''))


def _save_debug_png(path, image_tensor_global, background=False):
    """Horizontal strip of the [-1,1] images, RGB (`reconstruction.py:36-54`).  With `background=True` the PNG encoding
    (zlib, GIL released) runs on a thread that the caller joins before returning."""
    strip = torch.cat([image_tensor_global[v, :3] for v in range(image_tensor_global.shape[0])], 2)      # [3, H, n*W]
    arr = ((strip.detach().permute(1, 2, 0) * 0.5 + 0.5) * 255.0).contiguous()
    if not background or not arr.is_cuda:
        write_png(path, np.clip(np.rint(arr.cpu().numpy()), 0, 255).astype(np.uint8))
        return None
    # The pixels cross on the copy stream into pinned memory (issued here, not waited for), the thread waits for that copy
    # alone and encodes.  It must not touch the compute stream: a `.cpu()` from the thread synchronises the stream the octree
    # evaluation is being issued into at that moment and stalls the issuing thread - measured 0 - 35 ms per call.
    dev = arr.device
    host = mesh_util_pinned('png', arr.numel(), torch.float32, dev)[:arr.numel()].view(arr.shape)
    side, cur = copy_stream(dev), torch.cuda.current_stream(dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        host.copy_(arr, non_blocking=True)
        landed = torch.cuda.Event()
        landed.record(side)
    arr.record_stream(side)

    def encode():
        landed.synchronize()
        write_png(path, np.clip(np.rint(host.numpy()), 0, 255).astype(np.uint8))
    th = threading.Thread(target=encode)
    th.start()
    return th


def _prof_read(net, which):
    """(ms, launches, work) of one profiling slot of the net's context, not reset (`pifu_profile_read`; synchronises)."""
    import ctypes as C
    ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
    _lib.call('pifu_profile_read', net._ctx.h, which, C.byref(ms), C.byref(n), C.byref(w), 0)
    return ms.value, n.value, w.value


def _normal_spans(n_verts, interval=50000):
    """The slices of the reference's colour loop (`reconstruction.py:58-70`): `interval`-vertex calls, the last
    one ending at -1, so the last vertex is never coloured (SURVEY C-6)."""
    spans = []
    for i in range(n_verts // interval + 1):
        left = i * interval
        right = n_verts - 1 if i == n_verts // interval else (i + 1) * interval
        spans.append((left, max(right, left)))
    return spans


def _upload_beside(t, dev):
    """`t.to(dev)` issued on the copy stream, so that the transfer does not queue behind the kernels already issued on the
    current stream; the current stream waits for it."""
    dev = torch.device(dev)
    if t.device == dev:
        return t
    side = copy_stream(dev)
    cur = torch.cuda.current_stream(dev)
    with torch.cuda.stream(side):
        out = t.to(device=dev, non_blocking=True)
    cur.wait_stream(side)
    out.record_stream(cur)
    return out


class _Subject:
    """One subject on the device after both encoders ran: what every `gen_mesh*` variant of the reference starts with
    (`reconstruction.py:29-45` and its two copies at `:81-97`, `:131-147`)."""

    def __init__(self, net, data, sharded=False):
        dev = net.device
        self.calib = data['calib'].to(device=dev)
        # the same matrix on the host, for this package's own calls: they only need its 16 numbers there, and fetching them
        # from the device copy would synchronise the stream every time (once per calc_normal group in the colour loop)
        self.calib_host = data['calib'].detach().cpu()
        coarse = data['img_512'].to(device=dev)
        # same two calls as the reference, arranged so that independent work overlaps: the coarse call returns while its
        # hourglass still runs (the fine level does not read it), and the 1024 px image crosses PCIe on a copy stream while
        # the normal nets compute
        if sharded:
            # one subject over several ranks: the encoders themselves are dealt out (`dist.filter_sharded`)
            from . import dist as pdist
            self.image_fine = _upload_beside(data['img'], dev)
            pdist.filter_sharded(net, coarse, self.image_fine)
        else:
            net.filter_global(coarse, _defer=True)
            self.image_fine = _upload_beside(data['img'], dev)
            net.filter_local(self.image_fine[:, None])
        # the strip saved beside the mesh: input | front normal | back normal, as far as the coarse net predicts them; of an
        # RGB-D input (6 channels, SURVEY C-7) the colour half is shown
        tiles = [coarse[:, :3]] + [t for t in (net.netG.nmlF, net.netG.nmlB) if t is not None]
        self.strip = torch.cat(tiles, 0)
        self.b_min, self.b_max = data['b_min'], data['b_max']


def gen_mesh(res, net, cuda, data, save_path, thresh=0.5, use_octree=True, components=False, timings=None,
             sharded=False, result_on='rank0', _pipeline=True, _defer_host=True):
    """Generate the mesh of one subject.  `data` is the dict of `readData.py:106-116`.  Writes `<save_path>` and the debug
    PNG like the reference (`reconstruction.py:25-72`) and returns (verts float64 [V,3], faces int32 [F,3], colours float64
    [V,3]) - the arrays the reference hands to its writer.

    `sharded=True` (not in the reference, which is single-device): every rank of the initialised `torch.distributed` group
    calls gen_mesh with the SAME data; grid evaluation shards by whole chunks (`pifuhd_amd.dist`), the colour loop by whole
    calc_normal calls dealt out in rounds (`dist.colour_rounds`), rank 0 writes the files - streaming, as the single-GPU
    pipeline does.  `result_on='rank0'` (default): the other ranks copy nothing to the host and return None;
    `result_on='all'`: every rank returns the identical arrays."""
    from . import dist as pdist
    dev = net.device
    rank, ws = pdist.world() if sharded else (0, 1)
    in_group = sharded and pdist.collective_active()
    # below dist.SHARDED_MIN_RESOLUTION (the size guard) every rank computes the whole call by itself - no collective, the
    # same arrays everywhere; who writes the files and who returns the mesh stays as `result_on` says
    exchange = in_group and pdist.sharded_call(res)
    if result_on not in ('rank0', 'all'):
        raise ValueError("result_on must be 'rank0' or 'all'")
    wants_result = rank == 0 or result_on == 'all' or not in_group
    if exchange and timings is not None:
        pdist.stats_begin()
    # timings['profile_phases'] = True (bench.py, with pifu_profile_enable on): the executed GEMM FLOPs and the summed GEMM
    # kernel time are read at the phase boundaries - each read synchronises the device, so such a run is for the split, not
    # for the clock
    phases = bool(timings is not None and timings.pop('profile_phases', False))
    p0 = _prof_read(net, _lib.PROF_GEMM) if phases else None

    def mark():
        if timings is not None:
            # phase times are only meaningful once the queued kernels have finished; the COMPUTE stream is waited for, not
            # the copy stream, whose transfers are meant to run beside the next phase
            torch.cuda.current_stream(dev).synchronize()
        return time.perf_counter()
    t0 = mark()
    subj = _Subject(net, data, sharded=exchange)
    calib_tensor = subj.calib_host
    t1 = mark()
    png_thread = _save_debug_png(save_path[:-4] + '.png', subj.strip, background=True) if rank == 0 else None
    t2 = mark()

    # from here on the PNG thread reads a pinned buffer of this (device, thread), and the result arrays are views of pinned
    # slabs a copy-stream transfer may still be filling: whatever happens below - reconstruction() returning -1 (TypeError,
    # like the reference), an empty calc_normal slice, a failed collective - the thread is joined and the copies are waited
    # for before gen_mesh is left, so the next subject of a recon() loop cannot reuse memory that is still in flight
    host_copies = None
    try:
        # The colour loop as a pipeline.  It continues from the device vertices (cast to fp32 there exactly as
        # `torch.from_numpy(verts.T).unsqueeze(0).to(device=cuda).float()`, reconstruction.py:57, casts them); the
        # calc_normal calls go to the GPU in rounds of whole 50 000-vertex calls (the reference's GroupNorm batches,
        # `reconstruction.py:58-70`, several of them per kernel launch); each round's colours - after one all-gather over
        # the ranks when the loop is sharded - cross to pinned host memory on the copy stream as soon as they exist, and
        # the OBJ writer - which has every file offset from the positions alone and has put the face lines in place
        # already - formats and writes that round's vertex lines on its own threads while the GPU works on the next round.
        # The 200 MB of vertex / face arrays cross to the host on the copy stream - into the pinned slabs that ARE the
        # result arrays (`mesh_util.to_host_async`) - while the first round computes: `faces` is waited for once that round
        # is queued, `verts` before the OBJ layout needs the positions.
        # reconstruction() returns -1 when marching cubes fails; the tuple-unpack then raises TypeError exactly like the
        # reference (reconstruction.py:56)
        verts, faces, _, _, verts_dev, host_copies = reconstruction(net, cuda, calib_tensor, res, subj.b_min, subj.b_max,
                                                                    thresh, use_octree=use_octree, num_samples=5000,
                                                                    sharded=sharded, return_device_verts=True,
                                                                    _defer_host=True, _host_result=wants_result)
        if not _defer_host:       # A/B timing: the copies waited for at once, as before round 4
            host_copies.wait_all()
        t3 = mark()
        p3 = _prof_read(net, _lib.PROF_GEMM) if phases else None
        n_verts, n_faces = host_copies.n_verts, host_copies.n_faces
        spans = _normal_spans(n_verts)
        if any(r == l for (l, r) in spans):
            # an empty slice reaches Conv1d in the reference and raises (SURVEY C-6: V % 50000 in {0, 1})
            raise RuntimeError('calc_normal on an empty vertex slice')
        verts_tensor = verts_dev.T.unsqueeze(0).float().contiguous()
        interval = 50000

        def normals_of(left, right):
            # one chunked call per contiguous run of spans: the same GroupNorm batches as one calc_normal call per span
            net.calc_normal_chunked(verts_tensor[:, None, :, left:right], calib_tensor[:, None], calib_tensor, interval=interval)
            return (net.nmls.detach()[0] * 0.5 + 0.5).T          # fp32, like the reference's numpy expression

        n_col = spans[-1][1]                                      # the last vertex keeps colour 0 (SURVEY C-6)
        if wants_result:
            host_col = pinned_rows('colours', n_verts, torch.float32, dev)
            host_col[n_col:].zero_()
            host_np = host_col.numpy()
        side, cur = copy_stream(dev), torch.cuda.current_stream(dev)
        wsx = ws if exchange else 1
        rounds = pdist.colour_rounds(len(spans), wsx) if _pipeline else [(0, -(-len(spans) // wsx))]
        obj, done, laid_out = None, [], False
        color = np.zeros((n_verts, 3)) if wants_result else None  # the float64 colour array gen_mesh returns

        def drain(block):
            """Hand the rounds whose colours have landed to the OBJ writer (it formats and writes their vertex lines on its own
            threads while the GPU computes the next round) and to the float64 result; `block`: wait for every round."""
            nonlocal laid_out
            if obj is not None and not laid_out and (block or host_copies.verts is None or host_copies.verts.query()):
                host_copies.wait_verts()
                obj.layout(np.ascontiguousarray(verts, dtype=np.float64), host_np)      # every file offset follows from the positions
                laid_out = True
            while done and (block or done[0][2].query()):
                if obj is not None and not laid_out:
                    break
                L, R, ev = done.pop(0)
                ev.synchronize()
                if obj is not None:
                    obj.ready(R)                                  # this round's vertex lines are formatted and written now
                if color is not None:
                    color[L:R] = host_np[L:R]

        for (base, m) in rounds:
            mine, (L, R), slot = pdist.round_rows(spans, base, m, wsx, rank if exchange else 0, interval)
            col_d = normals_of(*mine).contiguous() if mine is not None else None
            if exchange:
                local = torch.zeros(slot, 3, device=dev, dtype=torch.float32)
                if col_d is not None:
                    local[:col_d.shape[0]] = col_d
                src = torch.empty(ws * slot, 3, device=dev, dtype=torch.float32)
            else:
                src = col_d
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                # the exchange is issued from the copy stream: the compute stream goes on with the next round's kernels, the
                # copy below waits for the collective (RCCL's stream is joined to the CURRENT stream, which is `side` here)
                if exchange:
                    pdist.all_gather_rows(src, local, dev)
                    local.record_stream(side)
                if wants_result:
                    host_col[L:R].copy_(src[:R - L], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            src.record_stream(side)
            done.append((L, R, ev))
            if obj is None and rank == 0:
                host_copies.wait_faces()
                obj = ObjFile(save_path, faces)                   # the face lines are formatted while the colours are computed
            # rounds that have landed meanwhile go to the writer now (with a transport whose collectives block the host -
            # gloo - this is where the file keeps up with the loop; with RCCL the loop is queued long before the first lands)
            drain(False)
        if wants_result:
            host_copies.wait_verts()
        drain(True)
        t4 = time.perf_counter()                                  # the last colours are on the host: the GPU phase is over
        if obj is not None:
            obj.finish()
        if wants_result:
            host_copies.wait_faces()
    finally:
        if host_copies is not None:
            host_copies.wait_all()
        if png_thread is not None:
            png_thread.join()
        exchange_stats = pdist.stats_end() if exchange and timings is not None else {}      # (also when the call failed)
    t5 = time.perf_counter()
    if timings is not None:
        timings.update(filter_s=t1 - t0, png_s=t2 - t1, reconstruction_s=t3 - t2, normals_s=t4 - t3, obj_s=t5 - t4,
                       total_s=t5 - t0, n_verts=n_verts, n_faces=n_faces, colour_rounds=len(rounds))
        timings.update(exchange_stats)
        if phases:
            p5 = _prof_read(net, _lib.PROF_GEMM)
            timings.update(gemm_flop_reconstruction=p3[2] - p0[2], gemm_ms_reconstruction=p3[0] - p0[0],
                           gemm_flop_normals=p5[2] - p3[2], gemm_ms_normals=p5[0] - p3[0])
    return (verts, faces, color) if wants_result else None


def _image_coloured_mesh(subj, res, net, cuda, data, save_path, thresh, use_octree):
    """Mesh with vertex colours looked up in the input image (`reconstruction.py:99-123`): 100 000-point query chunks,
    colour = bilinear sample of the fine image at the projected vertex, mapped from [-1, 1] to [0, 1]; vertices go back
    through `calib_world` when the dataset provides one.  Returns (verts, faces, colour)."""
    _save_debug_png(save_path[:-4] + '.png', subj.strip)
    verts, faces, _, _ = reconstruction(net, cuda, subj.calib, res, subj.b_min, subj.b_max, thresh, use_octree=use_octree,
                                        num_samples=100000)
    on_device = torch.from_numpy(verts.T).unsqueeze(0).to(device=net.device).float()
    uv = net.projection(on_device, subj.calib[:1])[:, :2, :]
    colour = index(subj.image_fine[:1], uv, net._ctx)[0].T.detach().cpu().numpy() * 0.5 + 0.5
    if 'calib_world' in data:
        to_world = inv(data['calib_world'].numpy()[0])
        verts = np.matmul(np.concatenate([verts, np.ones_like(verts[:, :1])], 1), to_world.T)[:, :3]
    return verts, faces, colour


def gen_mesh_imgColor(res, net, cuda, data, save_path, thresh=0.5, use_octree=True, components=False):
    """`gen_mesh_imgColor` (`reconstruction.py:77-125`).  Like the reference, a failure after the encoders is reported on
    stdout ('error' + the exception) instead of raised; the encoders themselves run outside that guard."""
    subj = _Subject(net, data)
    try:
        verts, faces, colour = _image_coloured_mesh(subj, res, net, cuda, data, save_path, thresh, use_octree)
        save_obj_mesh_with_color(save_path, verts, faces, colour)
        return verts, faces, colour
    except Exception as e:
        print('error')
        print(e)
        return None


# ------------------------------------------------------------------------------------------------
# Checkpoints and the recon() driver (reconstruction.py:232-321)
# ------------------------------------------------------------------------------------------------
def read_obj_with_color(path):
    """Parse an OBJ as `save_obj_mesh_with_color` writes it ('v x y z r g b' / 'f a b c', 1-based): returns
    (verts float64 [V,3], colours float64 [V,3], faces int32 [F,3] in file order)."""
    v, f = [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith('v '):
                v.append(line.split()[1:7])
            elif line.startswith('f '):
                f.append([t.split('/')[0] for t in line.split()[1:4]])
    va = np.asarray(v, dtype=np.float64).reshape(-1, 6)
    return va[:, :3].copy(), va[:, 3:6].copy(), np.asarray(f, dtype=np.int64).reshape(-1, 3).astype(np.int32) - 1


def merge_vertices(verts, colours, faces):
    """What `trimesh.load(obj_path)` does before `split()` sees the mesh (`reconstruction.py:332`; default `process=True`
    -> `merge_vertices`): referenced vertices with the same position become ONE vertex - the OBJ carries 4 decimals, so the
    marching-cubes vertices next to a grid node coincide after the round trip, hundreds of times per 512^3 mesh - the
    first occurrence survives with its colour, survivors keep their file order, unreferenced vertices go, faces are
    re-indexed and NOT filtered (triangles that become degenerate stay: `validate=False`).  trimesh is absent from this
    image: restated from its published behaviour (positions compared after rounding to 8 decimals), parity unpinned."""
    faces = np.asarray(faces)
    referenced = np.zeros(len(verts), dtype=bool)
    referenced[faces.reshape(-1)] = True
    ref = np.flatnonzero(referenced)
    keys = np.round(verts[ref] * 1e8).astype(np.int64)
    _, first, inverse = np.unique(keys, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')                     # unique rows come sorted by value: back to file order
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    remap = np.full(len(verts), -1, dtype=np.int64)
    remap[ref] = rank[np.asarray(inverse).reshape(-1)]
    kept = ref[first[order]]
    return verts[kept], colours[kept], remap[faces].astype(np.int32)


def largest_component(verts, faces, device):
    """Face indices (ascending) of the component `meshcleaning` keeps (`reconstruction.py:331-342`): of the WATERTIGHT
    edge-connected components (trimesh's `mesh.split()` default), the one with the largest extent along axis 0; the first
    one wins ties.  Components come from the device union-find `pifu_face_components` (faces adjacent iff they share an
    edge exactly two faces use).  Watertight is judged per component, as trimesh judges each sub-mesh: a face flagged
    for an edge nobody else uses opens its component; the faces around an edge that three or more faces share (they
    appear once vertices have been merged) are few and are counted per component here.  Raises IndexError when no
    component is watertight (`cc[0]` on an empty list in the reference)."""
    from .net_util import _ctx_cache
    from .nets import _ptr, _stream_ptr
    dev = torch.device(device)
    ctx = _ctx_cache(dev)
    fd = torch.from_numpy(np.ascontiguousarray(faces, dtype=np.int32)).to(ctx.device)
    label = torch.empty(len(faces), device=ctx.device, dtype=torch.int32)
    open_edge = torch.empty(len(faces), device=ctx.device, dtype=torch.uint8)
    _lib.call('pifu_face_components', ctx.h, _ptr(fd), len(faces), _ptr(label), _ptr(open_edge), _stream_ptr(ctx.device))
    label, flag = label.cpu().numpy(), open_edge.cpu().numpy()
    roots, inv = np.unique(label, return_inverse=True)            # ascending root = ascending first face: trimesh's order
    leaky = np.zeros(len(roots), bool)
    np.logical_or.at(leaky, inv, (flag & 1) != 0)
    shared = np.flatnonzero(flag & 2)                             # every face around a 3+-face edge carries the bit
    if len(shared):
        f = np.asarray(faces, dtype=np.int64)[shared]
        a = np.concatenate([f[:, 0], f[:, 1], f[:, 2]])
        b = np.concatenate([f[:, 1], f[:, 2], f[:, 0]])
        comp = np.tile(inv[shared], 3)
        rows = np.stack([np.minimum(a, b), np.maximum(a, b)], 1)
        _, e_inv, e_cnt = np.unique(rows, axis=0, return_inverse=True, return_counts=True)
        crowded = e_cnt[np.asarray(e_inv).reshape(-1)] >= 3       # the other edges of these faces are ordinary ones
        _, ec_inv, ec_cnt = np.unique(np.concatenate([rows[crowded], comp[crowded, None]], 1), axis=0, return_inverse=True,
                                      return_counts=True)
        odd = ec_cnt[np.asarray(ec_inv).reshape(-1)] != 2          # inside its component the edge does not pair up
        leaky[comp[crowded][odd]] = True
    x = verts[:, 0][faces]                                       # [F, 3]
    lo = np.full(len(roots), np.inf)
    hi = np.full(len(roots), -np.inf)
    np.minimum.at(lo, inv, x.min(1))
    np.maximum.at(hi, inv, x.max(1))
    best, best_extent = -1, None
    for c in np.flatnonzero(~leaky):
        extent = hi[c] - lo[c]
        if best < 0 or best_extent < extent:
            best, best_extent = c, extent
    if best < 0:
        raise IndexError('list index out of range')
    return np.flatnonzero(inv == best)


def meshcleaning(obj_path, device='cuda'):
    """`meshcleaning` (`reconstruction.py:325-344`): keep the connected component with the largest extent along axis 0
    and rewrite the file.  The OBJ is read back as written (4 decimals) and vertices at the same position are merged, as
    trimesh's loader does (`merge_vertices`); vertices no face of the kept component references are dropped (order kept),
    faces keep their file order.  Returns (verts, colours uint8 RGBA, faces) of the kept component; colours take trimesh's
    float -> uint8 route (round(c * 255), alpha 255) - trimesh is absent here, so its loader's behaviour is restated, not
    pinned."""
    print(f'Processing mesh cleaning: {obj_path}')
    verts, colours, faces = read_obj_with_color(obj_path)
    verts, colours, faces = merge_vertices(verts, colours, faces)
    keep = largest_component(verts, faces, device)
    used = np.unique(faces[keep])
    remap = np.full(len(verts), -1, np.int64)
    remap[used] = np.arange(len(used))
    verts, faces = verts[used], remap[faces[keep]].astype(np.int32)
    rgba = np.concatenate([np.clip(np.rint(colours[used] * 255.0), 0, 255), np.full((len(used), 1), 255.0)], 1).astype(np.uint8)
    _write_obj_plain(obj_path, verts, faces, rgba[:, :3] / 255.0)
    return verts, rgba, faces


def _write_obj_plain(path, verts, faces, colours):
    """An OBJ with the faces in the order given (the native writer swaps the last two indices, as
    `save_obj_mesh_with_color` does; a mesh read back from such a file is written back unswapped)."""
    save_obj_mesh_with_color(path, verts, faces[:, [0, 2, 1]], colours)


def esti_color(color, xyz_tensor):
    """`esti_color` (`reconstruction.py:180-229`): color uint16 [V,4], xyz_tensor [1,3,V] (projected vertices); returns the
    colour array with every back-side vertex (z < 0) recoloured from the silhouette.  Runs in `pifu_esti_color`."""
    from .net_util import _ctx_cache
    from .nets import _ptr, _stream_ptr
    xyz = xyz_tensor[0].to(torch.float32).contiguous()
    if not xyz.is_cuda:
        raise _lib.PifuError(_lib.ERR_INVALID, 'the HIP path needs device tensors (no CPU fallback)')
    ctx = _ctx_cache(xyz.device)
    # uint16 storage travels as int16 (torch has no arithmetic on uint16): same bits
    col = torch.from_numpy(np.ascontiguousarray(color, dtype=np.uint16).view(np.int16)).to(xyz.device)
    try:
        _lib.call('pifu_esti_color', ctx.h, _ptr(xyz), xyz.shape[1], _ptr(col), _stream_ptr(ctx.device))
    except _lib.PifuError as e:
        raise Exception(e.message)                      # the reference raises a bare Exception / IndexError here
    return col.cpu().numpy().view(np.uint16)


def gen_mesh_imgColor_plus(res, net, cuda, data, save_path, thresh=0.5, use_octree=True, components=False):
    """`gen_mesh_imgColor_plus` (`reconstruction.py:127-178`): the image-coloured mesh, reduced to its widest connected
    component (`meshcleaning`), back side recoloured from the silhouette (`esti_color`), written back to `save_path`.
    Failures after the encoders are printed ('error' + the exception), as in the reference."""
    subj = _Subject(net, data)
    try:
        verts, faces, colour = _image_coloured_mesh(subj, res, net, cuda, data, save_path, thresh, use_octree)
        save_obj_mesh_with_color(save_path, verts, faces, colour)
        verts, rgba, faces = meshcleaning(save_path, net.device)
        on_device = torch.from_numpy(verts.T).unsqueeze(0).to(device=net.device).float()
        rgba = np.array(esti_color(np.array(rgba, dtype='uint16'), net.projection(on_device, subj.calib[:1])), dtype='uint8')
        _write_obj_plain(save_path, verts, faces, rgba[:, :3] / 255.0)
        return verts, faces, rgba
    except Exception as e:
        print('error')
        print(e)
        return None


def load_checkpoint(path, map_location='cpu'):
    """The reference's checkpoint is `{'opt': Namespace, 'opt_netG': Namespace, 'model_state_dict': dict}` written with
    torch.save (`train.py`); `recon` reads it with a plain `torch.load` (`reconstruction.py:253-256`).  Loaded with the
    weights-only unpickler plus `argparse.Namespace`, so a checkpoint cannot execute code."""
    import argparse
    import types
    with torch.serialization.safe_globals([argparse.Namespace, types.SimpleNamespace]):
        return torch.load(path, map_location=map_location, weights_only=True)


def save_checkpoint(path, opt, opt_netG, netMR):
    """Counterpart of the reference's `torch.save` of a training state (same three keys)."""
    import argparse

    def ns(o):
        return o if isinstance(o, argparse.Namespace) else argparse.Namespace(**vars(o))
    torch.save({'opt': ns(opt), 'opt_netG': ns(opt_netG),
                'model_state_dict': {k: v.detach().cpu() for k, v in netMR.state_dict().items()}}, path)


def checkpoint_path(opt):
    """The file `recon` resumes from (`reconstruction.py:233-240`): an explicit `load_netMR_checkpoint_path` wins, else
    `<checkpoints_path>/<name>_train_latest` when `resume_epoch` is negative (it is then reset to 0, as the reference
    does), else `<checkpoints_path>/<name>_train_epoch_<resume_epoch>`."""
    if opt.load_netMR_checkpoint_path is not None:
        return opt.load_netMR_checkpoint_path
    if opt.resume_epoch < 0:
        opt.resume_epoch = 0
        return '%s/%s_train_latest' % (opt.checkpoints_path, opt.name)
    return '%s/%s_train_epoch_%d' % (opt.checkpoints_path, opt.name, opt.resume_epoch)


# what the command line keeps when the checkpoint's options replace the caller's (reconstruction.py:261-269)
_SESSION_OPTIONS = ('dataroot', 'resolution', 'results_path', 'loadSize')

_MESHERS = {0: 'gen_mesh', 1: 'gen_mesh_imgColor', 2: 'gen_mesh_imgColor_plus'}


def recon(opt, use_crop=False, use_gpu=True, dataset=None):
    """`recon` (`reconstruction.py:231-321`): load the checkpoint `checkpoint_path(opt)` names (an Exception
    'Failed loading state dict!' if it does not exist), continue with the CHECKPOINT's options except the session keys
    dataroot / resolution / results_path / loadSize, build both nets from it, put only netG in eval mode (SURVEY C-2:
    the fine filter keeps batch statistics) and write `<results_path>/<name>/recon/result_<subject>_<resolution>.obj`
    for subjects [start_id, end_id) of the dataset; `use_color` selects the colouring (0 normals, 1 image, 2 image +
    clean-up).  Returns the list of written paths.  `use_gpu=False` raises: this path has no CPU implementation;
    `dataset` replaces `readDataset(opt.dataroot, ...)`."""
    from .nets import PIFuMRNet, PIFuNetwNML
    from .readData import readDataset
    path = checkpoint_path(opt)
    if not use_gpu:
        raise NotImplementedError('the HIP path needs a GPU (no CPU fallback)')
    if path is None or not os.path.exists(path):
        raise Exception('Failed loading state dict!', path)
    device = torch.device('cuda:%d' % opt.gpu_id)
    first, last, colouring = opt.start_id, opt.end_id, opt.use_color
    ckpt = load_checkpoint(path)
    run = ckpt['opt']
    for key in _SESSION_OPTIONS:
        setattr(run, key, getattr(opt, key))
    print('checkpoint %s loaded; its options replace the command line except %s' % (path, ', '.join(_SESSION_OPTIONS)))

    subjects = dataset if dataset is not None else readDataset(run.dataroot, run.loadSize, use_crop=use_crop, device=device)
    print('%d subjects under %s' % (len(subjects), run.dataroot))
    netG = PIFuNetwNML(ckpt['opt_netG'], subjects.projection_mode, device=device)
    netMR = PIFuMRNet(run, netG, subjects.projection_mode)
    netMR.load_state_dict(ckpt['model_state_dict'])
    netG.eval()                                       # reconstruction.py:288-289,304: netMR itself stays in train mode

    out_dir = '{}/{}/recon'.format(run.results_path, run.name)
    for d in (run.checkpoints_path, run.results_path, out_dir):
        os.makedirs(d, exist_ok=True)
    written = []
    for i in range(max(first, 0), len(subjects) if last < 0 else min(last, len(subjects))):
        subject = subjects[i]
        target = out_dir + '/result_{}_{}.obj'.format(subject['name'], run.resolution)
        print(target)
        if colouring not in _MESHERS:
            raise NameError('unknown downsampling type')  # reconstruction.py:321 (the reference's wording)
        globals()[_MESHERS[colouring]](run.resolution, netMR, device, subject, target, components=run.use_compose)
        written.append(target)
    return written
