"""`gen_mesh`: drop-in for the reference's `reconstruction.gen_mesh` (`reconstruction.py:25-72`).

Same signature, same side effects (`<save_path>` OBJ + `<save_path[:-4]>.png` with image | front normal |
back normal), same quirks (SURVEY C-6: the last vertex is never coloured because the final slice ends
at -1; a vertex count that is a multiple of 50 000 raises like the reference does).  The heavy lifting
(filters, octree grid evaluation, marching cubes, per-vertex normals) runs on the GPU through
`include/pifuhd_hip.h`.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from numpy.linalg import inv

from .mesh_util import reconstruction, save_obj_mesh_with_color
from .nets import index


def _save_debug_png(path, image_tensor_global):
    """Horizontal strip of the [-1,1] images, RGB (`reconstruction.py:36-54`)."""
    try:
        from PIL import Image
    except Exception:          # pragma: no cover - PIL is optional plumbing
        return False
    tiles = []
    for v in range(image_tensor_global.shape[0]):
        arr = np.transpose(image_tensor_global[v].detach().cpu().numpy(), (1, 2, 0)) * 0.5 + 0.5
        tiles.append(np.clip(np.rint(arr[:, :, :3] * 255.0), 0, 255).astype(np.uint8))
    Image.fromarray(np.concatenate(tiles, axis=1)).save(path)
    return True


def _normal_spans(n_verts, interval=50000):
    """The slices of the reference's colour loop (`reconstruction.py:58-70`): `interval`-vertex calls, the last
    one ending at -1, so the last vertex is never coloured (SURVEY C-6)."""
    spans = []
    for i in range(n_verts // interval + 1):
        left = i * interval
        right = n_verts - 1 if i == n_verts // interval else (i + 1) * interval
        spans.append((left, max(right, left)))
    return spans


def gen_mesh(res, net, cuda, data, save_path, thresh=0.5, use_octree=True, components=False, timings=None,
             sharded=False):
    """Generate the mesh of one subject.  `data` is the dict of `readData.py:106-116`.

    `sharded=True` (not in the reference, which is single-device): every rank of the initialised
    `torch.distributed` group calls gen_mesh with the SAME data; grid evaluation shards by whole chunks and the
    colour loop by whole calc_normal calls (`pifuhd_amd.dist`), every rank returns the identical mesh and
    rank 0 writes the files."""
    from . import dist as pdist
    dev = net.device
    rank, ws = pdist.world() if sharded else (0, 1)

    def mark():
        if timings is not None:
            torch.cuda.synchronize(dev)      # phase times are only meaningful once the queued kernels have finished
        return time.perf_counter()
    t0 = mark()
    image_tensor_global = data['img_512'].to(device=dev)
    image_tensor = data['img'].to(device=dev)
    calib_tensor = data['calib'].to(device=dev)

    net.filter_global(image_tensor_global)
    net.filter_local(image_tensor[:, None])
    t1 = mark()

    try:
        if net.netG.netF is not None:
            image_tensor_global = torch.cat([image_tensor_global, net.netG.nmlF], 0)
        if net.netG.netB is not None:
            image_tensor_global = torch.cat([image_tensor_global, net.netG.nmlB], 0)
    except Exception:
        pass
    b_min = data['b_min']
    b_max = data['b_max']
    if rank == 0:
        _save_debug_png(save_path[:-4] + '.png', image_tensor_global)
    t2 = mark()

    # reconstruction() returns -1 when marching cubes fails; the tuple-unpack then raises TypeError exactly
    # like the reference (reconstruction.py:56)
    verts, faces, _, _ = reconstruction(net, cuda, calib_tensor, res, b_min, b_max, thresh, use_octree=use_octree,
                                        num_samples=5000, sharded=sharded)
    t3 = mark()
    verts_tensor = torch.from_numpy(verts.T).unsqueeze(0).to(device=dev).float()

    spans = _normal_spans(len(verts))
    if any(r == l for (l, r) in spans):
        # an empty slice reaches Conv1d in the reference and raises (SURVEY C-6: V % 50000 in {0, 1})
        raise RuntimeError('calc_normal on an empty vertex slice')

    def normals_of(left, right):
        net.calc_normal(verts_tensor[:, None, :, left:right], calib_tensor[:, None], calib_tensor)
        return (net.nmls.detach()[0] * 0.5 + 0.5).T          # fp32, like the reference's numpy expression

    color32 = pdist.sharded_calls(spans, normals_of, 3, dev) if ws > 1 else None
    if color32 is None:
        color32 = torch.zeros(spans[-1][1], 3, device=dev, dtype=torch.float32)
        for (l, r) in spans:
            color32[l:r] = normals_of(l, r)
    color = np.zeros(verts.shape)
    color[:color32.shape[0]] = color32.cpu().numpy()
    t4 = mark()

    if rank == 0:
        save_obj_mesh_with_color(save_path, verts, faces, color)
    t5 = time.perf_counter()
    if timings is not None:
        timings.update(filter_s=t1 - t0, png_s=t2 - t1, reconstruction_s=t3 - t2, normals_s=t4 - t3, obj_s=t5 - t4,
                       total_s=t5 - t0, n_verts=len(verts), n_faces=len(faces))
    return verts, faces, color


def gen_mesh_imgColor(res, net, cuda, data, save_path, thresh=0.5, use_octree=True, components=False):
    """`gen_mesh_imgColor` (`reconstruction.py:77-125`): same mesh, 100 000-point query chunks, vertex colours sampled
    from the input image at the projected vertices, vertices mapped back through `calib_world`.  Like the reference,
    everything after the filters runs inside a try block that prints 'error' and the exception."""
    dev = net.device
    image_tensor_global = data['img_512'].to(device=dev)
    image_tensor = data['img'].to(device=dev)
    calib_tensor = data['calib'].to(device=dev)

    net.filter_global(image_tensor_global)
    net.filter_local(image_tensor[:, None])
    try:
        if net.netG.netF is not None:
            image_tensor_global = torch.cat([image_tensor_global, net.netG.nmlF], 0)
        if net.netG.netB is not None:
            image_tensor_global = torch.cat([image_tensor_global, net.netG.nmlB], 0)
    except Exception:
        pass
    b_min = data['b_min']
    b_max = data['b_max']
    try:
        _save_debug_png(save_path[:-4] + '.png', image_tensor_global)
        verts, faces, _, _ = reconstruction(net, cuda, calib_tensor, res, b_min, b_max, thresh, use_octree=use_octree,
                                            num_samples=100000)
        verts_tensor = torch.from_numpy(verts.T).unsqueeze(0).to(device=dev).float()
        xyz_tensor = net.projection(verts_tensor, calib_tensor[:1])
        uv = xyz_tensor[:, :2, :]
        color = index(image_tensor[:1], uv, net._ctx).detach().cpu().numpy()[0].T
        color = color * 0.5 + 0.5
        if 'calib_world' in data:
            calib_world = data['calib_world'].numpy()[0]
            verts = np.matmul(np.concatenate([verts, np.ones_like(verts[:, :1])], 1), inv(calib_world).T)[:, :3]
        save_obj_mesh_with_color(save_path, verts, faces, color)
        return verts, faces, color
    except Exception as e:
        print('error')
        print(e)
        return None
