"""`gen_mesh`: drop-in for the reference's `reconstruction.gen_mesh` (`reconstruction.py:25-72`).

Same signature, same side effects (`<save_path>` OBJ + `<save_path[:-4]>.png` with image | front normal |
back normal), same quirks (SURVEY C-6: the last vertex is never coloured because the final slice ends
at -1; a vertex count that is a multiple of 50 000 raises like the reference does).  The heavy lifting
(filters, octree grid evaluation, marching cubes, per-vertex normals) runs on the GPU through
`include/pifuhd_hip.h`.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from numpy.linalg import inv

from .mesh_util import reconstruction, save_obj_mesh_with_color
from .nets import index


def _save_debug_png(path, image_tensor_global, background=False):
    """Horizontal strip of the [-1,1] images, RGB (`reconstruction.py:36-54`).  The pixels are fetched here; with
    `background=True` the PNG encoding (zlib, GIL released) runs on a thread that the caller joins before returning."""
    try:
        from PIL import Image
    except Exception:          # pragma: no cover - PIL is optional plumbing
        return None
    strip = torch.cat([image_tensor_global[v, :3] for v in range(image_tensor_global.shape[0])], 2)      # [3, H, n*W]
    arr = (strip.detach().permute(1, 2, 0) * 0.5 + 0.5) * 255.0
    tiles = np.clip(np.rint(arr.cpu().numpy()), 0, 255).astype(np.uint8)

    def encode():
        Image.fromarray(tiles).save(path, compress_level=1)      # cv2.imwrite's default PNG compression level is 1 too
    if not background:
        encode()
        return None
    import threading
    th = threading.Thread(target=encode)
    th.start()
    return th


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
    png_thread = _save_debug_png(save_path[:-4] + '.png', image_tensor_global, background=True) if rank == 0 else None
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

    # one chunked call per contiguous run of spans: the same GroupNorm batches as one calc_normal call per span
    # (`reconstruction.py:58-70`), several of them per kernel launch
    def normals_of(left, right):
        net.calc_normal_chunked(verts_tensor[:, None, :, left:right], calib_tensor[:, None], calib_tensor, interval=50000)
        return (net.nmls.detach()[0] * 0.5 + 0.5).T          # fp32, like the reference's numpy expression

    if ws > 1:
        i0, i1, _ = pdist.shard_range(len(spans), ws, rank)
        mine = [(spans[i0][0], spans[i1 - 1][1])] if i1 > i0 else []
        runs = [(spans[j0][0], spans[j1 - 1][1]) for j0, j1 in
                (pdist.shard_range(len(spans), ws, rk)[:2] for rk in range(ws)) if j1 > j0]
        color32 = pdist.gather_runs(runs, mine, normals_of, 3, dev)
    else:
        color32 = normals_of(0, spans[-1][1])
    color = np.zeros(verts.shape)
    color[:color32.shape[0]] = color32.cpu().numpy()
    t4 = mark()

    if rank == 0:
        save_obj_mesh_with_color(save_path, verts, faces, color)
    if png_thread is not None:
        png_thread.join()
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


# ------------------------------------------------------------------------------------------------
# Checkpoints and the recon() driver (reconstruction.py:232-321)
# ------------------------------------------------------------------------------------------------
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


def recon(opt, use_crop=False, use_gpu=True, dataset=None):
    """`recon` (`reconstruction.py:231-321`): resolve the checkpoint path, overwrite `opt` with the checkpoint's (keeping
    dataroot / resolution / results_path / loadSize), build both nets, `netG.eval()` only, loop the dataset.
    `use_gpu=False` raises: this path has no CPU implementation.  `dataset` overrides `readDataset(opt.dataroot, ...)`."""
    from .nets import PIFuMRNet, PIFuNetwNML
    from .readData import readDataset
    if opt.load_netMR_checkpoint_path is not None:
        state_dict_path = opt.load_netMR_checkpoint_path
    elif opt.resume_epoch < 0:
        state_dict_path = '%s/%s_train_latest' % (opt.checkpoints_path, opt.name)
        opt.resume_epoch = 0
    else:
        state_dict_path = '%s/%s_train_epoch_%d' % (opt.checkpoints_path, opt.name, opt.resume_epoch)
    if not use_gpu:
        raise NotImplementedError('the HIP path needs a GPU (no CPU fallback)')
    cuda = torch.device('cuda:%d' % opt.gpu_id)
    start_id, end_id, use_color = opt.start_id, opt.end_id, opt.use_color

    if state_dict_path is not None and os.path.exists(state_dict_path):
        print('Resuming from', state_dict_path)
        state_dict = load_checkpoint(state_dict_path)
        print('Load {} success!'.format(state_dict_path))
        print('Warning: opt is overwritten.')
        dataroot, resolution, results_path, loadSize = opt.dataroot, opt.resolution, opt.results_path, opt.loadSize
        opt = state_dict['opt']
        opt.dataroot, opt.resolution, opt.results_path, opt.loadSize = dataroot, resolution, results_path, loadSize
        print('Options has been updated!')
    else:
        raise Exception('Failed loading state dict!', state_dict_path)

    print('Reading test images...')
    test_dataset = dataset if dataset is not None else readDataset(opt.dataroot, opt.loadSize, use_crop=use_crop, device=cuda)
    print('Read success!')
    print('num of test images:', len(test_dataset))
    projection_mode = test_dataset.projection_mode
    opt_netG = state_dict['opt_netG']
    netG = PIFuNetwNML(opt_netG, projection_mode, device=cuda)
    netMR = PIFuMRNet(opt, netG, projection_mode)
    netMR.load_state_dict(state_dict['model_state_dict'])

    results_save_path = '{}/{}/recon'.format(opt.results_path, opt.name)
    os.makedirs(opt.checkpoints_path, exist_ok=True)
    os.makedirs(opt.results_path, exist_ok=True)
    os.makedirs(results_save_path, exist_ok=True)
    start_id = 0 if start_id < 0 else start_id
    end_id = len(test_dataset) if end_id < 0 else end_id
    netG.eval()                                       # reconstruction.py:288-289,304: netMR stays in train mode
    written = []
    for i in range(start_id, end_id):
        if i >= len(test_dataset):
            break
        test_data = test_dataset[i]
        result_save_path = results_save_path + '/result_{}_{}.obj'.format(test_data['name'], opt.resolution)
        print(result_save_path)
        if use_color == 0:
            gen_mesh(opt.resolution, netMR, cuda, test_data, result_save_path, components=opt.use_compose)
        elif use_color == 1:
            gen_mesh_imgColor(opt.resolution, netMR, cuda, test_data, result_save_path, components=opt.use_compose)
        elif use_color == 2:
            raise NotImplementedError('gen_mesh_imgColor_plus (trimesh mesh cleaning + back-side recolouring) is not built')
        else:
            raise NameError('unknown downsampling type')      # reconstruction.py:321 (message as in the reference)
        written.append(result_save_path)
    return written
