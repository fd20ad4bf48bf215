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


class _Subject:
    """One subject on the device after both encoders ran: what every `gen_mesh*` variant of the reference starts with
    (`reconstruction.py:29-45` and its two copies at `:81-97`, `:131-147`)."""

    def __init__(self, net, data):
        dev = net.device
        self.calib = data['calib'].to(device=dev)
        self.image_fine = data['img'].to(device=dev)
        coarse = data['img_512'].to(device=dev)
        net.filter_global(coarse)
        net.filter_local(self.image_fine[:, None])
        # the strip saved beside the mesh: input | front normal | back normal, as far as the coarse net predicts them
        tiles = [coarse] + [t for t in (net.netG.nmlF, net.netG.nmlB) if t is not None]
        self.strip = torch.cat(tiles, 0)
        self.b_min, self.b_max = data['b_min'], data['b_max']


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
    exchange = sharded and pdist.collective_active()

    def mark():
        if timings is not None:
            torch.cuda.synchronize(dev)      # phase times are only meaningful once the queued kernels have finished
        return time.perf_counter()
    t0 = mark()
    subj = _Subject(net, data)
    calib_tensor = subj.calib
    t1 = mark()
    png_thread = _save_debug_png(save_path[:-4] + '.png', subj.strip, background=True) if rank == 0 else None
    t2 = mark()

    # reconstruction() returns -1 when marching cubes fails; the tuple-unpack then raises TypeError exactly
    # like the reference (reconstruction.py:56)
    verts, faces, _, _ = reconstruction(net, cuda, calib_tensor, res, subj.b_min, subj.b_max, thresh, use_octree=use_octree,
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

    if exchange:
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
def gen_mesh_imgColor_plus(res, net, cuda, data, save_path, thresh=0.5, use_octree=True, components=False):
    """`gen_mesh_imgColor_plus` (`reconstruction.py:127-178`): the image-coloured mesh, reduced to its widest connected
    component and with the back side recoloured from the silhouette."""
    raise NotImplementedError('gen_mesh_imgColor_plus (mesh cleaning + back-side recolouring) is not built yet')


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
