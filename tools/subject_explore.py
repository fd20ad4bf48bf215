#!/usr/bin/env python3
"""What does a synthetic subject look like to gen_mesh?  For every configuration on the command line: vertices, faces,
connected components (watertight ones), points the octree evaluates, wall time of gen_mesh at the given resolution.

    python tools/subject_explore.py --res 512 group:0.16:texture:0.3 no:0.10:body:0.3:1.0 ...

A configuration is mlp_norm:mlp_std:image_kind:frac_inside[:feature_scale[:sharpen]].  feature_scale multiplies the last
1 x 1 convolution of both HGFilters (weights and bias): the image features the MLPs see shrink against the depth feature.
Used to choose the bench's body-like subject (round 6); not part of the product."""
import argparse
import ctypes as C
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import pifuhd_amd       # noqa: E402,F401
from pifuhd_amd import _lib, config, mesh_util, nets, reconstruction as recon, synth      # noqa: E402
from pifuhd_amd.nets import _ptr, _stream_ptr       # noqa: E402


def components(ctx, faces_np):
    f = torch.from_numpy(np.ascontiguousarray(faces_np, dtype=np.int32)).cuda()
    label = torch.empty(len(f), device='cuda', dtype=torch.int32)
    flag = torch.empty(len(f), device='cuda', dtype=torch.uint8)
    _lib.call('pifu_face_components', ctx.h, _ptr(f), len(f), _ptr(label), _ptr(flag), _stream_ptr(ctx.device))
    lab, fl = label.cpu().numpy(), flag.cpu().numpy()
    roots, inv, cnt = np.unique(lab, return_inverse=True, return_counts=True)
    leaky = np.zeros(len(roots), bool)
    np.logical_or.at(leaky, inv, (fl & 1) != 0)
    order = np.argsort(-cnt)
    return len(roots), int((~leaky).sum()), [(int(cnt[i]), bool(~leaky[i])) for i in order[:4]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--res', type=int, default=512)
    ap.add_argument('--channels', type=int, default=6)
    ap.add_argument('--norm', default=None, help="HGFilter norm ('batch' | 'group'); default: group for mlp_norm 'no', else batch")
    ap.add_argument('configs', nargs='+')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    for spec in args.configs:
        parts = spec.split(':')
        mlp_norm, mlp_std, kind, frac = parts[0], float(parts[1]), parts[2], float(parts[3])
        fscale = float(parts[4]) if len(parts) > 4 else 1.0
        sharp = float(parts[5]) if len(parts) > 5 else 1.0
        zoom = 1.0
        norm = args.norm or ('group' if mlp_norm == 'no' else 'batch')
        oc = config.coarse_opt(in_image_channels=args.channels, mlp_norm=mlp_norm, norm=norm)
        of = config.fine_opt(oc)
        sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=mlp_std)
        if fscale != 1.0:
            for k in list(sd):
                if ('image_filter.l' in k) and (k.endswith('.weight') or k.endswith('.bias')) and k.split('.')[-2].startswith('l'):
                    sd[k] = sd[k] * fscale
        if sharp != 1.0 and kind != 'crafted':          # the last layer of the fine MLP feeds the sigmoid alone: a steeper occupancy, the same level set
            last = 'mlp.filters.%d.weight' % (len(of.mlp_dim) - 2)
            sd[last] = sd[last] * sharp
        netG, netMR = nets.build_nets(oc, of, sd, dev)
        crafted = kind == 'crafted'
        data = synth.synthetic_inputs(1024, args.channels, seed=0, kind='body' if crafted else kind)
        if zoom != 1.0:
            data['calib'] = data['calib'].clone()
            data['calib'][0, :3, :3] *= zoom
        netMR.filter_global(data['img_512'].to(dev))
        netMR.filter_local(data['img'].to(dev)[:, None])
        if crafted:      # the MLPs rewritten so that the net predicts one smooth closed body (synth.craft_body_mlps); no bias calibration
            sd, st = synth.craft_body_mlps(sd, oc, of, netG.im_feat_list[-1], sharpness=100.0 * sharp)
            print('crafted:', st, flush=True)
            netMR.load_state_dict(sd)
            shift = 0.0
        else:
            shift = synth.calibrate_occupancy_bias(netMR, data['calib'], sd, frac)
        with tempfile.TemporaryDirectory() as td:
            try:
                recon.gen_mesh(args.res, netMR, dev, data, os.path.join(td, 'w.obj'))
                t = {}
                mesh = recon.gen_mesh(args.res, netMR, dev, data, os.path.join(td, 'm.obj'), timings=t)
                sdf, n_eval = mesh_util.eval_grid_device(netMR, data['calib'], args.res, 5000, True)
                inside = float((sdf > 0.5).double().mean().item())
                nc, nwt, top = components(netMR._ctx, mesh[1])
                v = mesh[0]
                print(f'{spec:36s} shift {shift:+8.3f} verts {t["n_verts"]:8d} faces {t["n_faces"]:8d} evaluated {n_eval:9d} '
                      f'({100.0 * n_eval / args.res ** 3:5.2f} %) inside {100 * inside:5.1f} % components {nc} (watertight {nwt}) top {top} '
                      f'bbox {v.min(0).round(2)} {v.max(0).round(2)} total {t["total_s"]:.4f} s (filter {t["filter_s"]:.4f} recon '
                      f'{t["reconstruction_s"]:.4f} normals {t["normals_s"]:.4f})', flush=True)
            except Exception as e:
                print(f'{spec:36s} shift {shift:+8.3f} FAILED: {e!r}', flush=True)
        del netG, netMR
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
