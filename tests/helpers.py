"""Seed-derived inputs shared by the golden generator (tests/golden/make_golden.py) and the tests."""
import numpy as np
import torch

import pifuhd_amd  # noqa: F401
from pifuhd_amd import config, synth


def seeded_features(seed, hc=32, hf=64):
    g = torch.Generator().manual_seed(1000 + seed)
    fc = torch.randn(1, 256, hc, hc, generator=g)
    ff = torch.randn(1, 16, hf, hf, generator=g)
    return fc, ff


def seeded_points(seed, n, lim=1.15):
    g = torch.Generator().manual_seed(2000 + seed)
    return (torch.rand(1, 3, n, generator=g) * 2 - 1) * lim


def calib_tensor():
    c = torch.eye(4)
    c[1, 1] = -1.0
    return c[None]


_GOLDEN_NETS = {}


def golden_nets(mlp_norm='group', seed=1, mlp_std=0.16, perturb=True, norm='batch'):
    """Options + name-seeded state dict exactly as make_golden.build_reference() loads them.  Generating the ~600 seeded tensors
    takes seconds and most tests ask for the same few sets: kept per process; every caller gets its own dict and its own option
    objects (tests replace entries and set option fields), the tensors themselves are shared and never written in place."""
    import copy
    from collections import OrderedDict
    if mlp_norm not in ('group', 'batch') and mlp_std is not None:
        mlp_std = 0.04
    key = (mlp_norm, seed, mlp_std, perturb, norm)
    if key not in _GOLDEN_NETS:
        oc = config.coarse_opt(mlp_norm=mlp_norm, norm=norm)
        of = config.fine_opt(oc)
        sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=seed, mlp_std=mlp_std,
                                          perturb=perturb)
        _GOLDEN_NETS[key] = (sd, oc, of)
    sd, oc, of = _GOLDEN_NETS[key]
    return OrderedDict(sd), copy.deepcopy(oc), copy.deepcopy(of)


def noisy_field(points):
    """The deterministic analytic field used by make_golden.golden_octree()."""
    p = np.asarray(points).astype(np.float32).astype(np.float64)
    r = np.sqrt((0.8 * p[0]) ** 2 + (0.45 * p[1]) ** 2 + (1.6 * p[2]) ** 2)
    v = 1.0 / (1.0 + np.exp(-(0.35 - r) / 0.05))
    v = v + 0.02 * np.sin(37.0 * p[0] + 11.0 * p[1]) * np.cos(23.0 * p[2])
    return v.astype(np.float32)


def mlp_only_nets(seed=3, mlp_std=0.16):
    """Options + a state dict holding only the two SurfaceClassifier MLPs (fast to generate); enough for
    query() when the feature maps are installed directly."""
    oc = config.coarse_opt()
    of = config.fine_opt(oc)
    spec = {}
    synth.mlp_keys('netG.mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
    synth.mlp_keys('mlp', of.mlp_dim, of.mlp_res_layers, of.mlp_norm, spec)
    return synth.name_seeded_state_dict(spec, seed=seed, mlp_std=mlp_std, perturb=True), oc, of


def calib_variant(tag):
    """Non-trivial calibration matrices [1,4,4] (float32) for the parity tests.  'scale_shift': zoom 1.25 with a
    translation - the projected (x, y) of a lattice column still does not depend on k, so column sharing stays on and the
    translation term of `orthogonal` is exercised.  'rot_y': 20 degrees about the y axis with zoom 1.1 - x depends on z,
    so lattice sources take the per-point path."""
    import math
    c = torch.eye(4)
    if tag == 'scale_shift':
        c[0, 0], c[1, 1], c[2, 2] = 1.25, -1.25, 1.25
        c[0, 3], c[1, 3], c[2, 3] = 0.1, -0.05, 0.07
    elif tag == 'rot_y':
        a, zoom = math.radians(20.0), 1.1
        c[0, 0], c[0, 2] = zoom * math.cos(a), zoom * math.sin(a)
        c[1, 1] = -zoom
        c[2, 0], c[2, 2] = -zoom * math.sin(a), zoom * math.cos(a)
        c[0, 3], c[2, 3] = -0.03, 0.02
    else:
        raise KeyError(tag)
    return c[None]


def free_port():
    """A free TCP port on 127.0.0.1 BELOW the kernel's ephemeral range: a port probed with bind(0) can be handed out again - as the
    source port of an outgoing connection - before the rendezvous store binds it (seen once on the GPU box: EADDRINUSE)."""
    import os
    import random
    import socket
    import time
    rng = random.Random(os.getpid() ^ int(time.time() * 1e3))
    for _ in range(64):
        cand = rng.randrange(15000, 30000)
        s = socket.socket()
        try:
            s.bind(('127.0.0.1', cand))
            return cand
        except OSError:
            continue
        finally:
            s.close()
    raise RuntimeError('no free port in 15000-29999')
