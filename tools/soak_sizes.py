"""Stability soak: gen_mesh called over and over with changing resolutions and subjects (as a recon() loop over a mixed data
set would): device memory (torch allocator + library arenas), host RSS and the pinned result pool must level off, the wall
time per size must not drift, and a repeated (subject, size) must reproduce its mesh."""
import hashlib, os, resource, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pifuhd_amd  # noqa: F401
from pifuhd_amd import _lib, config, mesh_util, nets, synth, reconstruction as recon
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
subjects = [synth.synthetic_inputs(1024, 6, seed=s) for s in range(3)]
netMR.filter_global(subjects[0]['img_512'].cuda()); netMR.filter_local(subjects[0]['img'].cuda()[:, None])
synth.calibrate_occupancy_bias(netMR, subjects[0]['calib'], sd, 0.3)
sizes = [128, 512, 256, 128, 384, 256]
seen, times = {}, {}
with tempfile.TemporaryDirectory() as td:
    for k in range(N):
        R, data = sizes[k % len(sizes)], subjects[(k // 2) % 3]
        p = os.path.join(td, 'm.obj')
        t0 = time.perf_counter()
        v, f, c = recon.gen_mesh(R, netMR, 'cuda', data, p)
        dt = time.perf_counter() - t0
        h = hashlib.sha256()
        for a in (v, f, c):
            h.update(np.ascontiguousarray(a).tobytes())
        key = (R, (k // 2) % 3)
        assert seen.setdefault(key, h.hexdigest()) == h.hexdigest(), key
        times.setdefault(key, []).append(dt)
        del v, f, c
        if k % 10 == 9 or k == N - 1:
            print(f'call {k + 1:4d}: torch allocated {torch.cuda.memory_allocated() / 2**30:6.2f} GiB reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB, '
                  f'library arenas {_lib.load().pifu_workspace_bytes(netMR._ctx.h) / 2**30:6.2f} GiB, result pool {mesh_util._pool_bytes() / 2**20:7.1f} MiB, '
                  f'host max RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:5.2f} GiB', flush=True)
for key, ts in sorted(times.items()):
    print(f'{key[0]:4d}^3 subject {key[1]}: first {ts[0]:.3f} s, median of the rest {np.median(ts[1:]) if len(ts) > 1 else float("nan"):.3f} s, last {ts[-1]:.3f} s ({len(ts)} calls)')
print('every repeated (size, subject) reproduced its mesh:', len(seen), 'distinct keys')
