"""Run gen_mesh 512^3 N times and compare sha256 of (verts, faces, colours) and of the OBJ bytes: the marching-cubes surface
list is filled in a run-dependent order and several host threads write the file, the results must not depend on either."""
import hashlib, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, reconstruction as recon
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
seen = set()
with tempfile.TemporaryDirectory() as td:
    for k in range(N):
        p = os.path.join(td, 'm.obj')
        v, f, c = recon.gen_mesh(512, netMR, 'cuda', data, p)
        h = hashlib.sha256()
        for a in (v, f, c):
            h.update(np.ascontiguousarray(a).tobytes())
        fh = hashlib.sha256(open(p, 'rb').read()).hexdigest()[:16]
        seen.add((h.hexdigest()[:16], fh))
        os.remove(p)
        print(k, h.hexdigest()[:16], fh, len(v), flush=True)
print('distinct results:', len(seen))
assert len(seen) == 1
