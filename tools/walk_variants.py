"""k_walk variants of the lab build side by side (PIFU_WALK_VARIANT, one process each): average launch time on the headline
lattice (HIP events of the library's PIFU_PROF_WALK slot) and a digest of the field - the variants differ in how the outputs
are dealt to threads, not in arithmetic, so the digests must be equal.    python tools/walk_variants.py [variants ...]"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'lab'))
    import ctypes as C
    import lab; lab.use()
    import torch
    from pifuhd_amd import config, nets, synth, _lib, dist as pdist
    oc = config.coarse_opt(); of = config.fine_opt(oc)
    spec = {}
    synth.mlp_keys('netG.mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
    synth.mlp_keys('mlp', of.mlp_dim, of.mlp_res_layers, of.mlp_norm, spec)
    sd = synth.name_seeded_state_dict(spec, seed=0, mlp_std=0.16)
    g = torch.Generator().manual_seed(4242)
    netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda'); netMR = nets.PIFuMRNet(of, netG); netG.eval()
    netG._sd = {k[5:]: t.cuda() for k, t in sd.items() if k.startswith('netG.')}; netG._push_weights()
    netMR._sd = {k: t.cuda() for k, t in sd.items() if not k.startswith('netG.')}; netMR._push_weights()
    netG.set_features(torch.randn(1, 256, 128, 128, generator=g)); netMR.set_features(torch.randn(1, 16, 512, 512, generator=g))
    calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
    h = netMR._ctx.h
    _lib.call('pifu_reserve', h, 2097152)
    f = pdist.eval_lattice_sharded(netMR, calib, (256, 256, 256), 5000, _lib.Q_FULL_COARSE)
    _lib.call('pifu_profile_read', h, _lib.PROF_WALK, None, None, None, 1)
    _lib.call('pifu_profile_enable', h, 1)
    for _ in range(3):
        f = pdist.eval_lattice_sharded(netMR, calib, (256, 256, 256), 5000, _lib.Q_FULL_COARSE)
    _lib.call('pifu_profile_enable', h, 0)
    ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
    _lib.call('pifu_profile_read', h, _lib.PROF_WALK, C.byref(ms), C.byref(n), C.byref(w), 1)
    print('variant %s: k_walk %.1f us per launch (%d launches), field sha %s' % (
        os.environ.get('PIFU_WALK_VARIANT', '0'), ms.value / max(n.value, 1) * 1e3, n.value,
        hashlib.sha256(f.cpu().numpy().tobytes()).hexdigest()[:16]), flush=True)
    sys.exit(0)
for v in (sys.argv[1:] or ['0', '9', '10', '11', '3', '0']):
    env = dict(os.environ, PIFU_WALK_VARIANT=v)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
