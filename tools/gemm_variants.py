"""A/B the k_mlp_gemm schedule variants in ONE process, interleaved rounds (cdna guide rule 24)."""
import ctypes as C, os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/lab'); import lab; lab.use()      # lab build: the PIFU_* switches live there
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, _lib, dist as pdist
variants = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else '0,1,2,3')]
res = int(sys.argv[2]) if len(sys.argv) > 2 else 128
oc = config.coarse_opt(); of = config.fine_opt(oc)
spec = {}
synth.mlp_keys('netG.mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
synth.mlp_keys('mlp', of.mlp_dim, of.mlp_res_layers, of.mlp_norm, spec)
sd = synth.name_seeded_state_dict(spec, seed=0, mlp_std=0.16)
g = torch.Generator().manual_seed(4242)
fc = torch.randn(1, 256, 128, 128, generator=g); ff = torch.randn(1, 16, 512, 512, generator=g)
calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
built = {}
for v in variants:
    os.environ['PIFU_GEMM_VARIANT'] = str(v % 100)
    os.environ['PIFU_PRECISION'] = '1' if v >= 100 else '0'      # 100 + variant = bf16x3 split precision
    netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda'); netMR = nets.PIFuMRNet(of, netG); netG.eval()
    netG._sd = {k[5:]: t.cuda() for k, t in sd.items() if k.startswith('netG.')}; netG._push_weights()
    netMR._sd = {k: t.cuda() for k, t in sd.items() if not k.startswith('netG.')}; netMR._push_weights()
    netG.set_features(fc); netMR.set_features(ff)
    _lib.call('pifu_reserve', netMR._ctx.h, 262144)
    built[v] = (netG, netMR)
R = (res, res, res)
outs = {}
for v in variants:
    outs[v] = pdist.eval_lattice_sharded(built[v][1], calib, R, 5000, _lib.Q_FULL_COARSE).clone()
torch.cuda.synchronize()
for v in variants[1:]:
    d = (outs[v] - outs[variants[0]]).abs().max().item()
    print(f'variant {v} vs {variants[0]}: max|d| = {d:.3e}  bitwise_equal={torch.equal(outs[v], outs[variants[0]])}')
stats = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        h = built[v][1]._ctx.h
        _lib.call('pifu_profile_read', h, None, None, None, 1); _lib.call('pifu_profile_enable', h, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pdist.eval_lattice_sharded(built[v][1], calib, R, 5000, _lib.Q_FULL_COARSE)
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        _lib.call('pifu_profile_read', h, C.byref(ms), C.byref(n), C.byref(fl), 1); _lib.call('pifu_profile_enable', h, 0)
        stats[v].append((fl.value / ms.value / 1e9, res ** 3 / wall / 1e6))
for v in variants:
    tf = sorted(s[0] for s in stats[v]); mq = sorted(s[1] for s in stats[v])
    print(f'variant {v}: gemm TF median {tf[len(tf)//2]:.1f} (min {tf[0]:.1f} max {tf[-1]:.1f});  Mq/s median {mq[len(mq)//2]:.2f}')
