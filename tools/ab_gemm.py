"""A/B of GEMM kernel variants in separate processes: same lattice evaluation, sha256 of the field + throughput.
usage: python tools/ab_gemm.py <variant> [res]"""
import hashlib, os, sys, time
os.environ['PIFU_GEMM_VARIANT'] = sys.argv[1]
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tools', 'lab')); import lab; lab.use()      # lab build: the PIFU_* switches live there
import torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, _lib, dist as pdist
R = int(sys.argv[2]) if len(sys.argv) > 2 else 128
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
_lib.call('pifu_reserve', netMR._ctx.h, 262144)
for flags, tag in ((_lib.Q_FULL_COARSE, 'col'), (_lib.Q_FULL_COARSE | _lib.Q_NO_COLUMN_SHARING, 'nocol')):
    f = pdist.eval_lattice_sharded(netMR, calib, (R, R, R), 5000, flags)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = pdist.eval_lattice_sharded(netMR, calib, (R, R, R), 5000, flags)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('variant', sys.argv[1], tag, 'sha', hashlib.sha256(f.cpu().numpy().tobytes()).hexdigest()[:16], '%.2f Mq/s' % (R ** 3 / dt / 1e6), flush=True)
# ragged single chunk through pifu_query (explicit points)
pts = torch.rand(1, 3, 4321, generator=g) * 2 - 1
netMR.query(pts.cuda(), calib.cuda())
print('variant', sys.argv[1], 'query sha', hashlib.sha256(netMR.get_preds().cpu().numpy().tobytes()).hexdigest()[:16])
