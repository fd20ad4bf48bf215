"""Two dense 128^3 lattice evaluations with the MLP GEMMs in the given precision mode (0 fp32, 1 bf16x3, 2 fp16x3): the workload
of the counter passes of tools/split_pmc.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pifuhd_amd  # noqa: F401
from pifuhd_amd import _lib, config, dist as pdist, nets, synth
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
g = torch.Generator().manual_seed(4242)
netG.set_features(torch.randn(1, 256, 128, 128, generator=g)); netMR.set_features(torch.randn(1, 16, 512, 512, generator=g))
data = synth.synthetic_inputs(1024, 6, seed=0)
h = netMR._ctx.h
_lib.call('pifu_reserve', h, 2097152)
_lib.call('pifu_set_precision', h, mode)
for _ in range(2):
    pdist.eval_lattice_sharded(netMR, data['calib'], (128, 128, 128), 5000, _lib.Q_FULL_COARSE)
torch.cuda.synchronize()
