mkdir -p gpurun_out
( timeout 300 python bench.py --steps 1 --warmup 1 --features synthetic --no-genmesh --cpu-chunks 40 2>&1 | tail -6 ) > gpurun_out/diag_cpu.log 2>&1
( timeout 600 python - <<'PY'
import time, torch, sys
sys.path.insert(0,'.')
import pifuhd_amd
from pifuhd_amd import config, nets, synth
t0=time.perf_counter()
oc=config.coarse_opt(); of=config.fine_opt(oc)
sd=synth.name_seeded_state_dict(synth.state_dict_spec(oc,of),seed=0,mlp_std=0.16)
netG,netMR=nets.build_nets(oc,of,sd,'cuda')
print('nets',time.perf_counter()-t0,flush=True)
data=synth.synthetic_inputs(1024,3,seed=0)
for it in range(2):
    t=time.perf_counter(); netMR.filter_global(data['img_512'].cuda()); torch.cuda.synchronize(); print('filter_global',it,time.perf_counter()-t,flush=True)
    t=time.perf_counter(); netMR.filter_local(data['img'].cuda()[:,None]); torch.cuda.synchronize(); print('filter_local',it,time.perf_counter()-t,flush=True)
PY
) > gpurun_out/diag_filter.log 2>&1
