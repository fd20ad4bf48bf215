"""Per-launch timeline of ONE steady-state encoder call from a rocprofv3 kernel trace of this script: every launch of
`filter_local` (default) or `filter_global` in issue order with its grid and duration, then totals by kernel.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/encoder_layers.py [local|global]
    [PACKS_BACK=2] python tools/encoder_layers.py --report OUT/*/*_kernel_trace.csv"""
import os, sys
if len(sys.argv) > 2 and sys.argv[1] == '--report':
    import csv, collections
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r['Start_Timestamp']))
    def short(n): return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    packs = [i for i, r in enumerate(rows) if short(r['Kernel_Name']).startswith('k_pack_prepared')]
    back = int(os.environ.get('PACKS_BACK', '1'))          # 2 with `global`: from the normal nets' packed input on
    seq = rows[packs[-back]:]
    t0 = int(seq[0]['Start_Timestamp'])
    acc = collections.Counter()
    for r in seq:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        n = short(r['Kernel_Name'])
        acc[n] += e - s
        print('%8.1f us  +%7.1f us  q%-2s %-22s grid %s x %s' % (s / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), n[:22],
                                                            r['Grid_Size_X'], r['Grid_Size_Y']))
    print('span %.2f ms' % ((int(seq[-1]['End_Timestamp']) - t0) / 1e6))
    for n, t in acc.most_common(): print('%-24s %8.2f ms' % (n, t / 1e6))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pifuhd_amd  # noqa: F401
from pifuhd_amd import config, nets, synth
which = sys.argv[1] if len(sys.argv) > 1 else 'local'
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
img, img512 = data['img'].cuda(), data['img_512'].cuda()
netMR.filter_global(img512); netMR.filter_local(img[:, None]); torch.cuda.synchronize()
for _ in range(2):
    if which == 'local':
        netMR.filter_local(img[:, None])
    else:
        netMR.filter_global(img512)
    torch.cuda.synchronize()
