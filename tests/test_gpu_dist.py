"""World-size-2 run of the sharded gen_mesh on ONE GPU (two processes sharing cuda:0, gloo transport because
RCCL refuses two ranks on one device): grid evaluation sharded by chunks, colour loop sharded by calc_normal
calls, every rank must return the mesh a single-process gen_mesh returns, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _data():
    from pifuhd_amd import synth
    from helpers import calib_tensor
    return {'name': 't', 'img': synth.synthetic_image(128, 3, seed=3), 'img_512': synth.synthetic_image(64, 3, seed=2),
            'calib': calib_tensor(), 'b_min': [-1, -1, -1], 'b_max': [1, 1, 1]}


def _build():
    from pifuhd_amd import nets, synth
    from helpers import golden_nets
    sd, oc, of = golden_nets()
    of.loadSizeBig = 128
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = _data()
    netMR.filter_global(data['img_512']); netMR.filter_local(data['img'][:, None])
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3, resolution=32)
    return netMR, data


def _worker(rank, ws, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import reconstruction
    netMR, data = _build()
    path = os.path.join(outdir, f'mesh_r{rank}.obj')
    verts, faces, color = reconstruction.gen_mesh(64, netMR, 'cuda', data, path, sharded=True)
    np.savez(os.path.join(outdir, f'out_r{rank}.npz'), verts=verts, faces=faces, color=color)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_gen_mesh_sharded_world2_matches_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=800)
        assert p.exitcode == 0
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import reconstruction
    netMR, data = _build()
    path = str(tmp_path / 'single.obj')
    verts, faces, color = reconstruction.gen_mesh(64, netMR, 'cuda', data, path)
    for r in range(2):
        got = np.load(tmp_path / f'out_r{r}.npz')
        assert np.array_equal(got['verts'], verts) and np.array_equal(got['faces'], faces)
        assert np.array_equal(got['color'], color)
    # rank 0 writes the files, the other ranks do not
    assert open(tmp_path / 'mesh_r0.obj').read() == open(path).read()
    assert not os.path.exists(tmp_path / 'mesh_r1.obj')
