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


def _build(mlp_norm='group'):
    from pifuhd_amd import nets, synth
    from helpers import golden_nets
    sd, oc, of = golden_nets(mlp_norm=mlp_norm)
    of.loadSizeBig = 128
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = _data()
    netMR.filter_global(data['img_512']); netMR.filter_local(data['img'][:, None])
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3, resolution=32)
    return netMR, data


def _pointwise_worker(rank, ws, port, outdir):
    """The un-normalised MLP of the published pifuhd.pt: the octree levels are dealt out in the library's own groups (16 384
    rows, no chunk padding) instead of the reference's 5 000-point chunks."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import dist as pdist, reconstruction
    pdist.SHARDED_MIN_RESOLUTION = 0                   # the 64^3 calls of this test are to exchange, not to fall under the size guard
    netMR, data = _build('no')
    assert pdist.effective_chunk(netMR._ctx, 5000, 0) == 16384
    field = pdist.eval_lattice_sharded(netMR, data['calib'], (48, 64, 64), 5000)
    sdf, n_eval = pdist.eval_grid_octree_sharded(netMR, data['calib'], 64, 5000, init_resolution=16)
    mesh = reconstruction.gen_mesh(64, netMR, 'cuda', data, os.path.join(outdir, f'pw_r{rank}.obj'), sharded=True, result_on='all')
    np.savez(os.path.join(outdir, f'pw_r{rank}.npz'), field=field.cpu().numpy(), sdf=sdf.cpu().numpy(), n_eval=np.array(n_eval),
             verts=mesh[0], faces=mesh[1], color=mesh[2])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_pointwise_mlp_sharded_matches_single_process(tmp_path):
    """Two ranks (gloo, one GPU) in the pifuhd.pt configuration (`mlp_norm='no'`): dense lattice in raw Z-slabs, octree levels
    in 16 384-row groups, gen_mesh end to end - every rank holds the single-process result bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import free_port; port = free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_pointwise_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=800)
        assert p.exitcode == 0
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import dist as pdist, mesh_util, reconstruction
    netMR, data = _build('no')
    field = pdist.eval_lattice_sharded(netMR, data['calib'], (48, 64, 64), 5000).cpu().numpy()
    sdf, n_eval = mesh_util.eval_grid_device(netMR, data['calib'], 64, 5000, True, init_resolution=16)
    mesh = reconstruction.gen_mesh(64, netMR, 'cuda', data, str(tmp_path / 'single.obj'))
    for r in range(2):
        got = np.load(tmp_path / f'pw_r{r}.npz')
        assert np.array_equal(got['field'], field)
        assert np.array_equal(got['sdf'], sdf.cpu().numpy()) and int(got['n_eval']) == n_eval
        assert all(np.array_equal(got[k], a) for k, a in zip(('verts', 'faces', 'color'), mesh))
    assert open(tmp_path / 'pw_r0.obj').read() == open(tmp_path / 'single.obj').read()


def _worker(rank, ws, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import dist as pdist0, reconstruction
    pdist0.SHARDED_MIN_RESOLUTION = 0                  # the 64^3 calls of this test are to exchange, not to fall under the size guard
    netMR, data = _build()
    path = os.path.join(outdir, f'mesh_r{rank}.obj')
    verts, faces, color = reconstruction.gen_mesh(64, netMR, 'cuda', data, path, sharded=True, result_on='all')
    # the default: only rank 0 copies the mesh to the host and returns it (and writes the same file again)
    timings = {}
    only0 = reconstruction.gen_mesh(64, netMR, 'cuda', data, path, sharded=True, timings=timings)
    assert (only0 is None) == (rank != 0)
    if rank == 0:
        assert all(np.array_equal(a, b) for a, b in zip(only0, (verts, faces, color)))
    # the exchange explains itself: collectives, their bytes and the time inside them (bench.py's N > 1 line)
    assert timings['collectives'] >= 2 and timings['collective_bytes'] > 0 and timings['collective_s'] >= 0.0
    assert timings['colour_rounds'] >= 1 and timings['n_verts'] == len(verts)
    # the same call with marching cubes over one slab per rank (the form `reconstruction` switches to at R >= 1024)
    from pifuhd_amd import dist as pdist
    pdist.SLAB_MC_MIN_RESOLUTION = 64
    verts2, faces2, color2 = reconstruction.gen_mesh(64, netMR, 'cuda', data, path, sharded=True, result_on='all')
    # the size guard: below dist.SHARDED_MIN_RESOLUTION (128 by default) every rank computes the call alone - no collective,
    # rank 0 writes and returns, the arrays are the sharded call's
    pdist0.SHARDED_MIN_RESOLUTION = 128
    tg = {}
    guarded = reconstruction.gen_mesh(64, netMR, 'cuda', data, path, sharded=True, timings=tg)
    assert 'collectives' not in tg and (guarded is None) == (rank != 0)
    if rank == 0:
        assert all(np.array_equal(a, b) for a, b in zip(guarded, (verts, faces, color)))
    pdist0.SHARDED_MIN_RESOLUTION = 0
    # the encoders of the sharded call were dealt out over the two ranks (dist.filter_sharded: netF + fine hourglass on rank 0,
    # netB + coarse hourglass on rank 1, maps broadcast): what every rank holds afterwards
    g = netMR.netG
    np.savez(os.path.join(outdir, f'out_r{rank}.npz'), verts=verts, faces=faces, color=color, verts2=verts2, faces2=faces2,
             color2=color2, nmlF=g.nmlF.cpu().numpy(), nmlB=g.nmlB.cpu().numpy(), feat_coarse=g.im_feat_list[-1].cpu().numpy(),
             feat_fine=netMR.im_feat_list[-1].cpu().numpy(), filter_collectives=np.array(timings['collectives']))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('ws', [2, 3])
def test_gen_mesh_sharded_world2_matches_single_process(tmp_path, ws):
    """Two and three ranks (three: a rank beyond the two that run the dealt-out encoders only receives their maps; the colour
    rounds deal 50 000-vertex calls to ranks that may end up without one)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import free_port; port = free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, ws, port, str(tmp_path))) for r in range(ws)]
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
    g = netMR.netG
    for r in range(ws):
        got = np.load(tmp_path / f'out_r{r}.npz')
        # the dealt-out encoders leave bit-identical maps on every rank (and 4 broadcasts among the call's collectives)
        for key, t in (('nmlF', g.nmlF), ('nmlB', g.nmlB), ('feat_coarse', g.im_feat_list[-1]), ('feat_fine', netMR.im_feat_list[-1])):
            assert np.array_equal(got[key], t.cpu().numpy()), (r, key)
        assert int(got['filter_collectives']) >= 4 + 2
        assert np.array_equal(got['verts'], verts) and np.array_equal(got['faces'], faces)
        assert np.array_equal(got['color'], color)
        assert np.array_equal(got['verts2'], verts) and np.array_equal(got['faces2'], faces)
        assert np.array_equal(got['color2'], color)
    # rank 0 writes the files, the other ranks do not
    assert open(tmp_path / 'mesh_r0.obj').read() == open(path).read()
    assert not any(os.path.exists(tmp_path / f'mesh_r{r}.obj') for r in range(1, ws))


def _rccl_worker(port, outdir):
    """One rank, backend nccl (= RCCL): the sharded entry points with their collectives forced on must reproduce the
    plain single-process results bit for bit."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import dist as pdist, reconstruction
    pdist.SHARDED_MIN_RESOLUTION = 0
    netMR, data = _build()
    want_field = pdist.eval_lattice_sharded(netMR, data['calib'], (40, 32, 48), 5000)
    want_oct, n_want = pdist.eval_grid_octree_sharded(netMR, data['calib'], 64, 5000, init_resolution=16)
    want = reconstruction.gen_mesh(64, netMR, 'cuda', data, os.path.join(outdir, 'plain.obj'))
    pdist.ALWAYS_COLLECTIVE = True
    assert pdist.collective_active()
    got_field = pdist.eval_lattice_sharded(netMR, data['calib'], (40, 32, 48), 5000)
    got_oct, n_got = pdist.eval_grid_octree_sharded(netMR, data['calib'], 64, 5000, init_resolution=16)
    got = reconstruction.gen_mesh(64, netMR, 'cuda', data, os.path.join(outdir, 'rccl.obj'), sharded=True, result_on='all')
    pdist.SLAB_MC_MIN_RESOLUTION = 64                  # marching cubes through its slab form: four more all-gathers
    got_slab = reconstruction.gen_mesh(64, netMR, 'cuda', data, os.path.join(outdir, 'rccl_slab.obj'), sharded=True, result_on='all')
    # the same collectives through the C ABI's own RCCL communicator (pifu_comm_* / pifu_allgather_*: what a binding without
    # torch.distributed calls); torch.distributed only carries the communicator id
    pdist.SLAB_MC_MIN_RESOLUTION = 1024
    pdist.use_native_comm('cuda')                      # no index given: resolved to the current device (round-5 advisor)
    # proof that the C exchange carries the data: while the native communicator is active torch.distributed's all-gather
    # must not be reached at all, and the communicator counts its own calls
    torch_gather = dist.all_gather_into_tensor

    def refuse(*a, **k):
        raise AssertionError('torch.distributed.all_gather_into_tensor reached while the native communicator is active')
    dist.all_gather_into_tensor = refuse
    try:
        nat_field = pdist.eval_lattice_sharded(netMR, data['calib'], (40, 32, 48), 5000)
        calls_field = pdist.native_comm_stats()[0]
        nat = reconstruction.gen_mesh(64, netMR, 'cuda', data, os.path.join(outdir, 'native.obj'), sharded=True, result_on='all')
        calls_mesh = pdist.native_comm_stats()[0]
        pdist.SLAB_MC_MIN_RESOLUTION = 64              # float64 / int32 all-gathers go through pifu_allgather_bytes
        nat_slab = reconstruction.gen_mesh(64, netMR, 'cuda', data, os.path.join(outdir, 'native_slab.obj'), sharded=True, result_on='all')
        calls_slab, bytes_all = pdist.native_comm_stats()
    finally:
        dist.all_gather_into_tensor = torch_gather
    pdist.release_native_comm()
    # one all-gather for the dense field; per gen_mesh at least one octree level + one colour round; four more for the slab
    # form of marching cubes (counts, halo planes, vertices, faces)
    native_ran = calls_field == 1 and calls_mesh >= calls_field + 2 and calls_slab >= calls_mesh + (calls_mesh - calls_field) + 4 \
        and bytes_all > 0
    print('native communicator calls: field %d, + gen_mesh %d, + slab gen_mesh %d, %d bytes gathered' %
          (calls_field, calls_mesh, calls_slab, bytes_all), flush=True)
    native_ok = (native_ran and torch.equal(want_field, nat_field) and all(np.array_equal(a, b) for a, b in zip(want, nat))
                 and all(np.array_equal(a, b) for a, b in zip(want, nat_slab))
                 and open(os.path.join(outdir, 'plain.obj')).read() == open(os.path.join(outdir, 'native.obj')).read())
    t = torch.tensor([3.0], device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    ok = (native_ok and torch.equal(want_field, got_field) and torch.equal(want_oct, got_oct) and n_want == n_got
          and all(np.array_equal(a, b) for a, b in zip(want, got)) and float(t.item()) == 3.0
          and all(np.array_equal(a, b) for a, b in zip(want, got_slab))
          and open(os.path.join(outdir, 'plain.obj')).read() == open(os.path.join(outdir, 'rccl.obj')).read())
    with open(os.path.join(outdir, 'rccl_ok.txt'), 'w') as f:
        f.write('%s %s' % (ok, dist.get_backend()))
    if not ok:
        print('rccl worker: native_ran', native_ran, 'native_ok', native_ok, 'calls', calls_field, calls_mesh, calls_slab, flush=True)
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_rccl_transport_single_rank(tmp_path):
    """The `nccl` (RCCL) branch of the sharded path, on the one GPU this box has: process group of one rank, every
    collective of the path executed (all_gather_into_tensor of the field and of the colours, all_reduce, barrier), results
    identical to the collective-free run.  No scaling claim - only that the transport initialises, runs and shuts down."""
    from helpers import free_port; port = free_port()
    ctx = mp.get_context('spawn')
    p = ctx.Process(target=_rccl_worker, args=(port, str(tmp_path)))
    p.start()
    p.join(timeout=800)
    assert p.exitcode == 0
    assert open(tmp_path / 'rccl_ok.txt').read() == 'True nccl'


@pytest.mark.timeout(900)
def test_bench_dist_selftest_under_torchrun(tmp_path):
    """bench.py exactly as the driver launches it for N GPUs (python -m torch.distributed.run ... bench.py --gpus N), with
    N = 1 and --dist-selftest so that the process group (backend nccl) is created and the N > 1 code path runs."""
    import json
    import subprocess
    from helpers import free_port; port = free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), 'bench.py', '--gpus', '1', '--steps', '1', '--warmup', '1', '--res', '64',
           '--genmesh-res', '64', '--batch-rows', '262144', '--dist-selftest', '--no-cpu-baseline', '--no-split-precision',
           '--no-lean-pass', '--no-canonical', '--no-smooth-genmesh', '--config5-res', '64']
    env = dict(os.environ)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['dist_selftest']['backend'] == 'nccl' and line['n_gpus'] == 1 and line['value'] > 0
    # what RCCL reported about the communicator it built (NCCL_DEBUG=INFO into a per-rank file, summarised once)
    assert line['rccl']['lines'] and any('nChannels' in ln or 'Ring' in ln for ln in line['rccl']['lines'])
    assert 'error' not in line['gen_mesh'].get('one_subject_sharded', {'error': 1})


def _mc_volume(shape, seed):
    rs = np.random.RandomState(seed)
    lin = [-1 + 2 * np.arange(n) / n for n in shape]
    X, Y, Z = np.meshgrid(*lin, indexing='ij')
    return (1.0 / (1.0 + np.exp(-(0.55 - np.sqrt(X ** 2 + (0.8 * Y) ** 2 + (1.2 * Z) ** 2)) / 0.04))
            + 0.25 * (rs.rand(*shape) - 0.5))


_MC_CASES = [((61, 40, 36), 1, 0.5), ((3, 9, 9), 2, 0.5), ((2, 6, 6), 3, 0.5), ((24, 16, 16), 4, 0.45)]


def _mc_worker(rank, ws, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import dist as pdist
    from pifuhd_amd.nets import _Ctx
    ctx = _Ctx('cuda')
    out = {}
    for i, (shape, seed, level) in enumerate(_MC_CASES):
        vol = torch.from_numpy(_mc_volume(shape, seed)).cuda()
        v, f = pdist.marching_cubes_sharded(ctx, vol, level)
        out[f'v{i}'], out[f'f{i}'] = v.cpu().numpy(), f.cpu().numpy()
    # both exceptions must come up on every rank (no rank may be left waiting in a collective)
    for vol, exc in ((torch.zeros(9, 4, 4, device='cuda'), ValueError), (torch.full((9, 4, 4), 0.5, device='cuda'), RuntimeError)):
        try:
            pdist.marching_cubes_sharded(ctx, vol, 0.5)
            out['raised'] = np.array(0)
            break
        except exc:
            out['raised'] = np.array(1)
    np.savez(os.path.join(outdir, f'mc_r{rank}.npz'), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_slab_marching_cubes_world3_matches_the_replicated_run(tmp_path):
    """One axis-0 slab per rank, three ranks sharing the GPU over gloo: every rank must end with the vertices and faces
    of the single-process marching cubes, bit for bit - also when a rank is left without a cell layer (volumes with 2
    and 1 layers) - and skimage's exceptions must be raised on all ranks."""
    from helpers import free_port; port = free_port()
    mpctx = mp.get_context('spawn')
    procs = [mpctx.Process(target=_mc_worker, args=(r, 3, port, str(tmp_path))) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=800)
        assert p.exitcode == 0
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import mesh_util
    from pifuhd_amd.nets import _Ctx
    ctx = _Ctx('cuda')
    for i, (shape, seed, level) in enumerate(_MC_CASES):
        v, f = mesh_util.marching_cubes_device(ctx, torch.from_numpy(_mc_volume(shape, seed)).cuda(), level)
        v, f = v.cpu().numpy(), f.cpu().numpy()
        assert len(v) > 0
        for r in range(3):
            got = np.load(tmp_path / f'mc_r{r}.npz')
            assert np.array_equal(got[f'f{i}'], f) and np.array_equal(got[f'v{i}'].view(np.uint32), v.view(np.uint32)), (i, r)
            assert int(got['raised']) == 1


@pytest.mark.timeout(900)
@pytest.mark.parametrize('launcher', ['torchrun', 'plain'])
def test_bench_two_ranks_on_one_gpu(launcher):
    """The N > 1 form of bench.py, launched the way the driver launches it (`python -m torch.distributed.run --nproc-per-node
    2 ... bench.py --gpus 2`) and PLAINLY (`python bench.py --gpus 2`: bench.py starts its ranks itself, as a fresh child,
    before it touches the GPU), with both ranks sharing this box's one GPU over gloo (PIFU_BENCH_ONE_GPU: RCCL refuses two
    ranks on one device): slab-sharded lattice with the all-gather inside the timed region, max-over-ranks timing, the
    sharded gen_mesh of one subject, the independent subjects and the config-5 leg, ONE JSON line from rank 0.  A rehearsal,
    not a number."""
    import json
    import subprocess
    from helpers import free_port; port = free_port()
    tail = ['--gpus', '2', '--steps', '1', '--warmup', '1', '--res', '64', '--genmesh-res', '64', '--config5-res', '128',
            '--batch-rows', '262144', '--no-split-precision', '--no-lean-pass', '--no-canonical']
    if launcher == 'torchrun':
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(port), 'bench.py'] + tail
    else:
        cmd = [sys.executable, 'bench.py'] + tail
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['PIFU_BENCH_ONE_GPU'] = '1'
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # rank 0 only
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    assert line['config']['queries_per_step'] == 2 * 64 ** 3            # 64^3 per rank
    gm = line['gen_mesh']
    assert 'error' not in gm['one_subject_sharded'] and gm['one_subject_sharded']['n_verts'] > 0
    assert 'error' not in gm['independent_subjects'] and gm['independent_subjects']['subjects'] == 2
    # the legs explain themselves (first contact with an 8-GPU node): world, collectives, their bytes and the time inside them,
    # the colour rounds; the independent subjects are timed warm, and every rank's host threads are bounded
    oss = gm['one_subject_sharded']
    assert oss['world'] == 2 and oss['result_on'] == 'rank0' and oss['collectives'] >= 2 and oss['collective_bytes'] > 0
    assert oss['collective_s'] >= 0 and oss['collective_s_max_over_ranks'] >= 0 and oss['colour_rounds'] >= 1
    assert gm['independent_subjects']['warmed'] is True and gm['host_threads_per_rank_independent_subjects'] >= 2
    # round 6: what of the sharded call does not shrink with N, from the call's own measurements (a bad curve explains itself)
    parts = oss['predicted_fixed_parts']
    assert set(parts) == {'encoder_chain_s', 'replicated_s', 'collective_s', 'png_s', 'obj_tail_s'} and oss['replicated_s'] > 0
    assert abs(oss['predicted_fixed_s'] + oss['parts_beside_each_other_s'] - sum(parts.values())) < 1e-3
    assert 0 < oss['predicted_fixed_s'] <= oss['total_s'] * 1.001
    assert abs(oss['predicted_fixed_s'] + oss['shrinking_s'] - oss['total_s']) < 2e-3 and oss['speedup_ceiling_vs_this_run'] >= 1.0
    assert line['config']['batch_rows'] <= line['config']['batch_rows_requested']       # sized from the card's free memory
    assert 'error' not in gm['config5_1024'], gm['config5_1024']
    assert gm['config5_1024']['mesh_identical_on_all_ranks'] and gm['config5_1024']['n_verts'] > 0


@pytest.mark.timeout(700)
def test_bench_a_failed_rank_ends_the_job():
    """A rank that fails must end the job with a non-zero code instead of leaving its siblings in a collective: rank 1 raises
    before the first all-gather (PIFU_BENCH_FAIL_RANK, a test hook), rank 0 is already waiting in it; the launcher ends rank 0
    and the plain `python bench.py --gpus 2` call hands the failure back."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(PIFU_BENCH_ONE_GPU='1', PIFU_BENCH_FAIL_RANK='1')
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--res', '64', '--no-genmesh', '--no-cpu-baseline',
                        '--features', 'synthetic'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert 'test hook' in r.stderr


@pytest.mark.timeout(700)
def test_bench_a_rank_failing_inside_a_sharded_leg_ends_the_job_quickly():
    """The asymmetric failure the N > 1 legs must survive without hanging: ONE rank raises in the middle of the sharded
    gen_mesh leg while its sibling is already waiting for it in the next collective.  The failing rank waits `--leg-timeout`
    seconds on the control group's barrier, finds nobody, and leaves the process; the launcher ends the job (non-zero code,
    no JSON line) - within seconds, not after the process group's 20-minute timeout."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(PIFU_BENCH_ONE_GPU='1', PIFU_BENCH_FAIL_IN_LEG='1')
    t0 = time.time()
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '1', '--warmup', '1', '--res', '64', '--genmesh-res', '64',
                        '--config5-res', '0', '--batch-rows', '262144', '--no-cpu-baseline', '--leg-timeout', '10'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert time.time() - t0 < 300
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert 'test hook: this rank fails inside the sharded leg' in r.stderr and 'leaving so that the launcher ends the job' in r.stderr


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] as far as one GPU reaches: ONE subject on a 1024^3 grid, two ranks (gloo, both on this box's GPU),
# octree levels sharded by chunk ranges, marching cubes over one axis-0 slab per rank with dist.SLAB_MC_MIN_RESOLUTION at
# its real value (1024), colour loop sharded by calc_normal calls.
# ---------------------------------------------------------------------------------------------------------------------
def _build_1024():
    """The tiny golden nets with the MLP weights at a scale that keeps the 1024^3 surface at a few million vertices."""
    from pifuhd_amd import nets, synth
    from helpers import golden_nets
    sd, oc, of = golden_nets(mlp_std=0.03)
    of.loadSizeBig = 128
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = _data()
    netMR.filter_global(data['img_512']); netMR.filter_local(data['img'][:, None])
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3, resolution=32)
    return netMR, data


def _digest(*arrays):
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _worker_1024(rank, ws, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import dist as pdist, reconstruction
    assert pdist.SLAB_MC_MIN_RESOLUTION == 1024
    netMR, data = _build_1024()
    timings = {}
    verts, faces, color = reconstruction.gen_mesh(1024, netMR, 'cuda', data, os.path.join(outdir, f'm1024_r{rank}.obj'),
                                                  sharded=True, timings=timings, result_on='all')
    with open(os.path.join(outdir, f'sha1024_r{rank}.txt'), 'w') as f:
        f.write(_digest(verts, faces, color))
    if rank == 0:
        np.savez(os.path.join(outdir, 'mesh1024_r0.npz'), verts=verts, faces=faces, color=color)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1100)
def test_gen_mesh_1024_sharded_world2_with_slab_marching_cubes(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import free_port; port = free_port()
    mpctx = mp.get_context('spawn')
    procs = [mpctx.Process(target=_worker_1024, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=1000)
        assert p.exitcode == 0
    shas = [open(tmp_path / f'sha1024_r{r}.txt').read() for r in range(2)]
    assert shas[0] == shas[1]                                             # every rank holds the same mesh
    assert os.path.exists(tmp_path / 'm1024_r0.obj') and not os.path.exists(tmp_path / 'm1024_r1.obj')

    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import _lib, reconstruction
    from pifuhd_amd.nets import _ptr, _stream_ptr
    netMR, data = _build_1024()
    verts, faces, color = reconstruction.gen_mesh(1024, netMR, 'cuda', data, str(tmp_path / 'single1024.obj'))
    assert _digest(verts, faces, color) == shas[0]                        # ... and it is the single-process mesh, bit for bit
    got = np.load(tmp_path / 'mesh1024_r0.npz')
    assert np.array_equal(got['faces'], faces) and np.array_equal(got['verts'], verts) and np.array_equal(got['color'], color)
    # rank 0's file is the single-process file
    import filecmp
    assert filecmp.cmp(tmp_path / 'm1024_r0.obj', tmp_path / 'single1024.obj', shallow=False)

    # size-independent properties of the mesh.  A Lewiner surface is a closed 2-manifold wherever it does not leave the
    # volume: every edge is used by exactly two triangles, in opposite directions, except edges lying in a face of the box.
    V, F = len(verts), len(faces)
    assert V > 100000 and F > 200000
    f64 = faces.astype(np.int64)
    a = np.concatenate([f64[:, 0], f64[:, 1], f64[:, 2]])
    b = np.concatenate([f64[:, 1], f64[:, 2], f64[:, 0]])
    directed = a * V + b
    assert len(np.unique(directed)) == len(directed)                      # no directed edge twice: consistent orientation
    und, cnt = np.unique(np.minimum(a, b) * V + np.maximum(a, b), return_counts=True)
    odd = und[cnt != 2]
    grid = (verts * np.array([1.0, -1.0, 1.0]) + 1.0) * (1024 / 2.0)      # calib = diag(1, -1, 1): back to node coordinates
    on_box = ((np.abs(grid) < 1e-6) | (np.abs(grid - 1023.0) < 1e-6)).any(1)
    assert on_box[odd // V].all() and on_box[odd % V].all()
    assert (cnt <= 2).all()
    E = len(und)
    chi = V - E + F
    # components from the device union-find (the meshcleaning kernel): chi = 2 (components - genus) for the closed ones
    ctx = netMR._ctx
    fd = torch.from_numpy(np.ascontiguousarray(faces, dtype=np.int32)).cuda()
    label = torch.empty(F, device='cuda', dtype=torch.int32)
    open_edge = torch.empty(F, device='cuda', dtype=torch.uint8)
    _lib.call('pifu_face_components', ctx.h, _ptr(fd), F, _ptr(label), _ptr(open_edge), _stream_ptr(ctx.device))
    ncomp = len(torch.unique(label))
    if len(odd) == 0:
        assert chi % 2 == 0 and chi <= 2 * ncomp
    print(f'1024^3 sharded gen_mesh: V={V} F={F} E={E} chi={chi} components={ncomp} open_edges={len(odd)}')


@pytest.mark.timeout(900)
def test_bench_single_gpu_line_carries_every_contract_field():
    """`python bench.py` (N = 1) at reduced sizes: ONE JSON line with the contract's fields - metric / value / unit /
    dtype / config.workload, `roofline` {bound, achieved, peak, unit, frac, traffic}, `cpu_baseline` {value, unit, cores,
    kind, sample} - and the round-3 additions (`chunk_100000`, the canonical GEMM's `frac_at_measured_clock`, `config5_1024`,
    the CPU gen_mesh baseline with the SDF difference of the whole multi-level octree field)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    cmd = [sys.executable, 'bench.py', '--steps', '1', '--warmup', '1', '--res', '64', '--genmesh-res', '64', '--config5-res', '96',
           '--batch-rows', '262144', '--cpu-seconds', '2', '--cpu-chunks', '4', '--cpu-genmesh-res', '64', '--cpu-genmesh-res2', '0',
           '--canonical-steps', '1']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and r.stdout.strip() == lines[0]           # stdout carries the JSON line and nothing else
    d = json.loads(lines[0])
    assert d['metric'] == 'occupancy_queries_throughput' and d['unit'] == 'Mqueries/s' and d['value'] > 0
    assert d['n_gpus'] == 1 and d['dtype'] == 'f32' and d['higher_is_better'] is True and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    ro = d['roofline']
    assert ro['bound'] == 'mfma' and ro['unit'] == 'TFLOP/s' and ro['peak'] == 157.3 and 'traffic' in ro
    assert abs(ro['frac'] - ro['achieved'] / ro['peak']) < 1e-3 and ro['achieved'] > 0
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['cores'] >= 1 and cb['unit'] == 'Mqueries/s' and cb['sample']
    assert d['parity_check_max_abs_diff'] < 1e-4
    assert d['chunk_100000']['value'] > 0
    assert d['canonical_path']['gemm_roofline']['frac_at_measured_clock'] > 0
    assert d['canonical_path']['max_abs_diff_vs_headline_field'] < 1e-4
    cn = d['calc_normal']                   # round 4: the colour loop's kernel path through the pixel tables and per point
    assert cn['pixel_tables']['value'] > 0 and cn['per_point_gemms']['value'] > 0 and cn['max_abs_diff_between_paths'] < 2e-5
    assert cn['pixel_tables']['gemm_flop_per_query_executed'] < cn['per_point_gemms']['gemm_flop_per_query_executed']
    g = d['gen_mesh']
    assert g['total_s'] > 0 and g['mc_check']['faces_bit_exact_vs_oracle'] and g['mc_check']['verts_bit_exact_vs_oracle']
    assert g['config5_1024']['resolution'] == 96 and g['config5_1024']['n_verts'] > 0
    cg = d['cpu_baseline_gen_mesh']
    assert cg['resolution'] == 64 and cg['value'] > 0
    assert cg['gpu_same_resolution']['max_abs_diff_sdf'] < 1e-4 and cg['gpu_same_resolution']['same_points_evaluated']
    assert {x['bound'] for x in d['roofline_secondary']} >= {'hbm'}
    # round 6: the body-like subject and the pifuhd.pt configuration (GroupNorm filters, MLP without norm layers), each with the
    # same roofline block as the headline subject; the CPU baseline at both thread counts
    for blk in (g['body_subject'], d['pifuhd_pt_config']['gen_mesh']):
        assert 'error' not in blk and blk['n_verts'] > 0 and blk['mesh']['components'] >= 1 and blk['points_evaluated'] > 0
        ro2 = blk['roofline']
        assert ro2['bound'] == 'mfma' and abs(ro2['frac'] - ro2['executed_tflop_per_call'] / blk['total_s'] / 157.3) < 2e-3
        assert {'reconstruction', 'normals', 'encoders'} <= set(ro2)
    po = d['pifuhd_pt_config']['occupancy']
    assert po['value'] > 0 and po['library_group_rows'] % 128 == 0 and po['library_group_rows'] % 64 == 0 and po['parity_check_max_abs_diff'] < 1e-4
    assert abs(po['roofline']['frac'] - po['roofline']['achieved'] / 157.3) < 1e-3
    if cb['cores'] != (os.cpu_count() or 1):
        assert cb['cores_alt'] == os.cpu_count() and cb['value_alt'] > 0 and cb['faster_choice'] in ('cores', 'cores_alt')
