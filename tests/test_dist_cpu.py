"""Chunk-granular sharding + all-gather (pifuhd_amd/dist.py) on CPU with the gloo backend, world size 2.
The evaluation function is the CPU oracle: a sharded run must reproduce the single-process result exactly,
because every rank evaluates whole chunks (GroupNorm couples a chunk)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_chunks_cover_and_align():
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd.dist import shard_chunks
    for total, chunk, ws in [(12345, 500, 2), (5000, 5000, 4), (4999, 5000, 8), (262144, 5000, 8), (7, 3, 5), (0, 10, 2)]:
        covered = []
        for r in range(ws):
            first, cnt, per = shard_chunks(total, chunk, ws, r)
            assert first % chunk == 0 or cnt == 0            # chunk boundaries = reference chunk boundaries
            assert cnt <= per * chunk
            covered.extend(range(first, first + cnt))
        assert covered == list(range(total))


def _free_port():
    from helpers import free_port
    return free_port()


def _worker(rank, ws, port, total, chunk, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    torch.set_num_threads(2)
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd.dist import sharded_eval
    from helpers import calib_tensor, mlp_only_nets, seeded_features, seeded_points
    from oracle import pifu_oracle as O
    sd, oc, of = mlp_only_nets()
    nets = O.OracleNets(sd, oc, of)
    nets.feat_coarse, nets.feat_fine = seeded_features(7)
    pts = seeded_points(21, total, lim=1.0)
    calib = calib_tensor()

    def eval_fn(first, count):
        out = []
        with torch.no_grad():
            for s in range(first, first + count, chunk):
                e = min(s + chunk, first + count)
                out.append(nets.query(pts[:, :, s:e], calib)[0, 0])
        return torch.cat(out)
    full = sharded_eval(total, chunk, eval_fn, torch.device('cpu'))
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_eval_gloo_world2_matches_single_process():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import calib_tensor, mlp_only_nets, seeded_features, seeded_points
    from oracle import pifu_oracle as O
    total, chunk = 1700, 300           # 6 chunks (last ragged): rank 0 gets 3, rank 1 gets 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=500)
    for p in procs:
        p.join(timeout=100)
        assert p.exitcode == 0
    sd, oc, of = mlp_only_nets()
    nets = O.OracleNets(sd, oc, of)
    nets.feat_coarse, nets.feat_fine = seeded_features(7)
    pts = seeded_points(21, total, lim=1.0)
    with torch.no_grad():
        want = torch.cat([nets.query(pts[:, :, s:s + chunk], calib_tensor())[0, 0] for s in range(0, total, chunk)]).numpy()
    # same chunk membership => same statistics; only the CPU thread count (2 vs all) perturbs the last ulp
    assert np.abs(got - want).max() < 5e-6
    # and NOT the result of ignoring chunk boundaries
    with torch.no_grad():
        whole = nets.query(pts, calib_tensor())[0, 0].numpy()
    assert np.abs(whole - want).max() > 1e-4


def test_normal_spans_follow_reference_colour_loop():
    """`reconstruction.py:58-70`: 50 000-vertex slices, the last one ends at -1."""
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd.reconstruction import _normal_spans
    for v in (1, 2, 49999, 50000, 50001, 50002, 120000, 249422):
        want = []
        idx = np.arange(v)
        for i in range(v // 50000 + 1):
            left = i * 50000
            right = -1 if i == v // 50000 else (i + 1) * 50000
            sl = idx[left:right]
            want.append((left, left + len(sl)))
        assert _normal_spans(v) == want
    assert _normal_spans(100000)[-1] == (100000, 100000)        # empty slice: the reference raises there (C-6)


def _rounds_worker(rank, ws, port, n_verts, interval, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import dist as pdist
    from pifuhd_amd.reconstruction import _normal_spans
    spans = _normal_spans(n_verts, interval)
    seen, prefix = [], []
    out = torch.zeros(n_verts, 3)
    pdist.stats_begin()
    for (base, m) in pdist.colour_rounds(len(spans), ws, per_call_cap=3):
        mine, (L, R), slot = pdist.round_rows(spans, base, m, ws, rank, interval)
        local = torch.zeros(slot, 3)
        if mine is not None:
            seen.append(mine)
            rows = torch.arange(mine[0], mine[1], dtype=torch.float32)
            local[:mine[1] - mine[0]] = torch.stack([rows, rows * 2 + 1, -rows], 1)
        full = pdist.all_gather_rows(torch.empty(ws * slot, 3), local)
        out[L:R] = full[:R - L]
        prefix.append((L, R))
    st = pdist.stats_end()
    q.put((rank, seen, prefix, out.numpy(), st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('ws', [2, 3])
def test_colour_rounds_gloo(ws):
    """The colour loop of the sharded gen_mesh: whole calc_normal calls are dealt out in rounds, every call runs exactly
    once, and after each round's all-gather the finished rows are a PREFIX of the vertex list in order (what the OBJ
    writer's ready(upto) needs); the last vertex is never visited (SURVEY C-6)."""
    interval, n_verts = 7, 7 * 11 + 3                       # 12 spans, the last one ragged and one row short
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rounds_worker, args=(r, ws, port, n_verts, interval, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in range(ws)]
    for p in procs:
        p.join(timeout=100)
        assert p.exitcode == 0
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd.reconstruction import _normal_spans
    spans = _normal_spans(n_verts, interval)
    rows = np.arange(n_verts, dtype=np.float32)
    want = np.stack([rows, rows * 2 + 1, -rows], 1)
    want[-1] = 0
    covered = []
    for (_, seen, prefix, out, st) in res:
        assert np.array_equal(out, want)
        assert prefix[0][0] == 0 and prefix[-1][1] == n_verts - 1
        assert all(a[1] == b[0] for a, b in zip(prefix, prefix[1:]))          # round k + 1 starts where round k ended
        assert st['collectives'] == len(prefix) and st['collective_bytes'] > 0
        for (l, r) in seen:                                                  # runs of whole spans
            covered += [s for s in spans if l <= s[0] and s[1] <= r]
    assert sorted(covered) == spans
    per_rank = [sum(1 for s in spans if any(l <= s[0] and s[1] <= r for (l, r) in seen)) for (_, seen, _, _, _) in res]
    assert max(per_rank) <= -(-len(spans) // ws)


def test_colour_rounds_shapes():
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd.dist import colour_rounds
    for n in (1, 2, 9, 10, 11, 86, 149, 400):
        for ws in (1, 2, 3, 4, 8):
            r = colour_rounds(n, ws)
            assert r[0][0] == 0 and all(m >= 1 and m <= 10 for (_, m) in r)
            assert all(b[0] == a[0] + ws * a[1] for a, b in zip(r, r[1:]))           # rounds tile the span list
            assert r[-1][0] < n <= r[-1][0] + ws * r[-1][1]
            assert sum(m for (_, m) in r) == -(-n // ws) or r[-1][0] + ws * r[-1][1] >= n
    # the rounds taper: what is left once the GPU has finished is the last round's exchange + vertex lines
    assert [m for (_, m) in colour_rounds(86, 1)] == [10] * 7 + [8, 4, 2, 1, 1]
    assert [m for (_, m) in colour_rounds(86, 2)] == [10, 10, 10, 7, 3, 2, 1]
    assert [m for (_, m) in colour_rounds(86, 8)] == [6, 3, 1, 1]


def test_slab_planes_cover_the_cell_layers_once():
    """Slabs of the sharded marching cubes: contiguous runs of cell layers in rank order, neighbours share one node
    plane, ranks left without a layer are the last ones."""
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd.mesh_util import slab_planes
    for n0, ws in [(512, 8), (1024, 8), (513, 8), (5, 3), (3, 4), (2, 1), (2, 8), (100, 7)]:
        runs = slab_planes(n0, ws)
        assert len(runs) == ws
        layers, seen_empty = [], False
        for run in runs:
            if run is None:
                seen_empty = True
                continue
            assert not seen_empty
            z, n = run
            assert n >= 2 and z + n <= n0
            layers.extend(range(z, z + n - 1))
        assert layers == list(range(n0 - 1))


def test_shard_chunks_at_config5_scale():
    """BASELINE config 5: a 1024^3 lattice (2^30 points, 214 749 chunks of 5000) over 8 ranks - the shards are contiguous,
    chunk-aligned, cover every point once, and only the last one is short - by fewer than 8 chunks of 26 844 (checked on the
    boundaries only)."""
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd.dist import shard_chunks
    for total, chunk, ws in [(1024 ** 3, 5000, 8), (1024 ** 3, 5000, 7), (512 ** 3, 5000, 8), (1024 ** 3, 10000, 8)]:
        end = 0
        counts = []
        for r in range(ws):
            first, cnt, per = shard_chunks(total, chunk, ws, r)
            assert first == end and first % chunk == 0
            end = first + cnt
            counts.append(cnt)
        assert end == total
        per = -(-(-(-total // chunk)) // ws)                     # chunks per rank, rounded up
        assert max(counts) == per * chunk and min(counts) > (per - ws) * chunk      # only the last rank is short, by < ws chunks


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without RANK / WORLD_SIZE must start the ranks itself: a fresh `python -m
    torch.distributed.run` child with this node's loop-back rendezvous, the bench script and the caller's own arguments,
    whose return code it hands back - decided before torch or the GPU is touched (checked on the command it builds)."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)                      # importing bench.py must not import torch.distributed work or the GPU
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, **kw):
        seen['cmd'], seen['kw'] = cmd, kw
        return Done()
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '5', '--warmup', '2'])
    assert bench.launch_ranks(4) == 7                   # the child's return code comes back
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[script + 1:] == ['--gpus', '4', '--steps', '5', '--warmup', '2']
    assert os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'          # set before the HIP runtime can load
