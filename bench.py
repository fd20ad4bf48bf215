#!/usr/bin/env python3
"""Headline benchmark: occupancy Mqueries/s of the two-level PIFuHD query over a dense evaluation grid.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A step = one pass of the hot path over one batch of synthetic input = one dense grid evaluation:
256^3 lattice points per GPU (BASELINE.json configs[1]: coarse+fine PIFuMRNet, 1024 px input, fp32),
cut into the reference's 5000-point query() chunks (GroupNorm couples a chunk, so chunking is part of
the numerical contract).  With N GPUs the lattice is (N*256) x 256 x 256: rank r evaluates a contiguous
range of whole chunks (an axis-0 slab) and ONE RCCL all-gather of the fp32 occupancy field follows,
inside the timed region (weak scaling: 256^3 queries per GPU).  Feature maps come from the real
filter_global / filter_local of a synthetic 1024 px image with name-seeded random weights and are
resident in HBM before the timed region, as are the weights; lattice points are generated on the device.

One JSON line on rank 0.  `roofline` is for the dominant kernel k_mlp_gemm_tri (exact-fp32 MFMA): algorithmic
FLOPs of its launches / their summed HIP-event durations, against the 157.3 TFLOP/s fp32-matrix peak.
`cpu_baseline` times the CPU oracle (kind "port": the reference's Python cannot travel to the GPU box) on
a bounded sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL (and any sharing of device tensors between processes) needs dmabuf IPC handles on this pool: the host driver has no
# legacy IPC mode, and without this setting hipIpcGetMemHandle fails with "invalid argument" as soon as two ranks meet.
# Must be in the environment before the HIP runtime loads, i.e. before torch is imported.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

FLOP_PER_QUERY = 2916098            # coarse MLP 2,100,738 + fine MLP 815,360
PEAK_FP32_MATRIX_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec peak
PEAK_INFINITY_CACHE_GATHER_GBS = 8600.0   # MI355X_MICROARCH.md "Indexed rows": uniformly random rows of a 38 MB table


def read_prof(_lib, h, which, reset=1):
    """(ms, launches, work) of one profiling slot of the library context."""
    ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
    _lib.call('pifu_profile_read', h, which, C.byref(ms), C.byref(n), C.byref(w), reset)
    return ms.value, n.value, w.value


def reset_prof(_lib, h):
    for which in (_lib.PROF_GEMM, _lib.PROF_SAMPLER, _lib.PROF_MC, _lib.PROF_ENC, _lib.PROF_MC_NODES, _lib.PROF_WALK, _lib.PROF_ENC_CANONICAL):
        _lib.call('pifu_profile_read', h, which, None, None, None, 1)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a FRESH child (`python -m torch.distributed.run`,
    the command the driver itself uses) before torch or the GPU is touched in this process, let it write to this process's
    stdout / stderr, and hand back its return code.  The elastic agent ends the other ranks when one of them fails, so a
    failed rank cannot leave the rest waiting in a collective."""
    import random
    import socket
    import subprocess
    # a free port BELOW the kernel's ephemeral range (32768+): one probed with bind(0) can be handed out again - as the source
    # port of somebody's outgoing connection - between the probe and the launcher's own bind (seen once: EADDRINUSE in the
    # rendezvous TCPStore); ports in 15000 - 29999 are only ever taken by listeners, which the probe sees
    port = None
    rng = random.Random(os.getpid() ^ int(time.time() * 1e3))
    for _ in range(64):
        cand = rng.randrange(15000, 30000)
        s = socket.socket()
        try:
            s.bind(('127.0.0.1', cand))
            port = cand
            break
        except OSError:
            continue
        finally:
            s.close()
    if port is None:
        raise SystemExit('no free rendezvous port found in 15000-29999')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, cwd=ROOT).returncode


def measured_mhz(clk0, clk1):
    """Average shader clock between two `pifu_clock_probe` samples: shader-counter ticks per real-time tick (100 MHz), per
    XCD that both probes sampled, median over the XCDs (an XCD whose counter was reset by power gating reads garbage and
    is dropped).  s_memtime counts the clock the CUs actually ran at, power management included."""
    mhz = [100.0 * (clk1[2 * x] - clk0[2 * x]) / (clk1[2 * x + 1] - clk0[2 * x + 1])
           for x in range(8) if clk0[2 * x + 1] and clk1[2 * x + 1] > clk0[2 * x + 1]]
    mhz = sorted(m for m in mhz if 100.0 < m < 4000.0)
    return mhz[len(mhz) // 2] if mhz else None


def rccl_choices(path):
    """What RCCL logged (NCCL_DEBUG=INFO, subsystems INIT / GRAPH / TUNING) about the communicator of this rank: how many
    channels, ring or tree, which transport each channel uses (P2P/IPC over xGMI, SHM, ...), and the algorithm / protocol
    lines of the tuning subsystem where the build emits them.  One short summary per rank, taken after the warm-up
    collectives have run."""
    import re
    if not path or not os.path.exists(path):
        return None
    keep = re.compile(r'(Algo|algorithm|[Pp]roto|Ring \d|Trees|Channel \d+/\d+ *:|via |nChannels|Pattern|XGMI|P2P|comm 0x.*nranks)')
    lines, seen = [], set()
    with open(path, errors='replace') as fh:
        for ln in fh:
            if not keep.search(ln):
                continue
            txt = ln.split('NCCL INFO', 1)[-1].strip()
            key = re.sub(r'\d+', '#', txt)[:60]          # one example per kind of line
            if key in seen:
                continue
            seen.add(key)
            lines.append(txt[:200])
    text = ' '.join(lines)
    return {'lines': lines[:24], 'mentions_ring': 'Ring' in text, 'mentions_tree': 'Tree' in text,
            'transports': sorted(set(re.findall(r'via ([A-Za-z0-9/_]+)', text)))}


def smoother_subject(args, dev, data, share=None):
    """The second bench subject: the same image and architecture with the MLP weights drawn at torch's stock scale (std
    0.02 instead of the 0.16 that gives the headline subject its rough, multi-million-vertex surface).  Fewer points survive
    the octree and fewer vertices need normals - the other end of what a reconstruction can look like.  Returns netMR with
    both filters run and the 0.5 level set centred in the volume."""
    from pifuhd_amd import config, nets, synth
    oc = config.coarse_opt(in_image_channels=args.image_channels)
    of = config.fine_opt(oc)
    sd2 = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.02)
    _, netMR2 = nets.build_nets(oc, of, sd2, dev)
    if share is not None:      # several ranks on one card (PIFU_BENCH_ONE_GPU): this context's batch workspace takes its share only
        reserve_share(_lib_mod(), netMR2, 1048576, share)
    netMR2.filter_global(data['img_512'].to(dev))
    netMR2.filter_local(data['img'].to(dev)[:, None])
    synth.calibrate_occupancy_bias(netMR2, data['calib'], sd2, 0.3)
    return netMR2


def reserve_share(_lib, net, max_rows, share):
    """Batch workspace of a context sized from the device's free memory (pifu_reserve_auto): `share` of it at most."""
    rows = C.c_int64(0)
    _lib.call('pifu_reserve_auto', net._ctx.h, int(max_rows), float(share), C.byref(rows))
    return int(rows.value)


def body_subject(args, dev, mlp_norm, norm, mlp_std, share=0.25, net=None, sd=None):
    """The body-like subject: the silhouette image (synth.synthetic_body_image) through the real encoders, and the two MLPs
    rewritten so that the net predicts ONE smooth closed body (synth.craft_body_mlps: a handful of wire channels in an
    otherwise name-seeded network; same architecture, kernels and FLOPs).  What a trained PIFuHD produces - a few hundred
    thousand vertices, an octree that refines a thin shell (2 - 4 % of the grid) - and name-seeded random weights do not.
    Returns (netG, netMR, data, fit statistics); `net` / `sd`: reuse an existing net of the same configuration."""
    from pifuhd_amd import config, nets, synth
    data_b = synth.synthetic_inputs(1024, args.image_channels, seed=0, kind='body')
    if net is None:
        oc = config.coarse_opt(in_image_channels=args.image_channels, mlp_norm=mlp_norm, norm=norm)
        of = config.fine_opt(oc)
        sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=mlp_std)
        netG_b, net = nets.build_nets(oc, of, sd, dev)
        reserve_share(_lib_mod(), net, 1048576, share)
    else:
        netG_b = net.netG
    net.filter_global(data_b['img_512'].to(dev))
    net.filter_local(data_b['img'].to(dev)[:, None])
    sd_b, stats = synth.craft_body_mlps(sd, netG_b.opt, net.opt, netG_b.im_feat_list[-1])
    net.load_state_dict(sd_b)
    return netG_b, net, data_b, stats


def _lib_mod():
    from pifuhd_amd import _lib
    return _lib


def mesh_components(net, faces):
    """(connected components, watertight ones, faces of the largest) of a triangle list (pifu_face_components)."""
    import numpy as np
    import torch
    from pifuhd_amd import _lib
    from pifuhd_amd.nets import _ptr, _stream_ptr
    ctx = net._ctx
    f = torch.from_numpy(np.ascontiguousarray(faces, dtype=np.int32)).to(ctx.device)
    label = torch.empty(len(f), device=ctx.device, dtype=torch.int32)
    flag = torch.empty(len(f), device=ctx.device, dtype=torch.uint8)
    _lib.call('pifu_face_components', ctx.h, _ptr(f), len(f), _ptr(label), _ptr(flag), _stream_ptr(ctx.device))
    lab, fl = label.cpu().numpy(), flag.cpu().numpy()
    roots, inv, cnt = np.unique(lab, return_inverse=True, return_counts=True)
    leaky = np.zeros(len(roots), bool)
    np.logical_or.at(leaky, inv, (fl & 1) != 0)
    return {'components': int(len(roots)), 'watertight_components': int((~leaky).sum()), 'faces_of_largest': int(cnt.max()),
            'largest_is_watertight': bool(~leaky[int(cnt.argmax())])}


def gen_mesh_on_roofline(recon, _lib, netMR, dev, data, res, runs, td, tag, log, cold=None):
    """`runs` timed gen_mesh calls of one subject at `res`^3 after a warm-up call at that size (arenas, pinned result slabs,
    pixel table), the median run's phases, and the call on the roofline: ALGORITHMIC FLOPs executed by one call (MLP GEMM
    launches + encoder convolutions, counted by the library) / wall time / the fp32-matrix peak, the GEMMs' busy time, and -
    from one more, untimed call whose phase boundaries synchronise - the split into reconstruction / normals / encoders."""
    import torch
    h = netMR._ctx.h

    def drop(name):
        for f in (name, name[:-4] + '.png'):
            try:
                os.remove(os.path.join(td, f))
            except OSError:
                pass
    torch.cuda.synchronize(dev)
    w0 = time.perf_counter()
    recon.gen_mesh(res, netMR, dev, data, os.path.join(td, tag + '_warm.obj'))
    torch.cuda.synchronize(dev)
    drop(tag + '_warm.obj')
    first_call = round(time.perf_counter() - w0, 4)
    if cold is not None:
        cold['first_gen_mesh_call_s'] = first_call
    reset_prof(_lib, h)
    _lib.call('pifu_profile_enable', h, 1)
    try:
        run_t = []
        for k in range(max(runs, 1)):      # distinct files: replacing a 470 MB file costs its page frees
            tk = {}
            recon.gen_mesh(res, netMR, dev, data, os.path.join(td, '%s_%d.obj' % (tag, k)), timings=tk)
            run_t.append(tk)
            drop('%s_%d.obj' % (tag, k))
        _lib.call('pifu_profile_enable', h, 0)
        extra = {'mc': read_prof(_lib, h, _lib.PROF_MC, 0), 'mc_nodes': read_prof(_lib, h, _lib.PROF_MC_NODES, 0)}
        gm_ms, gm_n, gm_fl = read_prof(_lib, h, _lib.PROF_GEMM)          # the MLP GEMM launches of the timed runs
        _, en_n, en_fl = read_prof(_lib, h, _lib.PROF_ENC)               # the encoders' convolutions (counted, not timed)
        nr = len(run_t)
        gm_ms, gm_n, gm_fl, en_n, en_fl = gm_ms / nr, gm_n // nr, gm_fl / nr, en_n // nr, en_fl / nr
        timings = sorted(run_t, key=lambda t: t['total_s'])[nr // 2]      # the median run, phase by phase
        res_d = {'resolution': res, 'use_octree': True,
                 **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in timings.items()},
                 'total_s_is': 'median of the runs below (each a complete gen_mesh call incl. both encoders, PNG and OBJ)',
                 'runs_total_s': [round(t['total_s'], 4) for t in run_t], 'first_call_s': first_call}
        tot = timings['total_s']
        ro = {'bound': 'mfma', 'peak': PEAK_FP32_MATRIX_TFLOPS, 'unit': 'TFLOP/s',
              'executed_tflop_per_call': round((gm_fl + en_fl) / 1e12, 3),
              'mlp_gemm_tflop_per_call': round(gm_fl / 1e12, 3), 'mlp_gemm_launches_per_call': gm_n,
              'encoder_conv_tflop_per_call': round(en_fl / 1e12, 3), 'encoder_conv_launches_per_call': en_n,
              'achieved': round((gm_fl + en_fl) / tot / 1e12, 2),
              'frac': round((gm_fl + en_fl) / tot / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
              'mlp_gemm_busy_s_per_call': round(gm_ms * 1e-3, 4),
              'mlp_gemm_busy_share_of_call': round(gm_ms * 1e-3 / tot, 4),
              'mlp_gemm_frac_while_busy': round(gm_fl / (gm_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4) if gm_ms > 0 else None,
              'definition': 'ALGORITHMIC FLOPs executed by one gen_mesh call (k_mlp_gemm_tri launches: 2 * valid points * Cout '
                            '* Cin; encoder convolutions: 2 * output pixels * Cout * Cin * taps) / total_s / peak; busy = summed '
                            'HIP-event durations of the GEMM launches'}
        try:
            reset_prof(_lib, h)
            _lib.call('pifu_profile_enable', h, 1)
            tp = {'profile_phases': True}
            recon.gen_mesh(res, netMR, dev, data, os.path.join(td, tag + '_phases.obj'), timings=tp)
            _lib.call('pifu_profile_enable', h, 0)
            drop(tag + '_phases.obj')
            for ph in ('reconstruction', 'normals'):
                fl, ms, sec = tp['gemm_flop_' + ph], tp['gemm_ms_' + ph], timings[ph + '_s']
                ro[ph] = {'mlp_gemm_tflop': round(fl / 1e12, 3), 'phase_s': round(sec, 4), 'mlp_gemm_busy_s': round(ms * 1e-3, 4),
                          'frac': round(fl / sec / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
                          'frac_while_busy': round(fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4) if ms > 0 else None}
            ro['encoders'] = {'conv_tflop': round(en_fl / 1e12, 3), 'phase_s': round(timings['filter_s'], 4),
                              'frac': round(en_fl / timings['filter_s'] / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4)}
        except Exception as e:
            ro['phase_split_error'] = repr(e)
        res_d['roofline'] = ro
        log(f"gen_mesh[{tag}] {res}^3: {tot:.4f} s, {timings.get('n_verts')} vertices; roofline {ro['executed_tflop_per_call']} TFLOP "
            f"= {ro['achieved']} TFLOP/s = {ro['frac']} of peak")
        return res_d, timings, extra, nr
    finally:
        _lib.call('pifu_profile_enable', h, 0)


def octree_share(mesh_util, netMR, calib, res):
    """Points the octree of `reconstruction()` evaluates on this subject, and their share of the res^3 grid."""
    _, n_eval = mesh_util.eval_grid_device(netMR, calib, res, 5000, True)
    return {'points_evaluated': int(n_eval), 'share_of_grid': round(n_eval / float(res) ** 3, 4)}


def mesh_digest(verts, faces, color):
    """56 bits of the sha256 over a mesh's arrays, as a non-negative int64 (for an all-reduce across ranks)."""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    for a in (verts, faces, color):
        h.update(np.ascontiguousarray(a).tobytes())
    return int.from_bytes(h.digest()[:7], 'little')


def genmesh_multi(args, netMR, dev, data, synth, dist, rank, world, fence, log):
    """gen_mesh wall time on N GPUs: (a) ONE subject, grid evaluation + colour loop + encoders sharded (max over ranks);
    (b) N subjects, one per GPU, no communication (time for all N to finish); (c) BASELINE configs[4].

    A local failure must neither skip a collective the other ranks are in nor leave them waiting for ever.  So: no
    collective of this function sits inside a try block, every leg ends with `leg_end()` - a counter in the rendezvous store
    that every rank increments and polls, with a timeout - and a rank that failed inside a sharded call (its siblings are then stuck in that
    call's collectives and will never reach the barrier) runs into the timeout and leaves the process: the launcher ends
    the job within seconds instead of after the process group's 20 minutes.  A failure that is the same on every rank
    (marching cubes finding no surface) passes the barrier everywhere and is reported as an error string in the line."""
    import torch
    from pifuhd_amd import reconstruction as recon
    res = {'resolution': args.genmesh_res, 'use_octree': True, 'n_gpus': world}
    keys = ('filter_s', 'png_s', 'reconstruction_s', 'normals_s', 'obj_s')
    store = dist.distributed_c10d._get_default_store()          # the rendezvous TCPStore: the control plane of the legs

    def reduce_max(vals):
        t = torch.tensor(vals, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    def rm(path):
        # the legs' files go as soon as they are no longer needed: a 512^3 OBJ is ~0.5 GB, a 1024^3 one ~0.8 GB, per rank in leg (b)
        for f in (path, path[:-4] + '.png'):
            try:
                os.remove(f)
            except OSError:
                pass

    def guarded(what, fn):
        try:
            return fn(), None
        except Exception as e:
            log(f'{what} failed on this rank: {e!r}')
            return None, repr(e)

    leg_no = [0]
    scratch = []                # the legs' temporary directory: removed by hand where the process is left with os._exit

    def leg_end(what, err, sharded):
        # ranks that got through their (sharded) call arrive here; a rank whose call failed arrives too - and waits at most
        # args.leg_timeout seconds for the others, who are either here as well (the failure was the same everywhere) or
        # stuck inside the call it left.  A counter in the rendezvous store, polled: no collective, nothing to mismatch.
        leg_no[0] += 1
        key = f'pifu_bench_leg_{leg_no[0]}'
        store.add(key, 1)
        limit = args.leg_timeout if (err and sharded) else 1200.0
        t0 = time.time()
        while store.add(key, 0) < world:
            if time.time() - t0 > limit:
                log(f'{what}: the other ranks did not reach the end of the leg within {limit:.0f} s; leaving so that the '
                    f'launcher ends the job')
                sys.stderr.flush()
                import shutil
                for d in scratch:
                    shutil.rmtree(d, ignore_errors=True)
                os._exit(3)
            time.sleep(0.02)

    # N processes share this host: where EVERY rank writes a file (the independent subjects) each one's OBJ writer / PNG encoder
    # gets its share of the cores, not all of them; where only rank 0 writes (the sharded legs) it may use the machine
    host_threads = max(2, (os.cpu_count() or 8) // max(world, 1))
    res['host_threads_per_rank_independent_subjects'] = host_threads
    extra = ('collectives', 'collective_bytes', 'collective_s', 'collective_host_s', 'colour_rounds', 'replicated_s', 'replicated_host_s')

    def fixed_cost(t, total_s, coll_max):
        """What of a sharded call does NOT shrink with the number of ranks, from the call's own measurements (rank 0's view):
        the dealt-out encoder chain (netF -> fine hourglass stays on the critical path), the work every rank repeats on the
        replicated field (octree bookkeeping, marching cubes below the slab threshold), the time inside the collectives
        (max over ranks: it includes waiting for the slowest rank), the PNG hand-off and the tail after the GPU phase (last
        colour round's host copy + vertex lines).  total = fixed + shrinking; with twice the ranks the shrinking part halves
        at best - so a curve that flattens can be read off the line that shows it."""
        parts = {'encoder_chain_s': t.get('filter_s', 0.0), 'replicated_s': t.get('replicated_s', 0.0) or 0.0,
                 'collective_s': coll_max, 'png_s': t.get('png_s', 0.0), 'obj_tail_s': t.get('obj_s', 0.0)}
        # the parts are maxima over the ranks and partly run beside each other (the colour rounds' all-gathers are issued from the
        # copy stream): on a tiny call their sum can exceed the call - what exceeds it is reported, the prediction is capped
        raw = float(sum(parts.values()))
        fixed = min(raw, total_s)
        shrinking = max(total_s - fixed, 0.0)
        return {'predicted_fixed_s': round(fixed, 4), 'predicted_fixed_parts': {k: round(float(v), 5) for k, v in parts.items()},
                'parts_beside_each_other_s': round(raw - fixed, 5),
                'shrinking_s': round(shrinking, 4), 'fixed_share_of_total': round(fixed / total_s, 4) if total_s > 0 else None,
                'projected_total_s_at_twice_the_ranks': round(fixed + shrinking / 2.0, 4),
                'speedup_ceiling_vs_this_run': round(total_s / fixed, 2) if fixed > 0 else None}

    with tempfile.TemporaryDirectory() as td:
        scratch.append(td)
        # (a) sharded: identical inputs on every rank.  Rank 0 streams the OBJ and is the only rank that copies the mesh to the
        # host (result_on='rank0', the default)
        timings = {}

        def leg_a():
            recon.gen_mesh(args.genmesh_res, netMR, dev, data, os.path.join(td, f'warm{rank}.obj'), sharded=True)      # at the timed size
            rm(os.path.join(td, f'warm{rank}.obj'))
            if os.environ.get('PIFU_BENCH_FAIL_IN_LEG') == str(rank):
                raise RuntimeError('test hook: this rank fails inside the sharded leg')      # tests/test_gpu_dist.py
            torch.cuda.synchronize(dev)
            dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            recon.gen_mesh(args.genmesh_res, netMR, dev, data, os.path.join(td, f'mesh{rank}.obj'), timings=timings, sharded=True)
            torch.cuda.synchronize(dev)
            dist.barrier()
            dt = time.perf_counter() - t0
            rm(os.path.join(td, f'mesh{rank}.obj'))
            return dt
        t_tot, err = guarded('gen_mesh sharded', leg_a)
        leg_end('gen_mesh sharded', err, True)
        red = reduce_max([t_tot if err is None else float('inf')] + [timings.get(k, float('inf')) for k in keys] +
                         [float(timings.get('collective_s', 0.0))])
        if err is None and red[0] != float('inf'):
            res['one_subject_sharded'] = {'world': world, 'total_s': round(red[0], 4), **{k: round(v, 4) for k, v in zip(keys, red[1:])},
                                          'n_verts': timings.get('n_verts'), 'n_faces': timings.get('n_faces'),
                                          'result_on': 'rank0',
                                          # rank 0's own view of the exchange: one all-gather of the fp32 predictions per octree
                                          # level + one of the colours per round + the four broadcasts of the dealt-out encoders;
                                          # the time between the events around them includes waiting for the slowest rank; its
                                          # max over ranks beside it
                                          **{k: timings.get(k) for k in extra}, 'collective_s_max_over_ranks': round(red[-1], 5),
                                          **fixed_cost(timings, red[0], red[-1])}
            log(f"gen_mesh sharded: {red[0]:.3f} s")
        else:
            res['one_subject_sharded'] = {'error': err or 'failed on another rank'}
        # (b) independent subjects (different images: a failure may be local to one rank; the call has no collective, so a
        # failed rank simply waits for the others at the fences).  One untimed call with the rank's own image first: the first
        # call with a new mesh allocates (pinned result slabs), which is not the steady state of a recon() loop over many subjects
        recon.set_host_threads(host_threads)
        own = synth.synthetic_inputs(1024, args.image_channels, seed=rank)

        def warm_b():
            recon.gen_mesh(args.genmesh_res, netMR, dev, own, os.path.join(td, f'subjwarm{rank}.obj'))
            rm(os.path.join(td, f'subjwarm{rank}.obj'))
        _, err = guarded('independent subjects (warm-up)', warm_b)
        fence()

        def timed_b():
            t0 = time.perf_counter()
            recon.gen_mesh(args.genmesh_res, netMR, dev, own, os.path.join(td, f'subj{rank}.obj'))
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            rm(os.path.join(td, f'subj{rank}.obj'))
            return dt
        t_tot = None
        if err is None:
            t_tot, err = guarded('independent subjects', timed_b)
        recon.set_host_threads(0)
        leg_end('independent subjects', err, False)
        red = reduce_max([t_tot if err is None else float('inf')])
        if red[0] != float('inf'):
            res['independent_subjects'] = {'subjects': world, 'total_s': round(red[0], 4), 'subjects_per_s': round(world / red[0], 3),
                                           'warmed': True}
        else:
            res['independent_subjects'] = {'error': err or 'failed on another rank'}
        # (c) BASELINE configs[4]: ONE subject at 1024^3, octree levels sharded by chunk ranges with one all-gather per level,
        # marching cubes over one axis-0 slab per rank (dist.SLAB_MC_MIN_RESOLUTION = 1024 is reached), colour loop sharded
        # by calc_normal calls.  The smoother subject (MLP weights at torch's stock scale): its surface is what a trained
        # net's looks like, and its mesh stays in the millions of vertices at this resolution.
        if args.config5_res > 0:
            timings = {}
            # the legs above are done with the headline net: its batch workspace goes back to the device first (set and lift a
            # bound: pifu_set_workspace_limit releases an arena above it) - ranks that SHARE a card need the room for the
            # 1024^3 state (9.6 GB of SDF + mask, 11.5 GB of column tables, the slab's marching-cubes volumes, per rank)
            one_gpu = os.environ.get('PIFU_BENCH_ONE_GPU') == '1'
            if one_gpu:
                _lm = _lib_mod()
                _lm.call('pifu_set_workspace_limit', netMR._ctx.h, 1)
                _lm.call('pifu_set_workspace_limit', netMR._ctx.h, 0)
                _lm.call('pifu_enc_clear', netMR._ctx.h)      # ... and its encoders' lane arenas + packed weights (~19 GB): no leg filters with it again
                torch.cuda.empty_cache()
            net5, err = guarded('config 5 (building the smoother subject)',
                                lambda: smoother_subject(args, dev, data, share=(0.25 / world) if one_gpu else None))
            fence()

            def leg_c():
                recon.gen_mesh(args.config5_res, net5, dev, data, os.path.join(td, f'warm5_{rank}.obj'), sharded=True, result_on='all')
                rm(os.path.join(td, f'warm5_{rank}.obj'))
                torch.cuda.synchronize(dev)
                dist.barrier()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                mesh = recon.gen_mesh(args.config5_res, net5, dev, data, os.path.join(td, f'mesh5_{rank}.obj'), timings=timings,
                                      sharded=True, result_on='all')      # every rank returns the mesh: their digests are compared
                torch.cuda.synchronize(dev)
                dist.barrier()
                dt = time.perf_counter() - t0
                rm(os.path.join(td, f'mesh5_{rank}.obj'))
                return dt, mesh_digest(*mesh)
            # a rank without the subject cannot take part in the sharded call: its siblings would wait for it - leave at once
            if err is not None:
                leg_end('config 5', err, True)
                out_c = None                       # (every rank failed alike, else this rank has left: the error is reported)
            else:
                out_c, err = guarded('config 5', leg_c)
            leg_end('config 5', err, True)
            t_tot, sha = out_c if err is None else (float('inf'), 0)
            del net5
            red = reduce_max([t_tot] + [timings.get(k, float('inf')) for k in keys])
            # every rank must hold the same mesh: max(sha) == min(sha) over the ranks (56 bits of the sha256 of vertices,
            # faces and colours)
            t = torch.tensor([sha, -sha], device=dev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if err is None and red[0] != float('inf'):
                from pifuhd_amd import dist as pdist
                res['config5_1024'] = {
                    'resolution': args.config5_res, 'subject': 'smoother (mlp_weight_std 0.02)', 'total_s': round(red[0], 4),
                    **{k: round(v, 4) for k, v in zip(keys, red[1:])}, 'n_verts': timings.get('n_verts'),
                    'n_faces': timings.get('n_faces'), 'result_on': 'all', **{k: timings.get(k) for k in extra},
                    'marching_cubes': 'one axis-0 slab per rank' if args.config5_res >= pdist.SLAB_MC_MIN_RESOLUTION
                    else 'replicated (below dist.SLAB_MC_MIN_RESOLUTION)',
                    'mesh_identical_on_all_ranks': bool(int(t[0]) == -int(t[1])), 'mesh_sha256_56bit': '%014x' % int(t[0]),
                    **fixed_cost(timings, red[0], float(timings.get('collective_s', 0.0) or 0.0))}
                log(f"config 5: gen_mesh {args.config5_res}^3 sharded over {world} ranks: {red[0]:.3f} s, {timings.get('n_verts')} vertices")
            else:
                res['config5_1024'] = {'error': err or 'failed on another rank'}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--res', type=int, default=256, help='lattice points per axis per GPU slab')
    ap.add_argument('--chunk', type=int, default=5000)
    ap.add_argument('--batch-rows', type=int, default=2097152, help='point rows evaluated per batch of launches (workspace ~12.5 KB per row: 26 GB at the default)')
    ap.add_argument('--cpu-chunks', type=int, default=200, help='chunks of the workload timed on the CPU oracle')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='wall-time bound of the CPU baseline sample')
    ap.add_argument('--cpu-threads', type=int, default=32)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cpu-alt', action='store_true', help='skip the second CPU baseline run on all host threads')
    ap.add_argument('--no-genmesh', action='store_true')
    ap.add_argument('--no-lean-pass', action='store_true', help='skip the extra pass without the unread coarse tail')
    ap.add_argument('--dist-selftest', action='store_true',
                    help='functional check, never a measurement: initialise the process group (backend nccl = RCCL) even with ONE '
                         'rank and run every collective of the sharded path (all-gather of the field, of the colours, barrier)')
    ap.add_argument('--no-chunk100k', action='store_true', help='skip the extra pass in 100 000-point chunks')
    ap.add_argument('--leg-timeout', type=float, default=90.0,
                    help='N > 1: seconds a rank whose sharded gen_mesh call FAILED waits for its siblings before it leaves the job')
    ap.add_argument('--no-calc-normal-leg', action='store_true', help='skip the calc_normal measurement (pixel tables against per-point GEMMs)')
    ap.add_argument('--no-encoder-leg', action='store_true', help='skip the timed encoder passes (roofline_secondary: encoders)')
    ap.add_argument('--no-canonical', action='store_true', help='skip the extra canonical per-point pass (no column sharing)')
    ap.add_argument('--canonical-steps', type=int, default=2)
    ap.add_argument('--no-cpu-genmesh', action='store_true', help='skip the CPU (oracle) gen_mesh baseline at --cpu-genmesh-res')
    ap.add_argument('--cpu-genmesh-res', type=int, default=128)
    ap.add_argument('--cpu-genmesh-res2', type=int, default=256, help='second CPU gen_mesh baseline, run when the first took < 40 s')
    ap.add_argument('--no-split-precision', action='store_true', help='skip the extra bf16x3 / fp16x3 measurements')
    ap.add_argument('--genmesh-res', type=int, default=512)
    ap.add_argument('--genmesh-runs', type=int, default=3, help='timed gen_mesh calls; the median is reported')
    ap.add_argument('--config5-res', type=int, default=1024,
                    help='resolution of the BASELINE configs[4] leg (one subject, sharded over all ranks at N > 1; 0 = skip)')
    ap.add_argument('--no-smooth-genmesh', action='store_true', help='skip the second gen_mesh subject (stock-scale MLP weights)')
    ap.add_argument('--no-body-subject', action='store_true', help='skip the body-like gen_mesh subject (crafted smooth closed body)')
    ap.add_argument('--no-pifuhd-pt', action='store_true', help="skip the pifuhd.pt configuration (GroupNorm filters, MLP without norm layers)")
    ap.add_argument('--image-channels', type=int, default=6, choices=[3, 6],
                    help='6 = RGB-D input (RGB || depth, BASELINE config 2; SURVEY C-7), 3 = RGB as the reference is committed')
    ap.add_argument('--features', choices=['filter', 'synthetic'], default='filter',
                    help="feature maps from the real encoders (filter) or seeded random maps of the same shape")
    ap.add_argument('--lean', action='store_true', help='skip the coarse MLP layers whose output nobody reads')
    ap.add_argument('--no-depth-walk', action='store_true',
                    help='evaluate coarse layer 1 as a per-point GEMM instead of walking the columns along z')
    ap.add_argument('--no-column-sharing', action='store_true',
                    help='evaluate every lattice point independently (the canonical 2 916 098 FLOP per query)')
    args = ap.parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args.gpus))      # plain `python bench.py --gpus N`: start the ranks; nothing has touched the GPU yet

    import numpy as np
    import torch
    import torch.distributed as dist
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import _lib, config, dist as pdist, nets, synth
    from pifuhd_amd.nets import _stream_ptr

    t_start = time.perf_counter()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch multi-GPU runs with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N '
                             '--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...')
    # PIFU_BENCH_ONE_GPU=1 is a functional test hook: all ranks share cuda:0 and talk over gloo (RCCL refuses two ranks
    # on one device), so the N > 1 code path can be exercised on a single-GPU box; never a measurement
    one_gpu_test = os.environ.get('PIFU_BENCH_ONE_GPU') == '1'
    rccl_log = None
    dev = torch.device('cuda', 0 if one_gpu_test else local_rank)
    torch.cuda.set_device(dev)
    if world > 1 or args.dist_selftest:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        import datetime
        if one_gpu_test:
            dist.init_process_group('gloo', timeout=datetime.timedelta(minutes=20))
        else:
            # RCCL's own account of what it built (rings / trees, transport per channel, and - where the build logs it - the
            # algorithm and protocol it picks per collective): written by every rank to its own file, summarised once below
            rccl_log = os.path.join(tempfile.gettempdir(), f'pifu_rccl_{os.getpid()}_r{rank}.log')
            if os.environ.get('NCCL_DEBUG', '').upper() in ('', 'VERSION', 'WARN'):      # the image exports VERSION
                os.environ['NCCL_DEBUG'] = 'INFO'                                         # (the output goes to the file below)
            os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH,TUNING')
            os.environ.setdefault('NCCL_DEBUG_FILE', rccl_log)
            # a rank that dies takes the job down (torch.distributed.run ends its siblings); a rank that hangs is cut by this
            dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(minutes=20))
        if args.dist_selftest:
            pdist.ALWAYS_COLLECTIVE = True
        if args.dist_selftest or one_gpu_test:
            pdist.SHARDED_MIN_RESOLUTION = 0          # functional runs use small grids: they are to exchange, not to fall under the size guard

    # ---- model + inputs (outside the timed region) ---------------------------------------------------
    oc = config.coarse_opt(in_image_channels=args.image_channels)
    of = config.fine_opt(oc)
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
    netG, netMR = nets.build_nets(oc, of, sd, dev)
    def log(msg):
        print(f'[bench r{rank} {time.perf_counter() - t_start:7.1f}s] {msg}', file=sys.stderr, flush=True)
    log('weights ready')
    data = synth.synthetic_inputs(1024, args.image_channels, seed=0)      # one subject: every rank holds the same image + weights
    cold = {}
    if args.features == 'filter':
        # the first pass packs and uploads the encoder weights (1.5 GB: netF / netB 0.73 GB each) - what a one-image run_recon.py
        # style invocation pays; the second pass is the steady state
        torch.cuda.synchronize(dev)
        for key in ('first_encoder_pass_s', 'second_encoder_pass_s'):
            c0 = time.perf_counter()
            netMR.filter_global(data['img_512'].to(dev))
            netMR.filter_local(data['img'].to(dev)[:, None])
            torch.cuda.synchronize(dev)
            cold[key] = round(time.perf_counter() - c0, 4)
    else:
        gfe = torch.Generator().manual_seed(4242)
        netG.set_features(torch.randn(1, 256, 128, 128, generator=gfe))
        netMR.set_features(torch.randn(1, 16, 512, 512, generator=gfe))
    torch.cuda.synchronize(dev)
    log(f'feature maps ready ({args.features})')
    calib = data['calib']
    # synthetic checkpoint only: centre the 0.5 level set inside the volume (30 % of a 64^3 lattice inside)
    bias_shift = synth.calibrate_occupancy_bias(netMR, calib, sd, 0.3)
    torch.cuda.synchronize(dev)
    log(f'occupancy bias calibrated (shift {bias_shift:+.3f})')

    R = args.res
    res = (world * R, R, R)
    total = res[0] * res[1] * res[2]
    flags = 0 if args.lean else _lib.Q_FULL_COARSE
    if args.no_column_sharing:
        flags |= _lib.Q_NO_COLUMN_SHARING
    if args.no_depth_walk:
        flags |= _lib.Q_NO_DEPTH_WALK
    h = netMR._ctx.h
    # the batch workspace sized from what the device has free: the whole of --batch-rows on a GPU of its own (26 GB of 288),
    # a share of the card when ranks share it (PIFU_BENCH_ONE_GPU) - N contexts per device never run it out of memory
    rows_fit = C.c_int64(0)
    _lib.call('pifu_reserve_auto', h, args.batch_rows, 0.5 / (world if one_gpu_test else 1), C.byref(rows_fit))
    batch_rows = int(rows_fit.value)

    def fresh_eval(chunk, fl):
        _lib.call('pifu_drop_tables', h)      # every measured evaluation rebuilds its column tables, like step()
        return pdist.eval_lattice_sharded(netMR, calib, res, chunk, fl)

    def step():
        # a step pays for everything a subject's grid evaluation pays for: the per-column tables (sampled features + their
        # products with the skip weights), which the library would otherwise keep from the previous step, are dropped first
        _lib.call('pifu_drop_tables', h)
        return pdist.eval_lattice_sharded(netMR, calib, res, args.chunk, flags)

    grouped = world > 1 or args.dist_selftest       # a process group exists: barriers / reductions / sharded gen_mesh run

    def fence():
        torch.cuda.synchronize(dev)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if grouped and os.environ.get('PIFU_BENCH_FAIL_RANK') == str(rank):
        raise RuntimeError('test hook: this rank fails before its first collective')      # tests/test_gpu_dist.py
    for _ in range(args.warmup):
        step()
    fence()
    log('warm-up done')
    rccl_summary = rccl_choices(os.environ.get('NCCL_DEBUG_FILE') if rccl_log else None)
    if rccl_summary:
        log('RCCL: ' + ' | '.join(rccl_summary['lines'][:6]))
    reset_prof(_lib, h)
    _lib.call('pifu_profile_enable', h, 1)

    def clock_probe():
        buf = (C.c_uint64 * 16)()
        _lib.call('pifu_clock_probe', h, buf, _stream_ptr(dev))
        return list(buf)
    clk0 = clock_probe()            # per-XCD (shader ticks, 100 MHz real-time ticks) just outside the timed region
    fence()
    t0 = time.perf_counter()
    field = None
    for _ in range(args.steps):
        field = step()
    fence()
    elapsed = time.perf_counter() - t0
    clk1 = clock_probe()
    log(f'timed region done: {elapsed:.3f} s')
    # average shader clock over the timed region: shader-counter ticks per real-time tick (100 MHz), per XCD that
    # both probes sampled; s_memtime counts the clock the CUs actually ran at (power management included)
    eff_mhz = measured_mhz(clk0, clk1)
    _lib.call('pifu_profile_enable', h, 0)
    g_ms, g_n, g_fl = read_prof(_lib, h, _lib.PROF_GEMM)
    gemm_ms, gemm_n, gemm_flops = C.c_double(g_ms), C.c_int64(g_n), C.c_double(g_fl)
    wk_ms, wk_n, wk_bytes = read_prof(_lib, h, _lib.PROF_WALK)

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if grouped:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    value = total * args.steps / elapsed / 1e6

    # ---- second half of the metric at N > 1: one subject's 512^3 gen_mesh sharded over the ranks, and N independent
    # subjects (one per GPU, BASELINE configs[3]); every rank takes part, rank 0 reports
    multi = None
    if grouped and not args.no_genmesh and args.features == 'filter':
        multi = genmesh_multi(args, netMR, dev, data, synth, dist, rank, world, fence, log)
    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    achieved = gemm_flops.value / (gemm_ms.value * 1e-3) / 1e12 if gemm_ms.value > 0 else 0.0
    # HBM bytes per launch from the PMC counters cannot be collected inside this process (rocprofv3 --pmc is its own run):
    # the figure comes from the committed profile of an EARLIER run of this command and is labelled as such
    traffic, traffic_src, prof_json = None, None, {}
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        try:
            prof_json = json.load(open(tpath))
            traffic = prof_json.get('k_mlp_gemm_hbm_bytes_per_launch')
            traffic_src = 'profiles/traffic.json (rocprofv3 --pmc pass of an earlier run of this command, not this run)'
        except Exception:
            traffic = None
    out = {
        'metric': 'occupancy_queries_throughput', 'value': round(value, 3), 'unit': 'Mqueries/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': f'dense {res[0]}x{res[1]}x{res[2]} lattice ({R}^3 per GPU, axis-0 chunk-range slabs, RCCL '
                        f'all-gather of the fp32 field), coarse+fine PIFuMRNet query, chunk {args.chunk}, '
                        f'1024px {"RGB-D (6-channel)" if args.image_channels == 6 else "RGB"} image + 512px, name-seeded weights',
            'queries_per_step': total, 'chunk': args.chunk, 'batch_rows': batch_rows, 'batch_rows_requested': args.batch_rows, 'coarse_tail': 'skipped' if args.lean else 'computed',
            'column_sharing': not args.no_column_sharing,
            'depth_walk': not (args.no_depth_walk or args.no_column_sharing),
            'flop_per_query_canonical': FLOP_PER_QUERY,
            # what the GEMM launches of the timed region actually executed (column sharing removes the per-point
            # image-feature products: 2 163 200 FLOP per query + the per-column tables)
            'gemm_flop_per_query_executed': round(gemm_flops.value * world / max(total * args.steps, 1)),   # rank 0's launches x N
        },
        'executed_gemm_tflops_over_step': round(gemm_flops.value * world / elapsed / 1e12, 2),
        'canonical_equivalent_tflops': round(value * 1e6 * FLOP_PER_QUERY / 1e12, 2),
        'roofline': {
            'kernel': 'k_mlp_gemm_tri', 'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_FP32_MATRIX_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_FP32_MATRIX_TFLOPS, 4), 'traffic': traffic,
            'traffic_source': traffic_src,
            'launches': gemm_n.value, 'avg_launch_ms': round(gemm_ms.value / max(gemm_n.value, 1), 4),
            'alg_gflop_per_launch': round(gemm_flops.value / max(gemm_n.value, 1) / 1e9, 3),
            'gemm_time_share_of_step': round(gemm_ms.value * 1e-3 / elapsed, 4),
            # measured beside the nominal-clock figures above: the peak assumes 2 400 MHz, the fp32-MFMA load runs lower
            'shader_clock_mhz_measured': round(eff_mhz, 1) if eff_mhz else None,
            'frac_at_measured_clock': round(achieved / (PEAK_FP32_MATRIX_TFLOPS * eff_mhz / 2400.0), 4) if eff_mhz else None,
        },
    }

    # ---- the canonical per-point evaluation (no column sharing, no depth walk: every one of the 2 916 098 FLOP of a query
    # is executed, and the per-point sampler k_sample runs on every point): reported beside the headline, never as `value`
    secondary = []
    if wk_n > 0 and wk_ms > 0:
        # the second kernel of the headline step (5 % of it; 8 % of a 512^3 gen_mesh): the depth walk of coarse layer 1
        ach = wk_bytes / (wk_ms * 1e-3) / 1e9
        secondary.append({
            'kernel': 'k_walk<16, 2> (depth walk: coarse layer 1 of the headline step, replaces a 1024 x 512 GEMM per point)',
            'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(ach / PEAK_HBM_GBS, 4),
            'launches': wk_n, 'avg_launch_ms': round(wk_ms / wk_n, 4), 'alg_bytes_per_launch': round(wk_bytes / wk_n),
            'alg_bytes_definition': 'rows x 512 fp32 outputs written + rows depth features read',
            'frac_of_achievable_write_rate_6300': round(ach / 6300.0, 4),
            'traffic': (prof_json.get('detail', {}).get('k_walk') or {}).get('hbm_bytes_per_launch'),
            'traffic_source': traffic_src,
            'note': 'not at the HBM rate: ~1 000 sign flips per 256-row column segment with the name-seeded weights, each a rank-1 '
                    'update of (P, R) in double per output that fetches a 2 KB row of W1^T from L2 (~17 GB of L2 reads per launch): '
                    'the kernel sits on the L2 (tools/walk_variants.py: outputs per thread / prefetch depth do not move it)'})
    if world == 1 and not args.no_canonical and not args.no_column_sharing:
        try:
            cflags = flags | _lib.Q_NO_COLUMN_SHARING
            fresh_eval(args.chunk, cflags)
            fence()
            reset_prof(_lib, h)
            _lib.call('pifu_profile_enable', h, 1)
            ck0 = clock_probe()
            fence()
            c0 = time.perf_counter()
            for _ in range(args.canonical_steps):
                field_c = fresh_eval(args.chunk, cflags)
            fence()
            c_el = time.perf_counter() - c0
            c_mhz = measured_mhz(ck0, clock_probe())
            _lib.call('pifu_profile_enable', h, 0)
            cg_ms, cg_n, cg_fl = read_prof(_lib, h, _lib.PROF_GEMM)
            cs_ms, cs_n, cs_by = read_prof(_lib, h, _lib.PROF_SAMPLER)
            c_val = total * args.canonical_steps / c_el / 1e6
            c_gemm = cg_fl / (cg_ms * 1e-3) / 1e12 if cg_ms > 0 else 0.0
            out['canonical_path'] = {
                'value': round(c_val, 3), 'unit': 'Mqueries/s', 'steps': args.canonical_steps,
                'ms_per_step': round(c_el / max(args.canonical_steps, 1) * 1e3, 3),
                'flop_per_query_executed': FLOP_PER_QUERY,
                'whole_path_tflops': round(c_val * 1e6 * FLOP_PER_QUERY / 1e12, 2),
                'whole_path_frac_of_fp32_matrix_peak': round(c_val * 1e6 * FLOP_PER_QUERY / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
                'gemm_roofline': {'kernel': 'k_mlp_gemm_tri', 'achieved': round(c_gemm, 2), 'peak': PEAK_FP32_MATRIX_TFLOPS,
                                  'unit': 'TFLOP/s', 'frac': round(c_gemm / PEAK_FP32_MATRIX_TFLOPS, 4), 'launches': cg_n,
                                  'shader_clock_mhz_measured': round(c_mhz, 1) if c_mhz else None,
                                  'frac_at_measured_clock': round(c_gemm / (PEAK_FP32_MATRIX_TFLOPS * c_mhz / 2400.0), 4) if c_mhz else None},
                'max_abs_diff_vs_headline_field': float((field_c - field).abs().max().item())}
            log(f"canonical per-point path: {c_val:.1f} Mq/s, GEMM {c_gemm:.1f} TFLOP/s")
            if cs_ms > 0:
                # the two feature maps (coarse 128^2 x 256 + fine 512^2 x 16 fp32 = 33.6 MB) stay in the 256 MB Infinity
                # Cache, so the gathers are priced against the cache-gather rate of the guide, not against HBM; what HBM
                # must move is the MLP input rows written (and the maps once)
                rows_bytes = float(cs_by)
                per_launch = rows_bytes / max(cs_n, 1)
                gather_share = 4.0 * (256 + 16) * 4.0 / (4.0 * (256 + 16) * 4.0 + (288 + 32) * 4.0)
                ach = rows_bytes / (cs_ms * 1e-3) / 1e9
                map_bytes = float(sum(t.numel() * 4 for t in (netG.im_feat_list[-1], netMR.im_feat_list[-1])))
                compulsory = per_launch * (1.0 - gather_share) + map_bytes
                pmc = prof_json.get('k_sample_hbm_bytes_per_launch')
                secondary.append({
                    'kernel': 'k_sample (per-point sampler of the canonical path)', 'bound': 'infinity_cache',
                    'achieved': round(ach, 1), 'peak': PEAK_INFINITY_CACHE_GATHER_GBS, 'unit': 'GB/s',
                    'frac': round(ach / PEAK_INFINITY_CACHE_GATHER_GBS, 4),
                    'peak_source': 'MI355X_MICROARCH.md, indexed rows of a 38 MB table served by the Infinity Cache',
                    'launches': cs_n, 'avg_launch_ms': round(cs_ms / max(cs_n, 1), 4),
                    'alg_bytes_per_launch': round(per_launch), 'gather_share_of_alg_bytes': round(gather_share, 3),
                    'compulsory_hbm_bytes_per_launch': round(compulsory),
                    'alg_over_compulsory': round(per_launch / compulsory, 2),
                    'traffic': pmc, 'traffic_source': traffic_src if pmc else None})
        except Exception as e:      # the metric line must still be printed
            out['canonical_path'] = {'error': repr(e)}
        finally:
            _lib.call('pifu_profile_enable', h, 0)

    # ---- the colour loop's kernel path on its own: calc_normal on seeded vertices in gen_mesh's 50 000-vertex calls (each one
    # GroupNorm batch of 200 000 point rows), through the pixel tables (default for calls of this size) and through the
    # per-point GEMMs (pifu_set_pixel_tables(ctx, 0): every FLOP of a lean query executed), alternating
    if world == 1 and not args.no_canonical and not args.no_calc_normal_leg:
        try:
            nv = 500000
            gv = torch.Generator().manual_seed(5)
            verts = ((torch.rand(3, nv, generator=gv) * 2 - 1) * 0.9).to(dev).contiguous()
            cal32 = np.ascontiguousarray(calib[0].numpy().astype(np.float32))
            nm_out = torch.empty(3, nv, device=dev)
            pred4 = {m: torch.empty(4 * nv, device=dev) for m in (_lib.PIX_AUTO, _lib.PIX_OFF)}

            def normals_pass(mode):
                _lib.call('pifu_set_pixel_tables', h, mode)
                fence()
                reset_prof(_lib, h)
                _lib.call('pifu_profile_enable', h, 1)
                n0 = time.perf_counter()
                _lib.call('pifu_calc_normal_chunked', h, C.c_void_p(verts.data_ptr()), nv, 50000, cal32.ctypes.data_as(C.c_void_p),
                          0.001, C.c_void_p(nm_out.data_ptr()), C.c_void_p(pred4[mode].data_ptr()), _stream_ptr(dev))
                fence()
                dt = time.perf_counter() - n0
                _lib.call('pifu_profile_enable', h, 0)
                ms, _, fl = read_prof(_lib, h, _lib.PROF_GEMM)
                return dt, ms, fl
            res_n = {}
            for mode, key in ((_lib.PIX_AUTO, 'pixel_tables'), (_lib.PIX_OFF, 'per_point_gemms')):
                normals_pass(mode)
                runs = sorted(normals_pass(mode) for _ in range(3))
                dt, ms, fl = runs[1]
                res_n[key] = {'value': round(4 * nv / dt / 1e6, 2), 'unit': 'Mqueries/s', 'ms': round(dt * 1e3, 2),
                              'gemm_flop_per_query_executed': round(fl / (4 * nv)),
                              'gemm_tflops': round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else None}
            out['calc_normal'] = {'vertices': nv, 'queries': 4 * nv, 'call_size': 50000, **res_n,
                                  'max_abs_diff_between_paths': float((pred4[_lib.PIX_AUTO] - pred4[_lib.PIX_OFF]).abs().max().item())}
            log(f"calc_normal: {res_n['pixel_tables']['value']} Mq/s through the pixel tables, {res_n['per_point_gemms']['value']} per point")
        except Exception as e:
            out['calc_normal'] = {'error': repr(e)}
        finally:
            _lib.call('pifu_profile_enable', h, 0)
            _lib.call('pifu_set_pixel_tables', h, _lib.PIX_AUTO)

    # ---- the same lattice without the coarse MLP's tail (layers 3-4 + sigmoid produce netG.preds, which nothing on the
    # gen_mesh path reads: reconstruction() runs this way); reported beside the headline, never as `value`
    if world == 1 and not args.lean and not args.no_lean_pass:
        try:
            lean_flags = flags & ~_lib.Q_FULL_COARSE
            fresh_eval(args.chunk, lean_flags)
            fence()
            l0 = time.perf_counter()
            for _ in range(args.steps):
                field_lean = fresh_eval(args.chunk, lean_flags)
            fence()
            l_el = time.perf_counter() - l0
            out['without_unread_coarse_tail'] = {
                'value': round(total * args.steps / l_el / 1e6, 3), 'unit': 'Mqueries/s',
                'ms_per_step': round(l_el / max(args.steps, 1) * 1e3, 3),
                'strictly_necessary_flop_per_query': 2784000,
                'field_identical_to_headline_run': bool(torch.equal(field_lean, field))}
            log(f"without the unread coarse tail: {out['without_unread_coarse_tail']['value']} Mq/s")
        except Exception as e:
            out['without_unread_coarse_tail'] = {'error': repr(e)}

    # ---- the same lattice in the 100 000-point query() calls of the image-coloured variants (`reconstruction.py:108,160`:
    # gen_mesh_imgColor / _plus pass num_samples=100000; GroupNorm then spans 100 000 points, so it is a different field)
    if world == 1 and not args.no_chunk100k:
        try:
            fresh_eval(100000, flags)
            fence()
            reset_prof(_lib, h)
            _lib.call('pifu_profile_enable', h, 1)
            k0 = time.perf_counter()
            for _ in range(args.steps):
                fresh_eval(100000, flags)
            fence()
            k_el = time.perf_counter() - k0
            _lib.call('pifu_profile_enable', h, 0)
            kg_ms, kg_n, kg_fl = read_prof(_lib, h, _lib.PROF_GEMM)
            out['chunk_100000'] = {'value': round(total * args.steps / k_el / 1e6, 3), 'unit': 'Mqueries/s',
                                   'ms_per_step': round(k_el / max(args.steps, 1) * 1e3, 3), 'chunk': 100000,
                                   'gemm_algorithmic_tflops': round(kg_fl / (kg_ms * 1e-3) / 1e12, 2) if kg_ms > 0 else None}
            log(f"chunk 100000: {out['chunk_100000']['value']} Mq/s")
        except Exception as e:
            out['chunk_100000'] = {'error': repr(e)}
        finally:
            _lib.call('pifu_profile_enable', h, 0)

    # ---- opt-in split precision on the 16-bit matrix cores (bf16 x 3, fp16 x 3): reported beside the headline, never as `value`
    split_fields = {}
    if world == 1 and not args.no_split_precision:
        arith = {1: 'bf16x3 split (hi*hi + lo*hi + hi*lo, fp32 accumulate) on v_mfma_f32_32x32x16_bf16; opt-in via pifu_set_precision(ctx, 1)',
                 2: 'fp16x3 split (same three products on v_mfma_f32_32x32x16_f16, weight planes scaled by 2^11: operands good to 2^-22); '
                    'opt-in via pifu_set_precision(ctx, 2)'}
        for mode, key in ((1, 'split_precision'), (2, 'split_precision_fp16x3')):
            _lib.call('pifu_set_precision', h, mode)
            try:
                step()
                fence()
                reset_prof(_lib, h)
                _lib.call('pifu_profile_enable', h, 1)
                s0 = time.perf_counter()
                field_split = None
                for _ in range(args.steps):
                    field_split = step()
                fence()
                s_el = time.perf_counter() - s0
                _lib.call('pifu_profile_enable', h, 0)
                s_ms, s_n, s_fl = read_prof(_lib, h, _lib.PROF_GEMM)
                sm, sn, sf = C.c_double(s_ms), C.c_int64(s_n), C.c_double(s_fl)
                split_fields[key] = field_split
                out[key] = {'arithmetic': arith[mode],
                            'value': round(total * args.steps / s_el / 1e6, 3), 'unit': 'Mqueries/s',
                            'ms_per_step': round(s_el / max(args.steps, 1) * 1e3, 3),
                            'gemm_algorithmic_tflops': round(sf.value / (sm.value * 1e-3) / 1e12, 2) if sm.value > 0 else None,
                            'max_abs_diff_vs_fp32_path': float((field_split - field).abs().max().item())}
                log(f"{key}: {out[key]['value']} Mq/s, max diff {out[key]['max_abs_diff_vs_fp32_path']:.2e}")
            except Exception as e:      # the metric line must still be printed
                out[key] = {'error': repr(e)}
            finally:
                _lib.call('pifu_profile_enable', h, 0)
                _lib.call('pifu_set_precision', h, 0)

    # ---- the encoder pass (filter_global + filter_local of one image, the overlapped schedule gen_mesh uses): counted
    # convolution FLOPs / wall time of a steady-state pass against the fp32-matrix peak
    if world == 1 and args.features == 'filter' and not args.no_encoder_leg:
        try:
            from pifuhd_amd import reconstruction as recon_enc
            recon_enc._Subject(netMR, data)
            fence()
            reset_prof(_lib, h)
            _lib.call('pifu_profile_enable', h, 1)
            passes = []
            for _ in range(5):
                e0 = time.perf_counter()
                recon_enc._Subject(netMR, data)
                torch.cuda.synchronize(dev)
                passes.append(time.perf_counter() - e0)
            _lib.call('pifu_profile_enable', h, 0)
            _, en_n, en_fl = read_prof(_lib, h, _lib.PROF_ENC)
            _, _, en_canon = read_prof(_lib, h, _lib.PROF_ENC_CANONICAL)
            en_fl /= len(passes)
            en_canon /= len(passes)
            p_med = sorted(passes)[len(passes) // 2]
            secondary.append({
                'kernel': 'encoders: k_conv_dma<64|32> (+ k_prep_act, statistics, split-K reduce) of filter_global + filter_local, one image',
                'bound': 'mfma', 'achieved': round(en_fl / p_med / 1e12, 2), 'peak': PEAK_FP32_MATRIX_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(en_fl / p_med / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4), 'pass_ms': round(p_med * 1e3, 3),
                'passes_ms': [round(p * 1e3, 3) for p in passes], 'conv_launches_per_pass': en_n // len(passes),
                'alg_tflop_per_pass': round(en_fl / 1e12, 4),
                # the 36 convolutions of the normal nets' ResnetBlock trunks (1024 -> 1024, 3 x 3, 32 x 32) run as Winograd
                # F(2x2, 3x3): `achieved` / `frac` price what is EXECUTED; as direct convolutions the same pass would be
                'direct_convolution_tflop_per_pass': round(en_canon / 1e12, 4),
                'direct_convolution_equivalent_frac': round(en_canon / p_med / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
                'definition': 'host image -> both feature maps resident (upload of the 1024 px image included), wall clock of a '
                              'steady-state pass; FLOPs = 2 * output pixels * Cout * Cin * taps summed over the convolutions (2 * 16 * '
                              'tiles * Cout * Cin for the Winograd ones), counted by the library'})
            cold['steady_encoder_pass_s'] = round(p_med, 4)
            log(f"encoders: {p_med * 1e3:.2f} ms per image, {en_fl / p_med / 1e12:.1f} TFLOP/s")
        except Exception as e:
            secondary.append({'kernel': 'encoders', 'error': repr(e)})
        finally:
            _lib.call('pifu_profile_enable', h, 0)

    # ---- parity spot check + CPU baseline (oracle = checker; never the thing measured as `value`) -----
    if world == 1 and not args.no_cpu_baseline:
        from oracle import pifu_oracle as O
        # torch-CPU scales badly past a few dozen threads on these small per-chunk GEMMs (a 256-thread run
        # of this oracle is >10x slower than a 32-thread one), so the baseline uses at most 32 threads and
        # says so in `cores`; the sample is bounded by wall time (~args.cpu_seconds), not by a chunk count.
        ncores = min(os.cpu_count() or 1, args.cpu_threads)
        ref = O.OracleNets(sd, oc, of, fine_training=True)
        ref.feat_coarse = netG.im_feat_list[-1].cpu()
        ref.feat_fine = netMR.im_feat_list[-1].cpu()
        calib_inv = np.linalg.inv(calib[0].numpy().astype(np.float64))
        first = (total // 2 // args.chunk) * args.chunk            # a slab from the middle of the grid

        def time_oracle(nthreads, max_chunks, seconds):
            """The same chunks of the lattice, from `first` on, through the oracle on `nthreads` torch threads."""
            torch.set_num_threads(nthreads)
            vals, npts, cpu_s = [], 0, 0.0
            with torch.no_grad():
                while npts < max_chunks * args.chunk and first + npts < total:
                    n_c = min(args.chunk, total - first - npts)
                    flat = np.arange(first + npts, first + npts + n_c, dtype=np.int64)
                    pts = torch.from_numpy(O.grid_points_rect(res, calib_inv, flat).astype(np.float32))[None]
                    c0 = time.perf_counter()
                    vals.append(ref.query(pts, calib)[0, 0])
                    cpu_s += time.perf_counter() - c0
                    npts += n_c
                    if cpu_s > seconds and npts >= 2 * args.chunk:
                        break
            return torch.cat(vals).numpy(), npts, cpu_s
        want, npts, cpu_s = time_oracle(ncores, args.cpu_chunks, args.cpu_seconds)
        log(f'cpu baseline done: {npts} queries in {cpu_s:.1f} s on {ncores} threads')
        got = field[first:first + npts].cpu().numpy()
        out['cpu_baseline'] = {
            'value': round(npts / cpu_s / 1e6, 4), 'unit': 'Mqueries/s', 'cores': ncores,
            'kind': 'port', 'sample': f'{npts} queries ({npts // args.chunk} chunks of {args.chunk}) from the middle of '
                                      f'the same lattice, torch-CPU oracle (oracle/pifu_oracle.py), {cpu_s:.1f} s, '
                                      f'{ncores} of {os.cpu_count()} host threads',
        }
        # the same sample (at most the same chunks, at most the same wall time) on EVERY hardware thread of the host: the
        # reader can check on this machine that the bounded thread count above is the faster choice for 5 000-point GEMMs
        all_cores = os.cpu_count() or 1
        if all_cores != ncores and not args.no_cpu_alt:
            _, npts_alt, cpu_s_alt = time_oracle(all_cores, npts // args.chunk, args.cpu_seconds)
            torch.set_num_threads(ncores)
            out['cpu_baseline'].update({
                'cores_alt': all_cores, 'value_alt': round(npts_alt / cpu_s_alt / 1e6, 6),
                'sample_alt': f'{npts_alt} queries of the same sample in {cpu_s_alt:.1f} s on all {all_cores} host threads',
                'faster_choice': 'cores' if npts / cpu_s >= npts_alt / cpu_s_alt else 'cores_alt'})
            log(f'cpu baseline on all {all_cores} threads: {npts_alt} queries in {cpu_s_alt:.1f} s')
        out['parity_check_max_abs_diff'] = float(np.abs(got - want).max())
        for key, fld in split_fields.items():      # the opt-in arithmetics against the same oracle sample
            out[key]['max_abs_diff_vs_cpu_oracle'] = float(np.abs(fld[first:first + npts].cpu().numpy() - want).max())

    # ---- second half of the metric: end-to-end gen_mesh wall time (single GPU) ---------------------------
    if world == 1 and not args.no_genmesh and args.features == 'filter':
        from pifuhd_amd import mesh_util, reconstruction as recon
        with tempfile.TemporaryDirectory() as td:
            timings = {}

            def drop(name):      # a 512^3 OBJ is ~0.5 GB, a 1024^3 one ~0.8 GB: gone as soon as its call has been timed
                try:
                    os.remove(os.path.join(td, name))
                except OSError:
                    pass
            try:
                gm, timings, extra, nruns = gen_mesh_on_roofline(recon, _lib, netMR, dev, data, args.genmesh_res, args.genmesh_runs, td,
                                                                 'headline', log, cold)
                out['gen_mesh'] = gm
                out['gen_mesh'].update(octree_share(mesh_util, netMR, calib, args.genmesh_res))
                mc_ms, mc_n, mc_bytes = extra['mc']
                mc_ms, mc_bytes = mc_ms / nruns, mc_bytes / nruns
                nd_ms, nd_n, nd_bytes = extra['mc_nodes']
                nd_ms, nd_bytes = nd_ms / max(nd_n, 1), nd_bytes / max(nd_n, 1)
                if mc_ms > 0:
                    ach = mc_bytes / (mc_ms * 1e-3) / 1e9
                    secondary.append({
                        'kernel': 'marching cubes (k_mc_nodebricks / brick_combine / groups / count / scan + k_mc_verts / faces) of the gen_mesh runs, per call',
                        'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': round(ach / PEAK_HBM_GBS, 4), 'ms': round(mc_ms, 3),
                        'alg_bytes': round(mc_bytes), 'alg_bytes_definition': 'one pass over the float64 SDF (R^3 x 8 B) + 12 B per '
                        'vertex and per face written',
                        # counter passes of tools/mc_prof.sh on a 250 k-vertex 512^3 body (not this run's volume): FETCH_SIZE
                        # doubled (gfx950) + WRITE_SIZE, summed over the marching-cubes kernels of one call
                        'traffic': prof_json.get('mc_hbm_bytes_per_call_512_body'),
                        'traffic_source': 'profiles/traffic.json (tools/mc_prof.sh PMC passes on a 512^3 analytic body, an earlier run)'
                        if prof_json.get('mc_hbm_bytes_per_call_512_body') else None,
                        'note': 'only the node pass below streams the volume; the classification works on the cells of the bricks that '
                                'hold surface (4.2 M surface cells on this subject) and is latency-bound (k_mc_count), so the HBM '
                                'fraction of the whole is not a statement about bandwidth use'})
                if nd_ms > 0:
                    ach = nd_bytes / (nd_ms * 1e-3) / 1e9
                    secondary.append({
                        'kernel': 'k_mc_nodebricks (marching cubes, node pass: the one kernel that reads the whole float64 SDF)',
                        'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(ach / PEAK_HBM_GBS, 4),
                        'avg_launch_ms': round(nd_ms, 4), 'launches': nd_n, 'alg_bytes_per_launch': round(nd_bytes),
                        'alg_bytes_definition': 'R^3 x 8 B read once (value range + which 9^3-node bricks can hold surface)',
                        'traffic': None})
                # parity of the mesh the timed run produced: the same grid once more (not timed), marching cubes on a 128^3
                # crop around the surface on the GPU and on the serial oracle - vertices and faces must agree bit for bit
                r = mesh_util.reconstruction(netMR, dev, calib, args.genmesh_res, None, None, 0.5, use_octree=True,
                                             num_samples=5000, return_sdf=True)
                if r != -1:
                    from oracle import pifu_oracle as O
                    verts_r, faces_r, _, _, sdf_d = r
                    same_mesh = len(verts_r) == timings.get('n_verts') and len(faces_r) == timings.get('n_faces')
                    inside = (sdf_d > 0.5).nonzero()
                    ctr = [int(inside[:, k].float().mean().item()) for k in range(3)] if len(inside) else [args.genmesh_res // 2] * 3
                    cw = min(128, args.genmesh_res)
                    lo = [min(max(c - cw // 2, 0), args.genmesh_res - cw) for c in ctr]
                    crop = sdf_d[lo[0]:lo[0] + cw, lo[1]:lo[1] + cw, lo[2]:lo[2] + cw].contiguous()
                    gv, gf = mesh_util.marching_cubes_device(netMR._ctx, crop, 0.5)
                    ov, of_ = O.marching_cubes_lewiner(crop.cpu().numpy(), 0.5)
                    out['gen_mesh']['mc_check'] = {
                        'crop': [lo, cw], 'n_verts': int(len(gv)), 'n_faces': int(len(gf)),
                        'faces_bit_exact_vs_oracle': bool(np.array_equal(gf.cpu().numpy(), of_)),
                        'verts_bit_exact_vs_oracle': bool(np.array_equal(gv.cpu().numpy().view(np.uint32), ov.view(np.uint32))),
                        'rerun_reproduces_mesh_size': bool(same_mesh)}
                    log(f"mc check: {out['gen_mesh']['mc_check']}")
            except Exception as e:      # the metric line must still be printed
                out['gen_mesh'] = {'error': repr(e)}
            finally:
                _lib.call('pifu_profile_enable', h, 0)
            # the same call with the MLP GEMMs on the opt-in fp16 x 3 split products (pifu_set_precision(ctx, 2): operands good
            # to 2^-22, fp32 accumulation; ~3e-6 from the exact-fp32 path on the occupancy).  A side number, never the headline:
            # the octree picks its points from values that differ in the sixth decimal, so the mesh may differ by a few cells
            if not args.no_split_precision and 'error' not in out['gen_mesh']:
                try:
                    _lib.call('pifu_set_precision', h, 2)
                    recon.gen_mesh(args.genmesh_res, netMR, dev, data, os.path.join(td, 'warm_fp16x3.obj'))
                    drop('warm_fp16x3.obj')
                    ts = {}
                    recon.gen_mesh(args.genmesh_res, netMR, dev, data, os.path.join(td, 'mesh_fp16x3.obj'), timings=ts)
                    drop('mesh_fp16x3.obj')
                    out['gen_mesh']['opt_in_fp16x3'] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ts.items()}
                    log(f"gen_mesh with fp16x3 split products (opt-in): {ts.get('total_s'):.3f} s, {ts.get('n_verts')} vertices")
                except Exception as e:
                    out['gen_mesh']['opt_in_fp16x3'] = {'error': repr(e)}
                finally:
                    _lib.call('pifu_set_precision', h, 0)
            # the same call on a SMOOTHER subject: the MLP weights drawn at torch's stock scale (std 0.02 instead of the 0.16
            # that gives the headline subject its rough, multi-million-vertex surface).  Fewer points survive the octree and
            # fewer vertices need normals, so the per-image work (encoders) and the fixed costs weigh more - the other end of
            # what config 3 can look like.  Reported beside the headline subject, never instead of it.
            if not args.no_smooth_genmesh and 'error' not in out['gen_mesh']:
                try:
                    netMR2 = smoother_subject(args, dev, data)
                    # warm-up at the timed size: this context's workspaces grow here, not inside the timed call
                    recon.gen_mesh(args.genmesh_res, netMR2, dev, data, os.path.join(td, 'warm2.obj'))
                    drop('warm2.obj')
                    t2 = {}
                    recon.gen_mesh(args.genmesh_res, netMR2, dev, data, os.path.join(td, 'mesh2.obj'), timings=t2)
                    drop('mesh2.obj')
                    out['gen_mesh']['smoother_subject'] = {
                        'mlp_weight_std': 0.02, **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in t2.items()}}
                    log(f"gen_mesh, smoother subject: {t2.get('total_s'):.3f} s, {t2.get('n_verts')} vertices")
                    # BASELINE configs[4]'s grid on this ONE GPU (the N > 1 line shards the same call over the ranks)
                    if args.config5_res > 0:
                        try:
                            recon.gen_mesh(args.config5_res, netMR2, dev, data, os.path.join(td, 'warm5.obj'))
                            drop('warm5.obj')
                            t5 = {}
                            m5 = recon.gen_mesh(args.config5_res, netMR2, dev, data, os.path.join(td, 'mesh5.obj'), timings=t5)
                            drop('mesh5.obj')
                            out['gen_mesh']['config5_1024'] = {
                                'resolution': args.config5_res, 'subject': 'smoother (mlp_weight_std 0.02)', 'n_gpus': 1,
                                **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in t5.items()},
                                'mesh_sha256_56bit': '%014x' % mesh_digest(*m5)}
                            log(f"config 5 grid on one GPU: gen_mesh {args.config5_res}^3 {t5.get('total_s'):.3f} s, {t5.get('n_verts')} vertices")
                            del m5
                        except Exception as e:
                            out['gen_mesh']['config5_1024'] = {'error': repr(e)}
                    del netMR2
                except Exception as e:
                    out['gen_mesh']['smoother_subject'] = {'error': repr(e)}

            # ---- the BODY-LIKE subject (what a trained net produces): one smooth closed body, a few hundred thousand vertices, an
            # octree that refines a thin shell of the grid.  Encoders, per-level synchronisations and small launches weigh here
            # what the GEMMs weigh on the rough headline subject; the same roofline block prices them.
            if not args.no_body_subject and 'error' not in out['gen_mesh']:
                try:
                    _, net_b, data_b, fit_b = body_subject(args, dev, 'group', 'batch', 0.16)
                    gb, tb, _, _ = gen_mesh_on_roofline(recon, _lib, net_b, dev, data_b, args.genmesh_res, args.genmesh_runs, td, 'body', log)
                    gb.update(octree_share(mesh_util, net_b, data_b['calib'], args.genmesh_res))
                    mb = recon.gen_mesh(args.genmesh_res, net_b, dev, data_b, os.path.join(td, 'body_check.obj'))
                    drop('body_check.obj')
                    gb['mesh'] = mesh_components(net_b, mb[1])
                    gb['subject'] = ('synthetic body image through the encoders; MLPs = name-seeded weights (std 0.16, GroupNorm) with '
                                     'the wire channels of synth.craft_body_mlps: |z| < 0.3 * H(x, y), H a ridge read-out of the coarse '
                                     'feature map fitted to the silhouette\'s ellipsoidal profile')
                    gb['fit'] = {k: round(v, 4) for k, v in fit_b.items()}
                    out['gen_mesh']['body_subject'] = gb
                    del net_b, mb
                except Exception as e:
                    out['gen_mesh']['body_subject'] = {'error': repr(e)}
            # ---- the configuration of the published pifuhd.pt (what `recon()` loads, reconstruction.py:233-292):
            # GroupNorm HGFilters and an MLP WITHOUT norm layers (MLP.py:37-41,65-69).  A point's occupancy then does not depend
            # on the query() call it is evaluated in: no statistics passes, the lattice cut into whole tiles / whole columns
            # instead of 5 000-point chunks (pifu_effective_chunk), raw Z-slabs across ranks.  Both halves of the metric: M1 on
            # the 256^3 lattice (name-seeded weights, std 0.10: the parity fixture e2e_full_pifuhd's), M2 at 512^3 on the
            # body-like subject, the same roofline blocks.
            if not args.no_pifuhd_pt:
                try:
                    ocp = config.coarse_opt(in_image_channels=args.image_channels, mlp_norm='no', norm='group')
                    ofp = config.fine_opt(ocp)
                    sdp = synth.name_seeded_state_dict(synth.state_dict_spec(ocp, ofp), seed=0, mlp_std=0.10)
                    netGp, netp = nets.build_nets(ocp, ofp, sdp, dev)
                    hp = netp._ctx.h
                    rows_p = reserve_share(_lib, netp, args.batch_rows, 0.25)
                    netp.filter_global(data['img_512'].to(dev))
                    netp.filter_local(data['img'].to(dev)[:, None])
                    synth.calibrate_occupancy_bias(netp, calib, sdp, 0.3)

                    def step_p():
                        _lib.call('pifu_drop_tables', hp)
                        return pdist.eval_lattice_sharded(netp, calib, res, args.chunk, flags)
                    step_p()
                    fence()
                    reset_prof(_lib, hp)
                    _lib.call('pifu_profile_enable', hp, 1)
                    p0 = time.perf_counter()
                    for _ in range(args.steps):
                        field_p = step_p()
                    fence()
                    p_el = time.perf_counter() - p0
                    _lib.call('pifu_profile_enable', hp, 0)
                    pg_ms, pg_n, pg_fl = read_prof(_lib, hp, _lib.PROF_GEMM)
                    pw_ms, pw_n, pw_by = read_prof(_lib, hp, _lib.PROF_WALK)
                    p_ach = pg_fl / (pg_ms * 1e-3) / 1e12 if pg_ms > 0 else 0.0
                    occ = {'value': round(total * args.steps / p_el / 1e6, 3), 'unit': 'Mqueries/s', 'steps': args.steps,
                           'ms_per_step': round(p_el / max(args.steps, 1) * 1e3, 3),
                           'workload': f'dense {res[0]}x{res[1]}x{res[2]} lattice, reference chunk {args.chunk} regrouped by the library',
                           'library_group_rows': pdist.effective_chunk(netp._ctx, args.chunk, res[2], flags), 'batch_rows': rows_p,
                           'gemm_flop_per_query_executed': round(pg_fl / max(total * args.steps, 1)),
                           'roofline': {'kernel': 'k_mlp_gemm_tri', 'bound': 'mfma', 'achieved': round(p_ach, 2), 'peak': PEAK_FP32_MATRIX_TFLOPS,
                                        'unit': 'TFLOP/s', 'frac': round(p_ach / PEAK_FP32_MATRIX_TFLOPS, 4), 'launches': pg_n,
                                        'avg_launch_ms': round(pg_ms / max(pg_n, 1), 4),
                                        'gemm_time_share_of_step': round(pg_ms * 1e-3 / p_el, 4)},
                           'k_walk_ms_per_step': round(pw_ms / max(args.steps, 1), 3), 'k_walk_launches': pw_n}
                    if not args.no_cpu_baseline:      # parity of this configuration's field against the CPU oracle (ten chunks)
                        from oracle import pifu_oracle as O
                        refp = O.OracleNets(sdp, ocp, ofp, fine_training=True)
                        refp.feat_coarse, refp.feat_fine = netGp.im_feat_list[-1].cpu(), netp.im_feat_list[-1].cpu()
                        cinv_p = np.linalg.inv(calib[0].numpy().astype(np.float64))
                        first_p = (total // 2 // args.chunk) * args.chunk
                        wants = []
                        with torch.no_grad():
                            for c in range(10):
                                flat = np.arange(first_p + c * args.chunk, first_p + (c + 1) * args.chunk, dtype=np.int64)
                                pts = torch.from_numpy(O.grid_points_rect(res, cinv_p, flat).astype(np.float32))[None]
                                wants.append(refp.query(pts, calib)[0, 0])
                        wantp = torch.cat(wants).numpy()
                        occ['parity_check_max_abs_diff'] = float(np.abs(field_p[first_p:first_p + len(wantp)].cpu().numpy() - wantp).max())
                        occ['parity_sample'] = f'{len(wantp)} queries from the middle of the lattice against oracle/pifu_oracle.py in {args.chunk}-point calls'
                    log(f"pifuhd.pt configuration: {occ['value']} Mq/s, GEMM {p_ach:.1f} TFLOP/s")
                    pt = {'configuration': "norm='group' (GroupNorm HGFilters), mlp_norm='no' (MLP.py:37-41: no norm layers), name-seeded "
                                           "weights with MLP std 0.10 (tests/golden/e2e_full_pifuhd.npz pins this configuration end to end)",
                          'occupancy': occ}
                    # M2 on the name-seeded weights themselves: a rough sheet through most of the volume - the GEMM-bound end
                    tr = {}
                    recon.gen_mesh(args.genmesh_res, netp, dev, data, os.path.join(td, 'pt_rough_warm.obj'))
                    drop('pt_rough_warm.obj')
                    recon.gen_mesh(args.genmesh_res, netp, dev, data, os.path.join(td, 'pt_rough.obj'), timings=tr)
                    drop('pt_rough.obj')
                    pt['gen_mesh_name_seeded_weights'] = {'resolution': args.genmesh_res,
                                                          **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in tr.items()},
                                                          **octree_share(mesh_util, netp, calib, args.genmesh_res)}
                    # M2 on the body-like subject (the same net object: body image through its encoders, MLPs crafted)
                    _, _, data_pb, fit_pb = body_subject(args, dev, 'no', 'group', 0.10, net=netp, sd=sdp)
                    gp, _, _, _ = gen_mesh_on_roofline(recon, _lib, netp, dev, data_pb, args.genmesh_res, args.genmesh_runs, td, 'pt_body', log)
                    gp.update(octree_share(mesh_util, netp, data_pb['calib'], args.genmesh_res))
                    mp_ = recon.gen_mesh(args.genmesh_res, netp, dev, data_pb, os.path.join(td, 'pt_body_check.obj'))
                    drop('pt_body_check.obj')
                    gp['mesh'] = mesh_components(netp, mp_[1])
                    gp['fit'] = {k: round(v, 4) for k, v in fit_pb.items()}
                    pt['gen_mesh'] = gp
                    out['pifuhd_pt_config'] = pt
                    del netp, netGp, mp_
                except Exception as e:
                    out['pifuhd_pt_config'] = {'error': repr(e)}
                    log(f'pifuhd.pt configuration failed: {e!r}')
                finally:
                    _lib.call('pifu_profile_enable', h, 0)
            # ---- CPU baseline for the second half of the metric: the oracle's gen_mesh phases (octree grid evaluation,
            # serial Lewiner marching cubes, the calc_normal colour loop) at a resolution the host finishes in seconds,
            # with the GPU run at the SAME resolution beside it.  The encoders are left out on the CPU side (minutes there):
            # the oracle uses the feature maps of the GPU filter pass, which the sample text says.
            if not args.no_cpu_genmesh and not args.no_cpu_baseline:
                from oracle import pifu_oracle as O
                ncores = min(os.cpu_count() or 1, args.cpu_threads)
                torch.set_num_threads(ncores)
                ref = O.OracleNets(sd, oc, of, fine_training=True)
                ref.feat_coarse = netG.im_feat_list[-1].cpu()
                ref.feat_fine = netMR.im_feat_list[-1].cpu()
                calib_np = calib[0].numpy().astype(np.float64)

                def oracle_field(pts):
                    with torch.no_grad():
                        return ref.query(torch.from_numpy(pts.astype(np.float32))[None], calib)[0, 0].numpy()

                def cpu_gen_mesh(Rc):
                    stats = []
                    c0 = time.perf_counter()
                    sdf_c = O.eval_grid_octree(Rc, np.linalg.inv(calib_np), oracle_field, 5000, stats=stats)
                    c1 = time.perf_counter()
                    cv, cf = O.marching_cubes_lewiner(sdf_c, 0.5)
                    c2 = time.perf_counter()
                    vt = torch.from_numpy(((cv.astype(np.float64) * (2.0 / Rc) - 1.0) * np.array([1.0, -1.0, 1.0])).T.astype(np.float32))[None]
                    spans = recon._normal_spans(len(cv))
                    done_v, n_spans = 0, 0
                    with torch.no_grad():
                        for (l, r_) in spans:
                            if r_ <= l:
                                continue
                            ref.calc_normal(vt[:, :, l:r_], calib)
                            done_v += r_ - l
                            n_spans += 1
                            if time.perf_counter() - c2 > args.cpu_seconds:
                                break
                    c3 = time.perf_counter()
                    normals_s = (c3 - c2) * (len(cv) / max(done_v, 1))          # extrapolated when the loop was cut short
                    gt = {}
                    recon.gen_mesh(Rc, netMR, dev, data, os.path.join(td, 'same_res.obj'), timings=gt)
                    drop('same_res.obj')
                    # the composition on the device against the composition on the host: the float64 SDF of the whole
                    # multi-level octree (network field), the points each evaluated, and the mesh
                    sdf_g, n_gpu = mesh_util.eval_grid_device(netMR, calib, Rc, 5000, True)
                    gv, gf = mesh_util.marching_cubes_device(netMR._ctx, sdf_g, 0.5)
                    res_c = {
                        'resolution': Rc, 'kind': 'port', 'cores': ncores, 'unit': 's',
                        'value': round((c1 - c0) + (c2 - c1) + normals_s, 3),
                        'reconstruction_s': round(c2 - c0, 3), 'grid_eval_s': round(c1 - c0, 3), 'marching_cubes_s': round(c2 - c1, 3),
                        'normals_s': round(normals_s, 3), 'normals_measured_on': f'{n_spans} of {len(spans)} calc_normal calls '
                        f'({done_v} of {len(cv)} vertices), scaled to all',
                        'points_evaluated': int(sum(n for _, n in stats)), 'n_verts': int(len(cv)), 'n_faces': int(len(cf)),
                        'sample': f'oracle (oracle/pifu_oracle.py + oracle/mc_lewiner.c) octree grid + marching cubes + colour loop at '
                                  f'{Rc}^3 on {ncores} of {os.cpu_count()} host threads; encoders excluded (feature maps taken from the GPU pass); '
                                  f'no OBJ write',
                        'gpu_same_resolution': {'total_s': round(gt.get('total_s', float('nan')), 4),
                                                'filter_s': round(gt.get('filter_s', float('nan')), 4),
                                                'reconstruction_s': round(gt.get('reconstruction_s', float('nan')), 4),
                                                'normals_s': round(gt.get('normals_s', float('nan')), 4),
                                                'obj_s': round(gt.get('obj_s', float('nan')), 4),
                                                'n_verts': gt.get('n_verts'), 'n_faces': gt.get('n_faces'),
                                                'points_evaluated': int(n_gpu),
                                                'max_abs_diff_sdf': float(np.abs(sdf_g.cpu().numpy() - sdf_c).max()),
                                                'same_points_evaluated': bool(int(n_gpu) == int(sum(n for _, n in stats))),
                                                # the two SDFs agree to ~3e-6, not bit for bit: where a sample lies that close to
                                                # the level the two meshes may differ by a few cells (marching cubes on the SAME
                                                # volume is bit-exact: gen_mesh.mc_check and the tests)
                                                'mesh_size_equal': bool(len(gv) == len(cv) and len(gf) == len(cf)),
                                                'mesh_size_device_sdf': [int(len(gv)), int(len(gf))]}}
                    log(f"cpu gen_mesh baseline at {Rc}^3: {res_c['value']} s; GPU {gt.get('total_s'):.3f} s; "
                        f"max |sdf_gpu - sdf_oracle| = {res_c['gpu_same_resolution']['max_abs_diff_sdf']:.2e}")
                    return res_c, c3 - c0
                try:
                    out['cpu_baseline_gen_mesh'], spent = cpu_gen_mesh(args.cpu_genmesh_res)
                    # BASELINE.md section 3 plans 64^3 / 256^3 / 512^3: the next size up when the host got through this one
                    # quickly enough (8x the lattice, ~4x the octree's points)
                    if spent < 40.0 and args.cpu_genmesh_res2 > args.cpu_genmesh_res:
                        out['cpu_baseline_gen_mesh_%d' % args.cpu_genmesh_res2], _ = cpu_gen_mesh(args.cpu_genmesh_res2)
                except Exception as e:
                    out.setdefault('cpu_baseline_gen_mesh', {'error': repr(e)})
                    out['cpu_baseline_gen_mesh_error'] = repr(e)
    if secondary:
        out['roofline_secondary'] = secondary
    if cold:
        out['cold_start'] = {**cold, 'note': 'first_encoder_pass_s includes packing + uploading 1.5 GB of encoder weights; '
                             'first_gen_mesh_call_s is the first call at the timed resolution (arenas, pinned result slabs, pixel table)'}
    if multi is not None:
        out['gen_mesh'] = multi
    if rccl_summary:
        out['rccl'] = rccl_summary
    if args.dist_selftest:
        out['dist_selftest'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                                'collectives': 'all_gather_into_tensor (field, colours), all_reduce(MAX), barrier',
                                'note': 'functional check of the multi-rank code path on one rank; not a measurement'}
    print(json.dumps(out), file=_real_stdout if '_real_stdout' in globals() else sys.stdout, flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    # stdout carries the JSON line and nothing else: whatever the package prints on the way (the reference's own messages,
    # e.g. 'error cannot marching cubes') goes to stderr
    _real_stdout = sys.stdout
    sys.stdout = sys.stderr
    try:
        main()
    finally:
        sys.stdout = _real_stdout
