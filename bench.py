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
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_QUERY = 2916098            # coarse MLP 2,100,738 + fine MLP 815,360
PEAK_FP32_MATRIX_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak


def genmesh_multi(args, netMR, dev, data, synth, dist, rank, world, fence, log):
    """gen_mesh wall time on N GPUs: (a) ONE subject, grid evaluation + colour loop sharded (max over ranks);
    (b) N subjects, one per GPU, no communication (time for all N to finish).  Every rank always takes part in every
    collective, also after a local failure (reported as an error string), so a bad subject cannot hang the job."""
    import tempfile
    import torch
    from pifuhd_amd import reconstruction as recon
    res = {'resolution': args.genmesh_res, 'use_octree': True, 'n_gpus': world}
    keys = ('filter_s', 'png_s', 'reconstruction_s', 'normals_s', 'obj_s')

    def reduce_max(vals):
        t = torch.tensor(vals, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    with tempfile.TemporaryDirectory() as td:
        # (a) sharded: identical inputs on every rank, so a failure is the same on every rank
        timings, err, t_tot = {}, None, float('inf')
        try:
            recon.gen_mesh(128, netMR, dev, data, os.path.join(td, f'warm{rank}.obj'), sharded=True)
            fence()
            t0 = time.perf_counter()
            recon.gen_mesh(args.genmesh_res, netMR, dev, data, os.path.join(td, f'mesh{rank}.obj'), timings=timings, sharded=True)
            fence()
            t_tot = time.perf_counter() - t0
        except Exception as e:
            err = repr(e)
        red = reduce_max([t_tot] + [timings.get(k, float('inf')) for k in keys])
        if err is None and red[0] != float('inf'):
            res['one_subject_sharded'] = {'total_s': round(red[0], 4), **{k: round(v, 4) for k, v in zip(keys, red[1:])},
                                          'n_verts': timings.get('n_verts'), 'n_faces': timings.get('n_faces')}
            log(f"gen_mesh sharded: {red[0]:.3f} s")
        else:
            res['one_subject_sharded'] = {'error': err or 'failed on another rank'}
        # (b) independent subjects (different images: a failure may be local to one rank)
        err, t_tot = None, float('inf')
        try:
            own = synth.synthetic_inputs(1024, 3, seed=rank)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            recon.gen_mesh(args.genmesh_res, netMR, dev, own, os.path.join(td, f'subj{rank}.obj'))
            torch.cuda.synchronize(dev)
            t_tot = time.perf_counter() - t0
        except Exception as e:
            err = repr(e)
        red = reduce_max([t_tot])
        if red[0] != float('inf'):
            res['independent_subjects'] = {'subjects': world, 'total_s': round(red[0], 4), 'subjects_per_s': round(world / red[0], 3)}
        else:
            res['independent_subjects'] = {'error': err or 'failed on another rank'}
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
    ap.add_argument('--no-genmesh', action='store_true')
    ap.add_argument('--no-lean-pass', action='store_true', help='skip the extra pass without the unread coarse tail')
    ap.add_argument('--no-split-precision', action='store_true', help='skip the extra bf16x3 / fp16x3 measurements')
    ap.add_argument('--genmesh-res', type=int, default=512)
    ap.add_argument('--features', choices=['filter', 'synthetic'], default='filter',
                    help="feature maps from the real encoders (filter) or seeded random maps of the same shape")
    ap.add_argument('--lean', action='store_true', help='skip the coarse MLP layers whose output nobody reads')
    ap.add_argument('--no-depth-walk', action='store_true',
                    help='evaluate coarse layer 1 as a per-point GEMM instead of walking the columns along z')
    ap.add_argument('--no-column-sharing', action='store_true',
                    help='evaluate every lattice point independently (the canonical 2 916 098 FLOP per query)')
    args = ap.parse_args()

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
    dev = torch.device('cuda', 0 if one_gpu_test else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_gpu_test:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    # ---- model + inputs (outside the timed region) ---------------------------------------------------
    oc = config.coarse_opt()
    of = config.fine_opt(oc)
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
    netG, netMR = nets.build_nets(oc, of, sd, dev)
    def log(msg):
        print(f'[bench r{rank} {time.perf_counter() - t_start:7.1f}s] {msg}', file=sys.stderr, flush=True)
    log('weights ready')
    data = synth.synthetic_inputs(1024, 3, seed=0)      # one subject: every rank holds the same image + weights
    if args.features == 'filter':
        netMR.filter_global(data['img_512'].to(dev))
        netMR.filter_local(data['img'].to(dev)[:, None])
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
    _lib.call('pifu_reserve', h, args.batch_rows)

    def step():
        return pdist.eval_lattice_sharded(netMR, calib, res, args.chunk, flags)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    log('warm-up done')
    _lib.call('pifu_profile_read', h, None, None, None, 1)
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
    mhz = [100.0 * (clk1[2 * x] - clk0[2 * x]) / (clk1[2 * x + 1] - clk0[2 * x + 1])
           for x in range(8) if clk0[2 * x + 1] and clk1[2 * x + 1] > clk0[2 * x + 1]]
    mhz = sorted(m for m in mhz if 100.0 < m < 4000.0)      # an XCD whose counter was reset by power gating reads garbage: dropped
    eff_mhz = mhz[len(mhz) // 2] if mhz else None            # median over the XCDs
    _lib.call('pifu_profile_enable', h, 0)
    gemm_ms, gemm_n, gemm_flops = C.c_double(0), C.c_int64(0), C.c_double(0)
    _lib.call('pifu_profile_read', h, C.byref(gemm_ms), C.byref(gemm_n), C.byref(gemm_flops), 1)

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    value = total * args.steps / elapsed / 1e6

    # ---- second half of the metric at N > 1: one subject's 512^3 gen_mesh sharded over the ranks, and N independent
    # subjects (one per GPU, BASELINE configs[3]); every rank takes part, rank 0 reports
    multi = None
    if world > 1 and not args.no_genmesh and args.features == 'filter':
        multi = genmesh_multi(args, netMR, dev, data, synth, dist, rank, world, fence, log)
    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    achieved = gemm_flops.value / (gemm_ms.value * 1e-3) / 1e12 if gemm_ms.value > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get('k_mlp_gemm_hbm_bytes_per_launch')
        except Exception:
            traffic = None
    out = {
        'metric': 'occupancy_queries_throughput', 'value': round(value, 3), 'unit': 'Mqueries/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': f'dense {res[0]}x{res[1]}x{res[2]} lattice ({R}^3 per GPU, axis-0 chunk-range slabs, RCCL '
                        f'all-gather of the fp32 field), coarse+fine PIFuMRNet query, chunk {args.chunk}, '
                        f'1024px RGB image + 512px, name-seeded weights',
            'queries_per_step': total, 'chunk': args.chunk, 'batch_rows': args.batch_rows, 'coarse_tail': 'skipped' if args.lean else 'computed',
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
            'launches': gemm_n.value, 'avg_launch_ms': round(gemm_ms.value / max(gemm_n.value, 1), 4),
            'alg_gflop_per_launch': round(gemm_flops.value / max(gemm_n.value, 1) / 1e9, 3),
            'gemm_time_share_of_step': round(gemm_ms.value * 1e-3 / elapsed, 4),
            # measured beside the nominal-clock figures above: the peak assumes 2 400 MHz, the fp32-MFMA load runs lower
            'shader_clock_mhz_measured': round(eff_mhz, 1) if eff_mhz else None,
            'frac_at_measured_clock': round(achieved / (PEAK_FP32_MATRIX_TFLOPS * eff_mhz / 2400.0), 4) if eff_mhz else None,
        },
    }

    # ---- the same lattice without the coarse MLP's tail (layers 3-4 + sigmoid produce netG.preds, which nothing on the
    # gen_mesh path reads: reconstruction() runs this way); reported beside the headline, never as `value`
    if world == 1 and not args.lean and not args.no_lean_pass:
        try:
            lean_flags = flags & ~_lib.Q_FULL_COARSE
            pdist.eval_lattice_sharded(netMR, calib, res, args.chunk, lean_flags)
            fence()
            l0 = time.perf_counter()
            for _ in range(args.steps):
                field_lean = pdist.eval_lattice_sharded(netMR, calib, res, args.chunk, lean_flags)
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
                _lib.call('pifu_profile_read', h, None, None, None, 1)
                _lib.call('pifu_profile_enable', h, 1)
                s0 = time.perf_counter()
                field_split = None
                for _ in range(args.steps):
                    field_split = step()
                fence()
                s_el = time.perf_counter() - s0
                _lib.call('pifu_profile_enable', h, 0)
                sm, sn, sf = C.c_double(0), C.c_int64(0), C.c_double(0)
                _lib.call('pifu_profile_read', h, C.byref(sm), C.byref(sn), C.byref(sf), 1)
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

    # ---- parity spot check + CPU baseline (oracle = checker; never the thing measured as `value`) -----
    if world == 1 and not args.no_cpu_baseline:
        from oracle import pifu_oracle as O
        # torch-CPU scales badly past a few dozen threads on these small per-chunk GEMMs (a 256-thread run
        # of this oracle is >10x slower than a 32-thread one), so the baseline uses at most 32 threads and
        # says so in `cores`; the sample is bounded by wall time (~args.cpu_seconds), not by a chunk count.
        ncores = min(os.cpu_count() or 1, args.cpu_threads)
        torch.set_num_threads(ncores)
        ref = O.OracleNets(sd, oc, of, fine_training=True)
        ref.feat_coarse = netG.im_feat_list[-1].cpu()
        ref.feat_fine = netMR.im_feat_list[-1].cpu()
        calib_inv = np.linalg.inv(calib[0].numpy().astype(np.float64))
        first = (total // 2 // args.chunk) * args.chunk            # a slab from the middle of the grid
        want = []
        npts = 0
        cpu_s = 0.0
        with torch.no_grad():
            while npts < args.cpu_chunks * args.chunk and first + npts < total:
                n_c = min(args.chunk, total - first - npts)
                flat = np.arange(first + npts, first + npts + n_c, dtype=np.int64)
                pts = torch.from_numpy(O.grid_points_rect(res, calib_inv, flat).astype(np.float32))[None]
                c0 = time.perf_counter()
                want.append(ref.query(pts, calib)[0, 0])
                cpu_s += time.perf_counter() - c0
                npts += n_c
                if cpu_s > args.cpu_seconds and npts >= 2 * args.chunk:
                    break
        want = torch.cat(want).numpy()
        log(f'cpu baseline done: {npts} queries in {cpu_s:.1f} s on {ncores} threads')
        got = field[first:first + npts].cpu().numpy()
        out['cpu_baseline'] = {
            'value': round(npts / cpu_s / 1e6, 4), 'unit': 'Mqueries/s', 'cores': ncores,
            'kind': 'port', 'sample': f'{npts} queries ({npts // args.chunk} chunks of {args.chunk}) from the middle of '
                                      f'the same lattice, torch-CPU oracle (oracle/pifu_oracle.py), {cpu_s:.1f} s, '
                                      f'{ncores} of {os.cpu_count()} host threads',
        }
        out['parity_check_max_abs_diff'] = float(np.abs(got - want).max())
        for key, fld in split_fields.items():      # the opt-in arithmetics against the same oracle sample
            out[key]['max_abs_diff_vs_cpu_oracle'] = float(np.abs(fld[first:first + npts].cpu().numpy() - want).max())

    # ---- second half of the metric: end-to-end gen_mesh wall time (single GPU) ---------------------------
    if world == 1 and not args.no_genmesh and args.features == 'filter':
        import tempfile
        from pifuhd_amd import reconstruction as recon
        with tempfile.TemporaryDirectory() as td:
            timings = {}
            try:
                recon.gen_mesh(128, netMR, dev, data, os.path.join(td, 'warm.obj'))            # warm-up (allocations)
                log('gen_mesh warm-up done')
                torch.cuda.synchronize(dev)
                recon.gen_mesh(args.genmesh_res, netMR, dev, data, os.path.join(td, 'mesh.obj'), timings=timings)
                out['gen_mesh'] = {'resolution': args.genmesh_res, 'use_octree': True,
                                   **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in timings.items()}}
            except Exception as e:      # the metric line must still be printed
                out['gen_mesh'] = {'error': repr(e)}
    if multi is not None:
        out['gen_mesh'] = multi
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
