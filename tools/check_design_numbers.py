#!/usr/bin/env python3
"""Every number of DESIGN.md section 6 ("Measured") regenerated from the tracked profiles of the round - and a check that the
document holds exactly that.

    python tools/check_design_numbers.py            # exit 1 when DESIGN.md disagrees with profiles/r06_* (CPU test suite)
    python tools/check_design_numbers.py --write    # rewrite the generated block of DESIGN.md from the profiles

Sources (all tracked under profiles/, all from ONE round; the box and its clock are stated in the block):
  r06_bench_line.json              the line `python bench.py` printed (driver contract + roofline blocks)
  r06_kernel_stats.csv             rocprofv3 --kernel-trace --stats of the headline leg (tools/profile_gpu.sh)
  r06_traffic.json                 the separate --pmc passes (HBM bytes per launch, matrix-pipe busy)
  r06_genmesh_kernel_stats.txt     tools/prof_genmesh.sh, headline subject
  r06_genmesh_body_kernel_stats.txt  the same for the body-like subject
  r06_encoder_kernel_stats.txt     one steady-state encoder pass by kernel (tools/run_profiles_r06.sh)
Besides formatting, the script CROSS-CHECKS the sources against each other (the driver line's HIP-event average of the GEMM
against rocprofv3's, launch counts against the batch size they were taken at, fractions against achieved / peak) and fails
when two tracked files contradict each other - the round-5 document quoted 2.70 / 2.80 ms from another box beside profiles
that held 2.83 / 2.89 ms, and 244 launches (1 M-row batches) beside the bench line's 144 (2 M-row batches)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')
TAG = 'r06'
BEGIN, END = '<!-- BEGIN GENERATED: measured (tools/check_design_numbers.py --write) -->', '<!-- END GENERATED: measured -->'
PEAK = 157.3


class Mismatch(Exception):
    pass


def need(cond, msg):
    if not cond:
        raise Mismatch(msg)


def close(a, b, rel, what):
    need(abs(a - b) <= rel * max(abs(a), abs(b)), f'{what}: {a} vs {b} (more than {rel * 100:.0f} % apart)')


def load():
    src = {}
    src['line'] = json.load(open(os.path.join(PROF, f'{TAG}_bench_line.json')))
    rows, header = [], []
    for ln in open(os.path.join(PROF, f'{TAG}_kernel_stats.csv')):
        (header if ln.startswith('#') else rows).append(ln)
    src['stats_header'] = header
    src['stats'] = list(csv.DictReader(rows))
    src['traffic'] = json.load(open(os.path.join(PROF, f'{TAG}_traffic.json')))
    for key, name in (('genmesh', f'{TAG}_genmesh_kernel_stats.txt'), ('genmesh_body', f'{TAG}_genmesh_body_kernel_stats.txt'),
                      ('encoder', f'{TAG}_encoder_kernel_stats.txt')):
        src[key] = open(os.path.join(PROF, name)).read()
    return src


CALL = re.compile(r'call (\d+): k_mlp_gemm_tri (\d+) launches \(batch_rows (\d+)\), ([\d.]+) TFLOP, ([\d.]+) ms summed \(HIP events\); '
                  r'encoder convolutions (\d+) launches, ([\d.]+) TFLOP; .* / ([\d.]+) s / 157.3 TFLOP/s = ([\d.]+) of')
KERNEL = re.compile(r'^(\S.*?)\s+calls\s+(\d+)\s+avg_us\s+([\d.]+)\s+total_ms\s+([\d.]+)', re.M)


def genmesh_calls(text):
    out = []
    for m in CALL.finditer(text):
        out.append({'launches': int(m.group(2)), 'batch_rows': int(m.group(3)), 'gemm_tflop': float(m.group(4)), 'gemm_ms': float(m.group(5)),
                    'enc_launches': int(m.group(6)), 'enc_tflop': float(m.group(7)), 'total_s': float(m.group(8)), 'frac': float(m.group(9))})
    need(out, 'no "call N: k_mlp_gemm_tri ..." line in a gen_mesh profile')
    return out


def kernels(text):
    return {m.group(1).strip(): {'calls': int(m.group(2)), 'avg_us': float(m.group(3)), 'total_ms': float(m.group(4))} for m in KERNEL.finditer(text)}


def generate(src):
    d = src['line']
    ro, cfg = d['roofline'], d['config']
    L = []
    add = L.append
    # ---- cross-checks inside the driver line
    close(d['value'], cfg['queries_per_step'] / (d['ms_per_step'] * 1e-3) / 1e6, 0.002, 'value vs queries_per_step / ms_per_step')
    close(ro['achieved'], ro['alg_gflop_per_launch'] / ro['avg_launch_ms'], 0.002, 'roofline.achieved vs alg_gflop_per_launch / avg_launch_ms')
    close(ro['frac'], ro['achieved'] / PEAK, 0.002, 'roofline.frac vs achieved / 157.3')
    close(ro['alg_gflop_per_launch'] * ro['launches'], cfg['gemm_flop_per_query_executed'] * cfg['queries_per_step'] * d['steps'] / 1e9, 0.001,
          'GEMM launches x GFLOP per launch vs executed FLOP per query x queries')
    # ---- the profiler's view of the same kernel
    gemm = [r for r in src['stats'] if r['Name'].startswith('k_mlp_gemm_tri')]
    need(len(gemm) >= 1, 'r06_kernel_stats.csv: no k_mlp_gemm_tri row')
    # one row per template instantiation (with / without the statistics epilogue): the bench line's average is over all of them
    calls, total_ns = sum(int(g['Calls']) for g in gemm), sum(float(g['TotalDurationNs']) for g in gemm)
    hdr = ' '.join(src['stats_header'])
    m = re.search(r'(\d+) launches = .*?\(= (\d+), the headline mix', hdr)
    need(m and int(m.group(1)) == calls, 'r06_kernel_stats.csv: the header must state the launch mix of the trace')
    headline = int(m.group(2))
    passes = headline // (ro['launches'] // d['steps'])
    need(headline == passes * (ro['launches'] // d['steps']), 'trace launch mix is not a whole number of lattice passes')
    prof_avg_ms = total_ns / calls / 1e6                       # incl. the few tiny set-up launches: a lower bound of the headline average
    prof_tflops = headline * ro['alg_gflop_per_launch'] / (total_ns / 1e6)      # set-up launches' time left in: conservative
    close(prof_avg_ms * calls / headline, ro['avg_launch_ms'], 0.06, 'k_mlp_gemm_tri average: rocprofv3 vs the HIP events of the bench line')
    tr = src['traffic']
    alg_bytes = None
    sq = tr['detail'].get('k_mlp_gemm_tri_sq', {})
    # ---- gen_mesh, headline subject
    gm = d['gen_mesh']
    gro = gm['roofline']
    close(gro['frac'], gro['executed_tflop_per_call'] / gm['total_s'] / PEAK, 0.003, 'gen_mesh.roofline.frac')
    calls_h = genmesh_calls(src['genmesh'])
    ch = calls_h[-1]
    if ch['batch_rows'] == cfg['batch_rows']:
        need(ch['launches'] == gro['mlp_gemm_launches_per_call'], f"gen_mesh GEMM launches per call: profile {ch['launches']} vs bench line "
             f"{gro['mlp_gemm_launches_per_call']} at the same batch_rows {cfg['batch_rows']}")
    close(ch['gemm_tflop'], gro['mlp_gemm_tflop_per_call'], 0.002, 'gen_mesh executed GEMM TFLOP per call: profile vs bench line')
    kh = kernels(src['genmesh'])
    # ---- gen_mesh, body-like subject
    body = gm['body_subject']
    bro = body['roofline']
    close(bro['frac'], bro['executed_tflop_per_call'] / body['total_s'] / PEAK, 0.003, 'gen_mesh.body_subject.roofline.frac')
    calls_b = genmesh_calls(src['genmesh_body'])
    cb = calls_b[-1]
    close(cb['gemm_tflop'], bro['mlp_gemm_tflop_per_call'], 0.002, 'body subject: executed GEMM TFLOP per call, profile vs bench line')
    kb = kernels(src['genmesh_body'])
    pt = d['pifuhd_pt_config']
    need('error' not in pt, 'pifuhd_pt_config failed in the tracked bench line')
    po, pg = pt['occupancy'], pt['gen_mesh']
    close(po['roofline']['frac'], po['roofline']['achieved'] / PEAK, 0.002, 'pifuhd_pt_config.occupancy.roofline.frac')
    close(pg['roofline']['frac'], pg['roofline']['executed_tflop_per_call'] / pg['total_s'] / PEAK, 0.003, 'pifuhd_pt_config.gen_mesh.roofline.frac')
    # ---- encoders
    enc = next(s for s in d['roofline_secondary'] if s['kernel'].startswith('encoders'))
    close(enc['frac'], enc['alg_tflop_per_pass'] / (enc['pass_ms'] * 1e-3) / PEAK, 0.003, 'encoders frac')
    ke = kernels(src['encoder'])
    m = re.search(r'one pass: (\d+) launches, kernel time ([\d.]+) ms, first start -> last end ([\d.]+) ms', src['encoder'])
    need(m, 'r06_encoder_kernel_stats.txt: "one pass: N launches, kernel time X ms" line missing')
    enc_launches, enc_kernel_ms = int(m.group(1)), float(m.group(2))
    conv_ms = sum(v['total_ms'] for k, v in ke.items() if k.startswith('k_conv_dma<') or k.startswith('k_conv_dma4<'))
    walk = next((s for s in d['roofline_secondary'] if s['kernel'].startswith('k_walk')), None)
    mc = next((s for s in d['roofline_secondary'] if s['kernel'].startswith('marching cubes')), None)
    nodes = next((s for s in d['roofline_secondary'] if s['kernel'].startswith('k_mc_nodebricks')), None)
    cpu, cg, cg2 = d['cpu_baseline'], d.get('cpu_baseline_gen_mesh'), d.get('cpu_baseline_gen_mesh_256')
    cn, can = d['calc_normal'], d['canonical_path']

    add(BEGIN)
    add(f"`python bench.py` (defaults) = BASELINE config 2 as named: RGB-D input (6 image channels), 1024 px + 512 px, dense 256^3 lattice, chunk "
        f"{cfg['chunk']}, fp32, **batch_rows {cfg['batch_rows']}** (point rows per batch of launches: every launch count below is at this "
        f"batch size unless it says otherwise).  Every figure of this block is computed by `tools/check_design_numbers.py` from the tracked "
        f"`profiles/{TAG}_*` files (one box; measured shader clock {ro['shader_clock_mhz_measured']:.0f} MHz under the headline load) and the CPU "
        f"test suite fails when this block and those files disagree.")
    add('')
    add('| quantity | value | source |')
    add('|---|---|---|')
    add(f"| occupancy throughput, dense 256^3, chunk {cfg['chunk']} (default path: column sharing + depth walk) | **{d['value']:.1f} Mqueries/s** "
        f"({d['ms_per_step']:.1f} ms per {cfg['queries_per_step'] / 1e6:.1f} M queries; {cfg['gemm_flop_per_query_executed']:,} GEMM FLOP per query executed "
        f"of the canonical {cfg['flop_per_query_canonical']:,}) | `{TAG}_bench_line.json` |")
    add(f"| `k_mlp_gemm_tri`, HIP events inside bench.py | {ro['launches']} launches x {ro['avg_launch_ms']:.3f} ms, {ro['alg_gflop_per_launch']:.1f} "
        f"GFLOP each = {ro['achieved']:.1f} TFLOP/s = **{ro['frac']:.3f} of the {PEAK} TFLOP/s fp32-MFMA peak**, {ro['frac_at_measured_clock']:.3f} at the "
        f"measured clock; {100 * ro['gemm_time_share_of_step']:.0f} % of the step | `{TAG}_bench_line.json` |")
    add(f"| the same kernel under rocprofv3 (`--kernel-trace --stats`, {passes} lattice passes) | {calls} launches ({headline} of the headline mix + "
        f"{calls - headline} small set-up launches), {total_ns / 1e6:.1f} ms in total: {total_ns / 1e6 / headline:.3f} ms per headline launch = "
        f"{prof_tflops:.1f} TFLOP/s = {prof_tflops / PEAK:.3f} (profiler overhead and another box's clock included) | `{TAG}_kernel_stats.csv` |")
    if sq:
        add(f"| PMC passes of the same command | HBM traffic {tr['k_mlp_gemm_hbm_bytes_per_launch'] / 1e9:.2f} GB per launch, matrix pipe busy "
            f"{sq.get('mfma_pipe_busy_fraction', float('nan')):.2f}, `SQ_LDS_BANK_CONFLICT` {sq.get('SQ_LDS_BANK_CONFLICT', float('nan')):.0f} | `{TAG}_traffic.json` |")
    add(f"| canonical per-point path (every one of the {can['flop_per_query_executed']:,} FLOP of a query executed) | {can['value']:.1f} Mq/s = "
        f"{can['whole_path_tflops']:.1f} TFLOP/s whole path ({can['whole_path_frac_of_fp32_matrix_peak']:.3f}); its GEMM launches {can['gemm_roofline']['frac']:.3f}; "
        f"field within {can['max_abs_diff_vs_headline_field']:.1e} of the headline run's | `{TAG}_bench_line.json` |")
    add(f"| `calc_normal` (the colour loop's kernel path, {cn['vertices']:,} vertices in {cn['call_size']:,}-vertex calls) | **{cn['pixel_tables']['value']:.1f} Mq/s** "
        f"through the pixel tables ({cn['pixel_tables']['ms']:.1f} ms, {cn['pixel_tables']['gemm_flop_per_query_executed'] / 1e6:.2f} MFLOP per query, GEMM "
        f"{cn['pixel_tables']['gemm_tflops']:.0f} TFLOP/s) against {cn['per_point_gemms']['value']:.1f} through the per-point GEMMs; "
        f"{cn['max_abs_diff_between_paths']:.1e} apart | `{TAG}_bench_line.json` |")
    if walk:
        add(f"| `k_walk` (depth walk of coarse layer 1) | {walk['avg_launch_ms']:.2f} ms per launch, {walk['achieved'] / 1e3:.2f} TB/s of bytes it must move = "
            f"{walk['frac']:.2f} of HBM peak (it sits on the L2: ~17 GB of weight-row reads per launch) | `{TAG}_bench_line.json` |")
    add(f"| CPU oracle on the same lattice | {cpu['value']:.4f} Mq/s on {cpu['cores']} host threads; on all {cpu.get('cores_alt')} threads "
        f"{cpu.get('value_alt')} Mq/s (`faster_choice`: {cpu.get('faster_choice')}); parity of the headline run vs the oracle "
        f"{d['parity_check_max_abs_diff']:.1e} | `{TAG}_bench_line.json` |")
    add(f"| **`gen_mesh` 512^3**, octree, the rough headline subject ({gm['n_verts'] / 1e6:.2f} M vertices, {100 * gm['share_of_grid']:.1f} % of the grid "
        f"evaluated) | **{gm['total_s']:.3f} s** (runs {', '.join('%.3f' % t for t in gm['runs_total_s'])}) = encoders {gm['filter_s']:.3f} + reconstruction "
        f"{gm['reconstruction_s']:.3f} + normals {gm['normals_s']:.3f} + OBJ tail {gm['obj_s']:.4f} | `{TAG}_bench_line.json` |")
    add(f"| ... on the roofline | {gro['executed_tflop_per_call']:.2f} TFLOP executed ({gro['mlp_gemm_tflop_per_call']:.2f} in {gro['mlp_gemm_launches_per_call']} "
        f"`k_mlp_gemm_tri` launches at batch_rows {cfg['batch_rows']} + {gro['encoder_conv_tflop_per_call']:.2f} in the encoders) / {gm['total_s']:.3f} s = "
        f"{gro['achieved']:.1f} TFLOP/s = **{gro['frac']:.3f}**; GEMM busy {100 * gro['mlp_gemm_busy_share_of_call']:.0f} % of the call at "
        f"{gro['mlp_gemm_frac_while_busy']:.2f}; reconstruction {gro['reconstruction']['frac']:.3f} ({gro['reconstruction']['frac_while_busy']:.2f} while busy), "
        f"normals {gro['normals']['frac']:.3f}, encoders {gro['encoders']['frac']:.3f} | `{TAG}_bench_line.json` |")
    add(f"| the same call under the profiling tool (batch_rows {ch['batch_rows']}) | {ch['launches']} GEMM launches, {ch['gemm_tflop']:.2f} TFLOP, "
        f"{ch['gemm_ms']:.0f} ms summed; {ch['total_s']:.3f} s = {ch['frac']:.3f}; `k_walk` {kh.get('k_walk<16, 2, 256>', {}).get('avg_us', 0) / 1e3:.2f} ms "
        f"x {kh.get('k_walk<16, 2, 256>', {}).get('calls', 0)} launches in the traced process, `k_oct_fill` "
        f"{kh.get('k_oct_fill', {}).get('avg_us', 0) / 1e3:.2f} ms each | `{TAG}_genmesh_kernel_stats.txt` |")
    add(f"| **`gen_mesh` 512^3, body-like subject** (ONE watertight component: {body['mesh']['components']} component, {body['n_verts'] / 1e6:.2f} M vertices, "
        f"{100 * body['share_of_grid']:.2f} % of the grid evaluated) | **{body['total_s']:.4f} s** = encoders {body['filter_s']:.4f} + reconstruction "
        f"{body['reconstruction_s']:.4f} + normals {body['normals_s']:.4f}; {bro['executed_tflop_per_call']:.2f} TFLOP executed = {bro['achieved']:.1f} TFLOP/s = "
        f"**{bro['frac']:.3f}**; GEMM busy {100 * bro['mlp_gemm_busy_share_of_call']:.0f} % of the call; reconstruction {bro['reconstruction']['frac']:.2f}, normals "
        f"{bro['normals']['frac']:.2f}, encoders {bro['encoders']['frac']:.2f} | `{TAG}_bench_line.json`, `{TAG}_genmesh_body_kernel_stats.txt` |")
    add(f"| **pifuhd.pt configuration** (GroupNorm filters, MLP without norm layers): occupancy, 256^3 lattice | **{po['value']:.1f} Mq/s** "
        f"({po['ms_per_step']:.1f} ms per step, groups of {po['library_group_rows']} rows instead of {cfg['chunk']}-point chunks), GEMM {po['roofline']['achieved']:.1f} "
        f"TFLOP/s = **{po['roofline']['frac']:.3f}**; parity vs the oracle {po.get('parity_check_max_abs_diff', float('nan')):.1e} | `{TAG}_bench_line.json` |")
    add(f"| ... `gen_mesh` 512^3, body-like subject | **{pg['total_s']:.4f} s** ({pg['n_verts'] / 1e6:.2f} M vertices, {100 * pg['share_of_grid']:.2f} % of the grid), "
        f"{pg['roofline']['executed_tflop_per_call']:.2f} TFLOP = **{pg['roofline']['frac']:.3f}**; reconstruction {pg['reconstruction_s']:.4f} s "
        f"({pg['roofline']['reconstruction']['frac']:.2f}); on its name-seeded weights (a rough sheet through {100 * pt['gen_mesh_name_seeded_weights']['share_of_grid']:.0f} % "
        f"of the grid, {pt['gen_mesh_name_seeded_weights']['n_verts'] / 1e6:.1f} M vertices): {pt['gen_mesh_name_seeded_weights']['total_s']:.2f} s | `{TAG}_bench_line.json` |")
    sm = gm.get('smoother_subject', {})
    c5 = gm.get('config5_1024', {})
    if sm and c5:
        add(f"| other subjects | smoother (MLP std 0.02, a field so flat that the octree stops after two levels): {sm['total_s']:.3f} s, {sm['n_verts'] / 1e6:.2f} M "
            f"vertices; **1024^3** on one GPU (BASELINE configs[4]'s grid): {c5['total_s']:.3f} s, {c5['n_verts'] / 1e6:.2f} M vertices | `{TAG}_bench_line.json` |")
    if cg and cg2:
        add(f"| CPU baseline for `gen_mesh` (oracle octree + serial Lewiner + colour loop, encoders excluded) | 128^3: {cg['value']:.1f} s vs GPU "
            f"{cg['gpu_same_resolution']['total_s']:.3f} s; 256^3: {cg2['value']:.0f} s vs {cg2['gpu_same_resolution']['total_s']:.3f} s; max |SDF_gpu - SDF_oracle| "
            f"{cg['gpu_same_resolution']['max_abs_diff_sdf']:.1e} / {cg2['gpu_same_resolution']['max_abs_diff_sdf']:.1e}, same points evaluated per level | `{TAG}_bench_line.json` |")
    if mc and nodes:
        add(f"| marching cubes on the headline subject's 512^3 field | {mc['ms']:.1f} ms per call ({mc['frac']:.3f} of HBM over the whole call: the classification is "
            f"latency-bound); node pass {nodes['avg_launch_ms']:.2f} ms = {nodes['frac']:.2f} of HBM peak | `{TAG}_bench_line.json` |")
    add(f"| encoders per image (host image -> both feature maps) | **{enc['pass_ms']:.2f} ms** (passes {', '.join('%.2f' % p for p in enc['passes_ms'])}) = "
        f"{enc['alg_tflop_per_pass']:.2f} TFLOP executed at **{enc['frac']:.3f}** of the fp32-matrix peak ({enc['direct_convolution_tflop_per_pass']:.2f} TFLOP of direct "
        f"convolutions: {enc['direct_convolution_equivalent_frac']:.2f}) | `{TAG}_bench_line.json` |")
    add(f"| one steady-state encoder pass by kernel | {enc_launches} launches, {enc_kernel_ms:.2f} ms of kernel time, convolutions {conv_ms:.2f} ms "
        f"({100 * conv_ms / enc_kernel_ms:.0f} %), everything else {enc_kernel_ms - conv_ms:.2f} ms ({100 * (1 - conv_ms / enc_kernel_ms):.0f} %): `k_prep_act` "
        f"{ke.get('k_prep_act', {}).get('total_ms', 0):.2f}, `k_stats_finalize` {ke.get('k_stats_finalize', {}).get('total_ms', 0):.2f} "
        f"({ke.get('k_stats_finalize', {}).get('avg_us', 0):.1f} us each) | `{TAG}_encoder_kernel_stats.txt` |")
    cold = d.get('cold_start', {})
    if cold:
        add(f"| cold start of a fresh process | first encoder pass {cold.get('first_encoder_pass_s')} s (steady {cold.get('steady_encoder_pass_s')}), first `gen_mesh` "
            f"512^3 {cold.get('first_gen_mesh_call_s')} s (steady {gm['total_s']:.3f}) | `{TAG}_bench_line.json` |")
    add(END)
    return '\n'.join(L)


def main():
    src = load()
    try:
        block = generate(src)
    except Mismatch as e:
        print('check_design_numbers: the tracked profiles contradict each other:', e)
        return 1
    path = os.path.join(ROOT, 'DESIGN.md')
    text = open(path).read()
    if '--write' in sys.argv:
        if BEGIN in text and END in text:
            text = text[:text.index(BEGIN)] + block + text[text.index(END) + len(END):]
        else:
            print('check_design_numbers: DESIGN.md has no generated block markers'); return 1
        open(path, 'w').write(text)
        print('DESIGN.md: generated block rewritten from profiles/%s_*' % TAG)
        return 0
    if BEGIN not in text or END not in text:
        print('check_design_numbers: DESIGN.md has no generated block'); return 1
    have = text[text.index(BEGIN):text.index(END) + len(END)]
    if have != block:
        import difflib
        print('check_design_numbers: DESIGN.md section 6 differs from what profiles/%s_* give:' % TAG)
        for ln in difflib.unified_diff(have.splitlines(), block.splitlines(), 'DESIGN.md', 'profiles', lineterm='', n=0):
            print(ln[:400])
        return 1
    print('check_design_numbers: DESIGN.md section 6 == profiles/%s_* (%d rows)' % (TAG, block.count('\n|') - 2))
    return 0


if __name__ == '__main__':
    sys.exit(main())
