"""Turn the rocprofv3 output of scripts/profile_round.sh into the summaries kept under profiles/:
    profiles/<tag>_kernel_stats_bench.csv   (rocprofv3 --kernel-trace --stats of the bench command)
    profiles/<tag>_pmc_per_wave.md          (every counter, per kernel, per launch and per wave)
    profiles/<tag>_pmc_kernelA.json         (what bench.py reads: VALU instructions per wave, HBM-side bytes per launch)
usage: python scripts/pmc_summary.py gpurun_out/prof_<tag> <tag>"""
import csv, glob, json, os, re, shutil, sqlite3, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import build as _build


def _dump(out, kind, path):
    """every summary carries the fingerprint of the kernel sources it was taken from (and the commit, where a work tree is at hand:
    the GPU box has none) -- bench.py compares it with the sources it runs"""
    out['pmc'] = dict(_build.pmc_fingerprint(kind), round=tag.split('_')[0])
    try:
        import subprocess
        out['pmc']['commit'] = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, check=True,
                                              cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    except Exception:
        out['pmc']['commit'] = None
    with open(path, 'w') as fh:
        json.dump(out, fh, indent=1)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, 'profiles')
os.makedirs(prof, exist_ok=True)

for f in glob.glob(os.path.join(src, 'stats', '**', '*kernel_stats.csv'), recursive=True):
    shutil.copy(f, os.path.join(prof, f'{tag}_kernel_stats_bench.csv'))
for f in glob.glob(os.path.join(src, 'stats', '**', '*_results.db'), recursive=True):   # rocprofv3's default output (rocpd)
    db = sqlite3.connect(f)
    per = defaultdict(list)
    for name, dur in db.execute('select name, duration from kernels'):
        per[name].append(dur)
    # every dispatch of the persistent STOMP kernel in launch order: the stats pass of profile_round.sh runs
    # `bench.py --steps K --warmup 5 --main-only`, i.e. pre-heat blocks (untimed), ONE warm-up launch of 5 steps, the R timed
    # K-step launches and the R launches with a HIP event pair on the dispatch
    try:
        seq = [(n, d) for n, d in db.execute('select name, duration from kernels order by start')]
    except sqlite3.Error:
        seq = [(n, d) for n, d in db.execute('select name, duration from kernels')]
    fused = [d for n, d in seq if 'stomp_fused' in n]
    if fused:
        with open(os.path.join(prof, f'{tag}_fused_dispatches.txt'), 'w') as fh:
            fh.write('# rocprofv3 --kernel-trace: duration (us) of every dispatch of the persistent STOMP kernel, in launch order, of\n'
                     '# `bench.py --steps K --warmup 5 --main-only` (scripts/profile_round.sh): pre-heat blocks, one 5-step warm-up launch\n'
                     '# (the short one), the R = 9 TIMED K-step launches, then the R = 9 launches with an event pair on the dispatch.\n')
            fh.write(' '.join('%.1f' % (d / 1e3) for d in fused) + '\n')
            short = [i for i, d in enumerate(fused) if d < 0.5 * sorted(fused)[len(fused) // 2]]
            if short:
                w = short[-1]
                timed = fused[w + 1:w + 10]
                prof_ = fused[w + 10:w + 19]
                if timed:
                    fh.write('timed launches (the 9 after the warm-up launch): mean %.1f us, median %.1f us\n'
                             % (sum(timed) / len(timed) / 1e3, sorted(timed)[len(timed) // 2] / 1e3))
                if prof_:
                    fh.write('event-profiled launches (the next 9): mean %.1f us, median %.1f us\n'
                             % (sum(prof_) / len(prof_) / 1e3, sorted(prof_)[len(prof_) // 2] / 1e3))
    tot = sum(sum(v) for v in per.values())
    with open(os.path.join(prof, f'{tag}_kernel_stats_bench.csv'), 'w') as fh:
        fh.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","MedianNs"\n')
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            v = sorted(v)
            fh.write(f'"{name}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / tot:.2f},{v[0]},{v[-1]},{v[len(v) // 2]}\n')
b = os.path.join(src, 'bench_under_rocprof.json')
if os.path.exists(b):
    shutil.copy(b, os.path.join(prof, f'{tag}_bench_under_rocprof.json'))

acc = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> values per dispatch
grid = {}
for f in glob.glob(os.path.join(src, 'pmc*', '**', '*_results.db'), recursive=True):
    db = sqlite3.connect(f)
    q = ('select kernel_name, counter_name, value, grid_size, workgroup_size, vgpr_count, sgpr_count, lds_block_size, '
         'scratch_size from counters_collection order by dispatch_id')
    for k, c, v, g, wg, vg, sg, lds, scr in db.execute(q):
        acc[k][c].append(float(v))
        grid[k] = (int(g), int(wg), int(vg or 0), int(sg or 0), int(lds or 0), int(scr or 0))
for f in glob.glob(os.path.join(src, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row['Kernel_Name']
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
            grid[k] = (int(row['Grid_Size']), int(row['Workgroup_Size']), int(row.get('VGPR_Count', 0) or 0),
                       int(row.get('SGPR_Count', 0) or 0), int(row.get('LDS_Block_Size', 0) or 0), int(row.get('Scratch_Size', 0) or 0))

lines = [f'# rocprofv3 PMC, {tag}: scripts/profile_round.sh (separate --pmc passes of scripts/prof_stomp.py, C3 shape);',
         '# per kernel: mean over its dispatches, per launch and per wave (grid / 64)', '']
summary = {}
for k in sorted(acc):
    g, wg, vg, sg, lds, scr = grid[k]
    waves = g // 64
    lines.append(f'## {k}')
    lines.append(f'grid {g} threads = {waves} waves, workgroup {wg}, VGPR {vg}, SGPR {sg}, LDS {lds} B, scratch {scr} B')
    lines.append(f'{"counter":34s} {"per launch":>16s} {"per wave":>12s} {"dispatches":>10s}')
    summary[k] = {}
    for c in sorted(acc[k]):
        v = acc[k][c]
        v = v[len(v) // 4:]                      # drop the first quarter (warm-up dispatches)
        m = sum(v) / len(v)
        summary[k][c] = m
        lines.append(f'{c:34s} {m:16.1f} {m / waves:12.1f} {len(v):10d}')
    lines.append('')
with open(os.path.join(prof, f'{tag}_pmc_per_wave.md'), 'w') as fh:
    fh.write('\n'.join(lines))

# the persistent STOMP kernel: one launch = MPB_ITERS iterations (scripts/profile_round.sh); per-iteration figures
# (MPB_PMC_NAME: which entry of the bench line the passes were taken for -- 'stomp' = the headline (C3), 'stomp_c5' = its
#  two-batch instantiation at 4096 particles (scripts/profile_c5.sh), 'stomp_h128' = the generalised kernel at H = 128
#  (scripts/profile_h128.sh); MPB_PMC_WORKLOAD describes the launches)
kf = [k for k in summary if 'stomp_fused_kernel' in k or 'stomp_fused_hx_kernel' in k]
if kf:
    import re
    k = kf[0]
    s = summary[k]
    waves = grid[k][0] // 64
    iters = int(os.environ.get('MPB_ITERS', 200))
    pmc_name = os.environ.get('MPB_PMC_NAME', 'stomp')
    fetch_kb, write_kb = s.get('FETCH_SIZE'), s.get('WRITE_SIZE')
    per = lambda c: (s[c] / waves / iters) if c in s else None
    out = {'kernel': re.sub(r'\(.*', '', k), 'workload': os.environ.get('MPB_PMC_WORKLOAD', 'C3 P=128 S=32 H=64 d=14') + ': bench.py\'s planner, launches of %d iterations from the initial means (scripts/prof_stomp.py, MPB_FUSED=1)' % iters,
           'waves_per_launch': waves, 'iterations_per_launch': iters, 'vgpr': grid[k][2], 'sgpr': grid[k][3], 'lds_bytes': grid[k][4],
           'scratch_bytes': grid[k][5],
           'SQ_INSTS_VALU_per_wave_iteration': per('SQ_INSTS_VALU'), 'SQ_INSTS_SALU_per_wave_iteration': per('SQ_INSTS_SALU'),
           'SQ_INSTS_LDS_per_wave_iteration': per('SQ_INSTS_LDS'), 'SQ_INSTS_MFMA_per_wave_iteration': per('SQ_INSTS_MFMA'),
           'SQ_WAVE_CYCLES_quads_per_wave_iteration': per('SQ_WAVE_CYCLES'), 'SQ_WAIT_ANY_quads_per_wave_iteration': per('SQ_WAIT_ANY'),
           'SQ_WAIT_INST_ANY_quads_per_wave_iteration': per('SQ_WAIT_INST_ANY'),
           'FETCH_SIZE_KB_raw_per_iteration': fetch_kb / iters if fetch_kb is not None else None,
           'WRITE_SIZE_KB_raw_per_iteration': write_kb / iters if write_kb is not None else None,
           'note': 'gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM) -> the read '
                   'side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores',
           'hbm_bytes_per_iteration': (2 * fetch_kb + write_kb) * 1024 / iters if fetch_kb is not None and write_kb is not None else None}
    # instruction classes (pmc6 of profile_round.sh) and the co-execution counter
    for c in ('SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_TRANS_F32', 'SQ_INSTS_VALU_INT32',
              'SQ_INSTS_VALU_INT64', 'SQ_INSTS_VALU_CVT'):
        if c in s:
            out[c + '_per_wave_iteration'] = per(c)
    if 'SQ_VALU_MFMA_COEXEC_CYCLES' in s:
        out['SQ_VALU_MFMA_COEXEC_CYCLES_per_launch'] = s['SQ_VALU_MFMA_COEXEC_CYCLES']
        out['SQ_VALU_MFMA_BUSY_CYCLES_per_launch'] = s.get('SQ_VALU_MFMA_BUSY_CYCLES')
    for c in ('SQ_ACTIVE_INST_VALU', 'SQ_INSTS_BRANCH', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_INSTS_SMEM', 'SQ_INSTS_VMEM', 'SQ_ACTIVE_INST_SCA'):
        if c in s:
            out[c + '_per_wave_iteration'] = per(c)
    _dump(out, pmc_name, os.path.join(prof, f'{tag}_pmc_{pmc_name}.json'))
    print(json.dumps(out, indent=1))

kg = [k for k in summary if 'gpmp2_solve_kernel' in k]
if kg:   # GPMP2 C4 (scripts/prof_gpmp2.py: one iteration per launch)
    k = kg[0]
    s = summary[k]
    waves = grid[k][0] // 64
    out = {'kernel': k.split('(')[0], 'workload': 'C4 panda_spheres GPMP2 B=2048 H=128 D=7, one iteration per launch (scripts/prof_gpmp2.py)',
           'waves_per_launch': waves, 'vgpr': grid[k][2], 'lds_bytes': grid[k][4], 'scratch_bytes': grid[k][5]}
    for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_MFMA', 'SQ_INSTS_VMEM', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU',
              'SQ_WAIT_INST_ANY', 'FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64',
              'SQ_INSTS_VALU_TRANS_F64', 'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_INT64', 'SQ_INSTS_VALU_CVT', 'SQ_VALU_MFMA_COEXEC_CYCLES',
              'SQ_VALU_MFMA_BUSY_CYCLES'):
        if c in s:
            out[c + ('_KB_raw_per_launch' if c.endswith('SIZE') else '_per_wave')] = s[c] if c.endswith('SIZE') else s[c] / waves
    _dump(out, 'solve', os.path.join(prof, f'{tag}_pmc_solve.json'))

kl = [k for k in summary if 'gpmp2_lr_' in k or 'gpmp2_pcr_' in k]
if kl:   # GPMP2 C4, low-rank form (round 6; scripts/prof_gpmp2.py: one iteration per launch from the initial state)
    out = {'workload': 'C4 panda_spheres GPMP2 B=2048 H=128 D=7, low-rank form, first iteration from the initial means (scripts/prof_gpmp2.py)', 'kernels': {}}
    for k in kl:
        s = summary[k]
        waves = max(grid[k][0] // 64, 1)
        e = {'waves_per_launch': waves, 'vgpr': grid[k][2], 'lds_bytes': grid[k][4], 'scratch_bytes': grid[k][5]}
        for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_MFMA', 'SQ_INSTS_VMEM', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_ANY',
                  'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'):
            if c in s:
                e[c + '_per_wave'] = s[c] / waves
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            if c in s:
                e[c + '_KB_raw_per_launch'] = s[c]
        out['kernels'][k.split('(')[0]] = e
    _dump(out, 'solve', os.path.join(prof, f'{tag}_pmc_gpmp2_lr.json'))

kc = [k for k in summary if 'chomp_point4_kernel' in k]
if kc:   # CHOMP C2 (scripts/prof_chomp.py: bench.py's c2 entry, MPB_CHOMP_ITERS iterations per launch)
    k = kc[0]
    s = summary[k]
    waves = grid[k][0] // 64
    iters = int(os.environ.get('MPB_CHOMP_ITERS', 500))
    out = {'kernel': re.sub(r'\(.*', '', k) if 're' in dir() else k.split('(')[0], 'workload': 'C2 pointmass_dense_2d CHOMP B=1024 H=64 D=2, %d iterations per launch (scripts/prof_chomp.py)' % iters,
           'waves_per_launch': waves, 'iterations_per_launch': iters, 'vgpr': grid[k][2], 'lds_bytes': grid[k][4], 'scratch_bytes': grid[k][5],
           'SQ_INSTS_VALU_per_wave_iteration': s['SQ_INSTS_VALU'] / waves / iters if 'SQ_INSTS_VALU' in s else None,
           'SQ_INSTS_SALU_per_wave_iteration': s['SQ_INSTS_SALU'] / waves / iters if 'SQ_INSTS_SALU' in s else None,
           'SQ_INSTS_LDS_per_wave_iteration': s['SQ_INSTS_LDS'] / waves / iters if 'SQ_INSTS_LDS' in s else None}
    _dump(out, 'chomp', os.path.join(prof, f'{tag}_pmc_chomp.json'))

km = [k for k in summary if 'mppi_kernel' in k]
if km:   # MPPI, NP = 1024 problems (scripts/prof_mppi.py: bench.py's mppi entry, 50 iterations per launch)
    k = km[0]
    s = summary[k]
    waves = grid[k][0] // 64
    iters = int(os.environ.get('MPB_MPPI_ITERS', 50))
    out = {'kernel': k.split('(')[0], 'workload': 'MPPI point mass, 1024 problems x S=32 x T=64 x c=2, %d iterations per launch (scripts/prof_mppi.py)' % iters,
           'waves_per_launch': waves, 'iterations_per_launch': iters, 'vgpr': grid[k][2], 'lds_bytes': grid[k][4], 'scratch_bytes': grid[k][5]}
    for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'):
        if c in s:
            out[c + '_per_wave_iteration'] = s[c] / waves / iters
    _dump(out, 'mppi', os.path.join(prof, f'{tag}_pmc_mppi.json'))

ka = [k for k in summary if 'stomp_sample_cost' in k and 'true' in k]
if ka:
    k = ka[0]
    s = summary[k]
    waves = grid[k][0] // 64
    fetch_kb, write_kb = s.get('FETCH_SIZE'), s.get('WRITE_SIZE')
    out = {'kernel': k, 'workload': 'C3 P=128 S=32 H=64 d=14 (scripts/prof_stomp.py)', 'waves_per_launch': waves,
           'SQ_INSTS_VALU_per_wave': s.get('SQ_INSTS_VALU', 0) / waves if 'SQ_INSTS_VALU' in s else None,
           'SQ_INSTS_SALU_per_wave': s.get('SQ_INSTS_SALU', 0) / waves if 'SQ_INSTS_SALU' in s else None,
           'SQ_WAVE_CYCLES_per_wave_quads': s.get('SQ_WAVE_CYCLES', 0) / waves if 'SQ_WAVE_CYCLES' in s else None,
           'FETCH_SIZE_KB_raw': fetch_kb, 'WRITE_SIZE_KB_raw': write_kb,
           'note': 'gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM) -> the read '
                   'side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores',
           'hbm_bytes_per_launch': (2 * fetch_kb + write_kb) * 1024 if fetch_kb is not None and write_kb is not None else None}
    _dump(out, 'kernelA', os.path.join(prof, f'{tag}_pmc_kernelA.json'))
    print(json.dumps(out, indent=1))
print('\n'.join(lines[:60]))
