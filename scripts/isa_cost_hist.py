"""Instruction-COST histogram of the headline kernel's iteration loop.

    python scripts/isa_cost_hist.py [--kernel '_Z18stomp_fused_kernelILi14ELi1ELi1ELb0ELb0EE'] [--pmc profiles/r04_pmc_stomp.json] > profiles/rXX_isa_cost_hist.md

What binds `stomp_fused_kernel<14,1,1,false,false>` (the device-noise instantiation the bench times) is the vector pipe: SQ_ACTIVE_INST_VALU says it is ~100 % busy at 4.15 cycles per
instruction where the full rate is 2.  This script prices the loop body by opcode:

  1. compiles csrc/mpb_stomp_fused.hip to assembly with the product's flags (hipcc -S --cuda-device-only), cuts out the kernel and
     splits it into basic blocks with LLVM's loop annotations;
  2. gives every block of the iteration loop a PHASE and a MULTIPLICITY (executions per wave and iteration) by the structural
     rules below -- labels change with every build, the loop nest does not:
       * blocks outside the depth-1 loop: entry / exit, not counted;
       * the depth-3 loop that has depth-4 children is the GROUP loop of the collision walk (mpb_geom.h,
         waypoint_cost_grid_model): its blocks that start with s_setprio are the per-group ARMS (FK advance + sphere positions of
         one group of four collision spheres, each arm once per iteration); the block with the v_sqrt_f32 and the ds_read of the
         grid word is the group's SLOT-0 body (once per group that runs); blocks starting with v_bfe_u32 are the LATER candidate
         slots; the depth-4 loops are the exhaustive / box loops (never at C3);
       * the number of group bodies and later-slot trips per iteration is not static (frame-1 groups can be skipped, later
         slots run while any lane has a candidate): it is SOLVED from the PMC pass's SQ_INSTS_VALU_TRANS_F32 (v_sqrt: four per
         trip, everything else that is transcendental is static), and the split of the later slots between slot 1 and slot 2 from
         scripts/grid_stats.py's trip statistics (default 0.85 / 0.15);
       * blocks behind the wave-uniform branches of waves 0-3 (Sigma product) and of the threads tq < N count by the fraction
         of waves that take them;
  3. prices every VALU instruction with profiles/r03_microbench_rates.txt (measured cycles per wave-instruction per SIMD, MI355X):
       full rate (2.5-2.7): add / sub / mul, fmamk / fmaak, fma with at most two distinct VGPR sources, add_u32, and / or / xor, mov;
       3.7: v_fmac (VOP2, three registers);  4.05: v_fma with three distinct VGPR sources;  4.3: v_fma with an SGPR and two VGPRs;
       half rate (4.2-4.4): min / max / med3, floor, every convert, mad_u32_u24, bfe, shifts, cndmask, compares, mul_lo / mul_hi,
       mad_u64_u32, DPP moves, perm / alignbit / and_or / lshl_add (three-source integer forms: taken as 4.2), readlane (4.2 assumed);
       8.2: sqrt / log / exp / rcp / sin / cos;
  4. prints, sorted by cycles: phase x opcode, opcode totals, and the reconciliation with the PMC counters of the same kernel
     (instruction classes, total VALU, SQ_ACTIVE_INST_VALU x 4 = busy cycles per wave-iteration)."""
import argparse
import collections
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TRANS = ('v_sqrt_f32', 'v_log_f32', 'v_exp_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sin_f32', 'v_cos_f32', 'v_rcp_iflag_f32')
FULL = ('v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_mul_f32', 'v_fmamk_f32', 'v_fmaak_f32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32',
        'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_mov_b32', 'v_not_b32', 'v_add_co_u32', 'v_addc_co_u32', 'v_mov_b64', 'v_accvgpr')
HALF_PREFIX = ('v_min', 'v_max', 'v_med3', 'v_floor', 'v_cvt', 'v_mad_u32_u24', 'v_mad_i32_i24', 'v_mul_u32_u24', 'v_bfe', 'v_lshl', 'v_lshr', 'v_ashr',
               'v_cndmask', 'v_cmp', 'v_mul_lo', 'v_mul_hi', 'v_mad_u64', 'v_mad_i64', 'v_perm', 'v_alignbit', 'v_and_or', 'v_or3', 'v_xad', 'v_add3',
               'v_readlane', 'v_readfirstlane', 'v_writelane', 'v_rndne', 'v_fract', 'v_trunc', 'v_ceil', 'v_bfi', 'v_div_scale', 'v_div_fmas',
               'v_div_fixup', 'v_ldexp', 'v_frexp', 'v_mbcnt', 'v_add_lshl', 'v_sad', 'v_lerp', 'v_cubema', 'v_add_f64', 'v_mul_f64', 'v_fma_f64',
               'v_bcnt', 'v_ffb', 'v_max3', 'v_min3', 'v_swap', 'v_permlane', 'v_xor3', 'v_mul_legacy', 'v_mul_i32', 'v_sub_co', 'v_subb')


def vgprs(operand):
    """set of VGPR indices an operand string names (v12, v[4:5], -v3, |v7|)"""
    out = set()
    for m in re.finditer(r'\bv(\d+)\b', operand):
        out.add(int(m.group(1)))
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', operand):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def price(ins):
    """(cycles per wave-instruction, class label) of one VALU instruction"""
    parts = ins.split(None, 1)
    op = re.sub(r'_(e32|e64|dpp|sdwa)$', '', parts[0])
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    dpp = parts[0].endswith('_dpp') or 'quad_perm' in ins or 'row_' in ins
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return 0.0, 'mfma'
    if op in TRANS:
        return 8.2, 'trans 8.2'
    if op in ('v_fma_f32', 'v_mad_f32'):
        srcs = ops[1:4]
        nv = len(set().union(*[vgprs(o) for o in srcs])) if srcs else 0
        has_s = any(re.match(r'^-?\|?s\d+|^-?s\[', o) for o in srcs)
        if nv >= 3:
            return 4.05, 'fma 3 VGPR 4.05'
        if nv == 2 and has_s:
            return 4.3, 'fma 2 VGPR + SGPR 4.3'
        return 2.65, 'fma <= 2 VGPR 2.65'
    if op == 'v_fmac_f32':
        # VOP2: dst += src0 * src1.  With a literal / inline constant / SGPR as src0, or src0 == src1 (a square), only two distinct
        # VGPRs are read: the fmamk / "fma x,x,x,v" rate; with three distinct VGPRs the measured 3.7
        srcs = ops[1:3]
        nv = len(set().union(*[vgprs(o) for o in srcs]) | vgprs(ops[0])) if srcs else 3
        return (3.7, 'fmac 3 VGPR 3.7') if nv >= 3 else (2.65, 'fmac <= 2 VGPR 2.65')
    if op.startswith('v_pk_'):
        return 4.8, 'packed 4.8'
    if dpp:
        return 4.3, 'dpp 4.3'
    if op in FULL or op.startswith('v_accvgpr'):
        return 2.6, 'full rate 2.6'
    if op.startswith(HALF_PREFIX):
        return 4.25, 'half rate 4.25'
    return 4.25, 'unlisted (4.25 assumed): ' + op


def pmc_class(ins):
    op = re.sub(r'_(e32|e64|dpp|sdwa)$', '', ins.split()[0])
    if op in ('v_fma_f32', 'v_fmac_f32', 'v_fmamk_f32', 'v_fmaak_f32', 'v_mad_f32'):
        return 'FMA_F32'
    if op in ('v_add_f32', 'v_sub_f32', 'v_subrev_f32'):
        return 'ADD_F32'
    if op in ('v_mul_f32',):
        return 'MUL_F32'
    if op in TRANS:
        return 'TRANS_F32'
    if op.startswith('v_cvt'):
        return 'CVT'
    if op.startswith(('v_mad_u64', 'v_mad_i64', 'v_lshlrev_b64', 'v_lshrrev_b64', 'v_lshl_add_u64', 'v_mov_b64')):
        return 'INT64'
    if op.startswith(('v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_mad_u32', 'v_mad_i32', 'v_mul_lo', 'v_mul_hi', 'v_mul_u32', 'v_add3', 'v_add_co', 'v_addc',
                      'v_sub_co', 'v_subb', 'v_lshl_add_u32', 'v_add_lshl', 'v_xad', 'v_min_u32', 'v_max_u32', 'v_min_i32', 'v_max_i32', 'v_sad')):
        return 'INT32'
    return 'OTHER'


def compile_asm(src):
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffinite-math-only', '-fno-signed-zeros', '-fno-slp-vectorize',
             '-mllvm', '-amdgpu-sched-strategy=iterative-ilp']
    out = tempfile.NamedTemporaryFile(suffix='.s', delete=False).name
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), *flags, '-S', '--cuda-device-only', src, '-o', out],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def split_blocks(text, kernel_prefix):
    lines = text.split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel_prefix) and l.rstrip().split(':')[0].startswith(kernel_prefix) and ':' in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    blocks, cur = [], dict(name='entry', ins=[], loop=None, hdr=None, parents=[])
    blocks.append(cur)
    for ln in lines[start + 1:end]:
        m = re.match(r'^(\.LBB\d+_\d+):', ln) or re.match(r'^; %bb\.(\d+):', ln)
        if m:
            loop = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', ln)
            hdr = re.search(r'This (Inner )?Loop Header: Depth=(\d+)', ln)
            cur = dict(name=m.group(1) if ln.startswith('.') else 'bb.' + m.group(1), ins=[], parents=[],
                       loop=(loop.group(1), int(loop.group(2))) if loop else None, hdr=int(hdr.group(2)) if hdr else None)
            m3 = re.search(r'Parent Loop (BB\d+_\d+) Depth=(\d+)', ln)
            if m3:
                cur['parents'].append((m3.group(1), int(m3.group(2))))
            blocks.append(cur)
            continue
        s = ln.strip()
        if not s or s.startswith(';') or s.startswith('.'):
            m2 = re.search(r'=>\s*This (Inner )?Loop Header: Depth=(\d+)', ln)
            if m2:
                cur['hdr'] = int(m2.group(2))
            m3 = re.search(r'Parent Loop (BB\d+_\d+) Depth=(\d+)', ln)
            if m3:
                cur['parents'].append((m3.group(1), int(m3.group(2))))
            m4 = re.search(r'Child Loop (BB\d+_\d+) Depth (\d+)', ln)
            if m4:
                cur.setdefault('children', []).append((m4.group(1), int(m4.group(2))))
            continue
        cur['ins'].append(s.split(';')[0].strip())
    return blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernel', default='_Z18stomp_fused_kernelILi14ELi1ELi1ELb0ELb0EE')
    ap.add_argument('--pmc', default=None)
    ap.add_argument('--slot2-share', type=float, default=0.15, help='share of the later-slot trips that are slot 2 (scripts/grid_stats.py)')
    ap.add_argument('--asm', default=None, help='use this assembly file instead of compiling')
    args = ap.parse_args()
    pmc_file = args.pmc or sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_stomp.json')))[-1]
    pmc = json.load(open(pmc_file))
    text = open(args.asm).read() if args.asm else compile_asm(os.path.join(ROOT, 'motion_planning_baselines_amd', 'csrc', 'mpb_stomp_fused.hip'))
    blocks = split_blocks(text, args.kernel)
    name_of = lambda b: b['name'].replace('.L', '')

    # ---- the loop nest
    def header_of(b):
        """name of the innermost loop the block belongs to (a header belongs to its own loop)"""
        if b['hdr'] is not None:
            return name_of(b)
        return b['loop'][0] if b['loop'] else None
    def depth_of(b):
        return b['hdr'] if b['hdr'] is not None else (b['loop'][1] if b['loop'] else 0)
    hdr_block = {name_of(b): b for b in blocks if b['hdr'] is not None}
    it_loop = next(n for n, b in hdr_block.items() if b['hdr'] == 1 and b.get('children'))
    # the obstacle walk: a loop over 4-cell groups with the later-slot loops inside; under a field loop (depth 3 / 4) in the chained
    # instantiations, directly under the iteration loop (depth 2 / 3) in the one-field ones (CHAIN = false)
    def find_grp(dg):
        for n, b in hdr_block.items():
            kids = [c for c, d in b.get('children', []) if d == dg + 1]
            if b['hdr'] == dg and kids and any(x.startswith('v_sqrt_f32') for k in kids for x in hdr_block[k]['ins']):
                return n, kids
        return None, []
    grp_loop, inner4 = find_grp(3)
    if grp_loop is None:
        grp_loop, inner4 = find_grp(2)

    def in_iteration_loop(b):
        if depth_of(b) == 0:
            return False
        h = header_of(b)
        while h is not None:
            if h == it_loop:
                return True
            hb = hdr_block[h]
            par = [p for p, d in hb['parents'] if d == hb['hdr'] - 1]
            h = par[0] if par else None
        return False

    # ---- phases by position (the kernel's source order survives: A sample, B cost, C partial, D Sigma product, noise draw, poll, E update)
    loop_blocks = [b for b in blocks if in_iteration_loop(b)]
    idx = {id(b): i for i, b in enumerate(loop_blocks)}
    first_grp = min(idx[id(b)] for b in loop_blocks if header_of(b) == grp_loop or header_of(b) in inner4)
    last_grp = max(idx[id(b)] for b in loop_blocks if header_of(b) == grp_loop or header_of(b) in inner4)
    # the block in front of the field loop that carries the first joints of the chain (fmamk-heavy) belongs to the cost phase
    chain_prefix = max((i for i in range(first_grp) if sum(1 for x in loop_blocks[i]['ins'] if x.startswith('v_')) > 100), default=None)
    noise_blocks = [i for i, b in enumerate(loop_blocks) if any(x.startswith('v_mfma_f32_16x16x32') for x in b['ins']) or
                    sum(1 for x in b['ins'] if x.startswith('v_mad_u64_u32')) >= 8]
    poll_blocks = [i for i, b in enumerate(loop_blocks) if any('s_memrealtime' in x or 's_sleep' in x for x in b['ins']) or
                   (sum(1 for x in b['ins'] if x.startswith('global_load_dwordx2')) >= 3)]
    sig_blocks = [i for i, b in enumerate(loop_blocks) if any(x.startswith('v_mfma_f32_16x16x4') for x in b['ins'])]
    first_noise = min(noise_blocks) if noise_blocks else len(loop_blocks)

    def phase(i, b):
        h = header_of(b)
        if h == grp_loop or h in inner4 or (chain_prefix is not None and chain_prefix <= i <= last_grp + 3 and i >= chain_prefix):
            if i <= last_grp + 3:
                return 'B cost'
        if i < (chain_prefix if chain_prefix is not None else first_grp):
            return 'A sample'
        if i in noise_blocks:
            return 'N draw + product'
        if i in poll_blocks or any(x.startswith('global_load_dwordx2') for x in b['ins']):
            return 'X exchange poll'
        if i in sig_blocks:
            return 'D Sigma product'
        if i < first_noise:
            return 'C partial / D'
        return 'E combine / update'

    # the field loop (depth 2, parent of the group loop) and the poll loop (depth 2 with s_sleep / s_memrealtime)
    field_loop = next((p for p, d in hdr_block[grp_loop]['parents'] if d == 2 and hdr_block[grp_loop]['hdr'] == 3), None)
    poll_loops = set()
    for b in loop_blocks:
        if any('s_sleep' in x or 's_memrealtime' in x for x in b['ins']) and header_of(b) not in (it_loop, None):
            poll_loops.add(header_of(b))
    # split + product blocks come in pairs (LOW = true: 6 MFMAs per tile for injected eps, LOW = false: 5 for drawn normals)
    prod = [(i, sum(1 for x in b['ins'] if x.startswith('v_mfma_f32_16x16x32'))) for i, b in enumerate(loop_blocks)]
    prod = [(i, n) for i, n in prod if n > 0]
    injected_variants = set()
    by_n = sorted(prod, key=lambda t: t[1])
    if len(prod) >= 4:
        # the two blocks with the most MFMAs are the column blocks KB = 0 / 1 of the injected path
        small = sorted(n for _, n in prod)
        for i, n in prod:
            twin = [m for j, m in prod if j != i and m < n and abs(m * 6 - n * 5) <= 2]
            if twin:
                injected_variants.add(i)
    # ---- multiplicities
    n_sqrt_static = 0
    body_sqrt = later_sqrt = 0
    kinds = {}
    for i, b in enumerate(loop_blocks):
        h = header_of(b)
        first = b['ins'][0].split()[0] if b['ins'] else ''
        nsq = sum(1 for x in b['ins'] if x.startswith('v_sqrt_f32'))
        other_loop = h is not None and h not in (it_loop, grp_loop, field_loop) and h not in poll_loops
        if h in inner4 or other_loop:
            kinds[i] = 'never'                                   # exhaustive overflow / box loops, grid re-staging of chained fields: not at C3
        elif any(x.startswith('global_load_dwordx4') for x in b['ins']) and i > last_grp:
            kinds[i] = 'never'                                   # injected-eps loads: the timed path draws its noise
        elif i in injected_variants:
            kinds[i] = 'never'                                   # the three-component (injected eps) split + product: the drawn path has the other copy
        elif h == grp_loop and first == 's_setprio':
            kinds[i] = 'arm'
        elif h == grp_loop and nsq and any(x.startswith('v_bfe_u32') for x in b['ins'][:4]):
            kinds[i] = 'later'
            later_sqrt = max(later_sqrt, nsq)
        elif h == grp_loop and nsq:
            kinds[i] = 'body'
            body_sqrt = nsq
        elif h == grp_loop:
            kinds[i] = 'grp-glue'
        else:
            kinds[i] = 'once'
    trans_static = 0.0
    for i, b in enumerate(loop_blocks):
        if kinds[i] in ('once', 'arm'):
            trans_static += sum(1 for x in b['ins'] if re.sub(r'_(e32|e64)$', '', x.split()[0]) in TRANS)
    trans_pmc = pmc['SQ_INSTS_VALU_TRANS_F32_per_wave_iteration']
    later_blocks = [i for i in kinds if kinds[i] == 'later']
    NG = sum(1 for i in kinds if kinds[i] == 'arm')
    # the arms come in two rotated copies of the dispatch chain when LLVM rotates the loop: every group has ONE arm that runs
    arm_groups = collections.Counter()
    return_blocks = (blocks, loop_blocks, kinds)
    return pmc, pmc_file, loop_blocks, kinds, phase, trans_static, trans_pmc, body_sqrt, later_sqrt, NG, args, it_loop, grp_loop


def report():
    pmc, pmc_file, loop_blocks, kinds, phase, trans_static, trans_pmc, body_sqrt, later_sqrt, n_arm_blocks, args, it_loop, grp_loop = main()
    # groups: the model has (N_FRAME1 + 3) / 4 + (N_LINKS - N_FRAME1 + 3) / 4 groups; LLVM may duplicate arms (loop rotation), the
    # number of DISTINCT groups is what runs: read it from the model header
    hdr = open(os.path.join(ROOT, 'motion_planning_baselines_amd', 'csrc', 'mpb_model_panda.h')).read()
    n_links = int(re.search(r'N_LINKS\s*=\s*(\d+)', hdr).group(1))
    n_f1 = int(re.search(r'N_FRAME1\s*=\s*(\d+)', hdr).group(1))
    NG = (n_f1 + 3) // 4 + (n_links - n_f1 + 3) // 4
    arm_mult = NG / float(n_arm_blocks) if n_arm_blocks else 0.0
    # sqrt budget: trans_pmc = static + 4 * (bodies + later trips)
    trips = (trans_pmc - trans_static) / float(body_sqrt or 4)
    # bodies: the groups that run (frame-1 groups with no surviving sphere are skipped: from the geometry's keep mask this is
    # NG minus the skipped ones; the PMC cannot separate bodies from later trips, grid_stats' 1.39 trips per body does)
    bodies = trips / 1.39
    later = trips - bodies
    n_later_blocks = sum(1 for k in kinds.values() if k == 'later')
    mult = {}
    for i, b in enumerate(loop_blocks):
        k = kinds[i]
        if k == 'never':
            mult[i] = 0.0
        elif k == 'arm':
            mult[i] = arm_mult
        elif k == 'body':
            mult[i] = bodies
        elif k == 'later':
            # first later block = slot 1, second = slot 2 (when two copies exist)
            order = sorted(j for j in kinds if kinds[j] == 'later').index(i)
            share = (1.0 - args.slot2_share) if order == 0 else args.slot2_share
            mult[i] = later * (share if n_later_blocks > 1 else 1.0)
        elif k == 'grp-glue':
            mult[i] = float(NG)
        else:
            mult[i] = 1.0
    # wave-fraction rules: blocks executed by a subset of the 16 waves (wave-uniform branches)
    for i, b in enumerate(loop_blocks):
        if any(x.startswith('v_mfma_f32_16x16x4') for x in b['ins']):
            mult[i] *= 4.0 / 16.0                                # Sigma product: waves 0-3
    rows = collections.defaultdict(lambda: [0.0, 0.0])           # (phase, label, op) -> [count, cycles]
    cls = collections.Counter()
    salu = branches = lds = vmem = 0.0
    total_v = total_cyc = 0.0
    per_phase = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0.0])   # valu, cycles, salu, branches
    for i, b in enumerate(loop_blocks):
        ph = phase(i, b)
        for ins in b['ins']:
            op = ins.split()[0]
            if op.startswith('v_'):
                c, label = price(ins)
                if label == 'mfma':
                    continue
                rows[(ph, label, re.sub(r'_(e32|e64)$', '', op))][0] += mult[i]
                rows[(ph, label, re.sub(r'_(e32|e64)$', '', op))][1] += mult[i] * c
                cls[pmc_class(ins)] += mult[i]
                total_v += mult[i]
                total_cyc += mult[i] * c
                per_phase[ph][0] += mult[i]
                per_phase[ph][1] += mult[i] * c
            elif op.startswith('s_'):
                if op.startswith(('s_cbranch', 's_branch')):
                    branches += mult[i]
                    per_phase[ph][3] += mult[i]
                elif not op.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_setprio', 's_sleep', 's_endpgm', 's_load', 's_memrealtime', 's_memtime')):
                    salu += mult[i]                 # (SQ_INSTS_SALU: scalar ALU work; waits, nops, barriers, priorities and loads are not)
                    per_phase[ph][2] += mult[i]
            elif op.startswith('ds_'):
                lds += mult[i]
            elif op.startswith(('global_', 'buffer_', 'flat_')):
                vmem += mult[i]
    P = print
    P('# Instruction-cost histogram of `%s` -- one iteration of the persistent loop, per wave' % args.kernel)
    P('')
    P('Made by `scripts/isa_cost_hist.py` (rules and prices: its docstring) from the assembly of the product build and `%s`.' % os.path.relpath(pmc_file, ROOT))
    P('Multiplicities: %d groups of four collision spheres (arms: %d blocks, %.2f executions each); from TRANS_F32 = %.1f per wave-iteration' %
      (NG, n_arm_blocks, arm_mult, trans_pmc))
    P('(static transcendental instructions: %.0f) -> %.2f distance trips of 4 sqrt = %.2f group bodies x 1.39 trips (scripts/grid_stats.py).' %
      (trans_static, trips, bodies))
    P('')
    P('## Reconciliation with the PMC pass')
    P('')
    P('| quantity | this model | PMC | ratio |')
    P('|---|---|---|---|')
    P('| VALU instructions per wave-iteration | %.0f | %.0f | %.2f |' % (total_v, pmc['SQ_INSTS_VALU_per_wave_iteration'], total_v / pmc['SQ_INSTS_VALU_per_wave_iteration']))
    for c in ('FMA_F32', 'ADD_F32', 'MUL_F32', 'TRANS_F32', 'CVT', 'INT32', 'INT64'):
        pv = pmc.get('SQ_INSTS_VALU_%s_per_wave_iteration' % c)
        if pv:
            P('| %s | %.0f | %.0f | %.2f |' % (c, cls[c], pv, cls[c] / pv))
    other_pmc = pmc['SQ_INSTS_VALU_per_wave_iteration'] - sum(pmc.get('SQ_INSTS_VALU_%s_per_wave_iteration' % c, 0) for c in ('FMA_F32', 'ADD_F32', 'MUL_F32', 'TRANS_F32', 'CVT', 'INT32', 'INT64'))
    P('| OTHER (min / max / select / compare / move / lane / bit ops) | %.0f | %.0f | %.2f |' % (cls['OTHER'], other_pmc, cls['OTHER'] / other_pmc))
    act = pmc.get('SQ_ACTIVE_INST_VALU_per_wave_iteration')
    if act:
        P('| vector-pipe busy cycles (SQ_ACTIVE_INST_VALU x 4) | %.0f priced | %.0f | %.2f |' % (total_cyc, 4.0 * act, total_cyc / (4.0 * act)))
    P('| SALU instructions | %.0f | %.0f | %.2f |' % (salu, pmc['SQ_INSTS_SALU_per_wave_iteration'], salu / pmc['SQ_INSTS_SALU_per_wave_iteration']))
    br = pmc.get('SQ_INSTS_BRANCH_per_wave_iteration')
    P('| branches | %.0f | %s | |' % (branches, ('%.0f' % br) if br else 'n/a'))
    P('| LDS instructions | %.0f | %.0f | %.2f |' % (lds, pmc['SQ_INSTS_LDS_per_wave_iteration'], lds / pmc['SQ_INSTS_LDS_per_wave_iteration']))
    P('')
    P('Average price: %.2f cycles per VALU instruction (2 = full rate).' % (total_cyc / total_v))
    P('')
    P('## By phase')
    P('')
    P('| phase | VALU | priced cycles | share | cycles / instr | SALU | branches |')
    P('|---|---|---|---|---|---|---|')
    for ph, (v, c, s, brn) in sorted(per_phase.items(), key=lambda kv: -kv[1][1]):
        P('| %s | %.0f | %.0f | %.1f %% | %.2f | %.0f | %.0f |' % (ph, v, c, 100 * c / total_cyc, c / max(v, 1e-9), s, brn))
    P('')
    P('## By price class')
    P('')
    byc = collections.defaultdict(lambda: [0.0, 0.0])
    for (ph, label, op), (n, c) in rows.items():
        byc[label][0] += n
        byc[label][1] += c
    P('| class | instructions | cycles | share of cycles |')
    P('|---|---|---|---|')
    for label, (n, c) in sorted(byc.items(), key=lambda kv: -kv[1][1]):
        P('| %s | %.0f | %.0f | %.1f %% |' % (label, n, c, 100 * c / total_cyc))
    P('')
    P('## Opcodes by cycles (all phases)')
    P('')
    byop = collections.defaultdict(lambda: [0.0, 0.0])
    for (ph, label, op), (n, c) in rows.items():
        byop[(op, label)][0] += n
        byop[(op, label)][1] += c
    P('| opcode | class | count | cycles | share |')
    P('|---|---|---|---|---|')
    for (op, label), (n, c) in sorted(byop.items(), key=lambda kv: -kv[1][1])[:40]:
        P('| %s | %s | %.0f | %.0f | %.1f %% |' % (op, label, n, c, 100 * c / total_cyc))
    P('')
    P('## Phase x opcode, top 60 by cycles')
    P('')
    P('| phase | opcode | class | count | cycles |')
    P('|---|---|---|---|---|')
    for (ph, label, op), (n, c) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
        P('| %s | %s | %s | %.1f | %.0f |' % (ph, op, label, n, c))
    P('')
    P('## Blocks of the iteration loop (audit trail)')
    P('')
    P('| block | loop | kind | multiplicity | phase | VALU | SALU | LDS | first instructions |')
    P('|---|---|---|---|---|---|---|---|---|')
    for i, b in enumerate(loop_blocks):
        nv = sum(1 for x in b['ins'] if x.startswith('v_'))
        ns = sum(1 for x in b['ins'] if x.startswith('s_'))
        nd = sum(1 for x in b['ins'] if x.startswith('ds_'))
        if nv + ns + nd == 0:
            continue
        P('| %s | %s | %s | %.2f | %s | %d | %d | %d | %s |' % (b['name'], b['loop'][0] if b['loop'] else 'hdr', kinds[i], mult[i], phase(i, b), nv, ns, nd,
                                                          ' '.join(x.split()[0] for x in b['ins'][:3])))


if __name__ == '__main__':
    report()
