"""Build csrc/libmpb_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m motion_planning_baselines_amd.build
"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['mpb_kernels.hip', 'mpb_stomp_fused.hip', 'mpb_stomp_fused_hx.hip', 'mpb_chomp.hip', 'mpb_gpmp2.hip', 'mpb_gpmp2_lr.hip', 'mpb_mppi.hip', 'mpb_prior.hip', 'mpb_stoch_gpmp.hip', 'mpb_costs.hip', 'mpb_points.hip']
# the test aids (include/mpb_debug.h) are a library of their own: the product library exports the product ABI only
DEBUG_SOURCES = ['mpb_debug.hip']
# per-file extra flags: the latency-bound single-wave-per-problem kernels (CHOMP, GPMP2 solve, MPPI) gain 3-10 % from
# LLVM's max-ILP scheduling strategy; the STOMP kernels of mpb_kernels.hip lose 2 % with it (measured, round 1)
MAX_ILP = ['-mllvm', '-amdgpu-sched-strategy=max-ilp']
# the STOMP / stand-alone cost kernels do best with the iterative-ILP strategy (fused step -1.3 %)
EXTRA = {'mpb_chomp.hip': MAX_ILP, 'mpb_gpmp2.hip': MAX_ILP, 'mpb_gpmp2_lr.hip': MAX_ILP, 'mpb_mppi.hip': MAX_ILP,
         'mpb_kernels.hip': ['-mllvm', '-amdgpu-sched-strategy=iterative-ilp'],
         'mpb_stomp_fused.hip': ['-mllvm', '-amdgpu-sched-strategy=iterative-ilp'],
         'mpb_stomp_fused_hx.hip': ['-mllvm', '-amdgpu-sched-strategy=iterative-ilp']}
OUT = os.path.join(CSRC, 'libmpb_hip.so')
DEBUG_OUT = os.path.join(CSRC, 'libmpb_hip_debug.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-ffinite-math-only', '-fno-signed-zeros', '-fno-slp-vectorize']
# what the compiler made of every kernel (registers, spills, scratch, LDS), written next to the library by build(): the
# persistent kernels sit at their register budget, and a change anywhere in their phases can tip loop invariants into scratch
# (round 4: twice) -- tests/test_host_logic.py holds the headline instantiations to 0 B
RESOURCES = os.path.join(CSRC, 'kernel_resources.json')
REMARKS = ['-Rpass-analysis=kernel-resource-usage']


def parse_resource_remarks(text):
    """{mangled kernel name: {vgprs, sgpr_spill, vgpr_spill, scratch, lds, occupancy}} from -Rpass-analysis=kernel-resource-usage."""
    import re
    out, cur = {}, None
    keys = {'VGPRs': 'vgprs', 'AGPRs': 'agprs', 'ScratchSize [bytes/lane]': 'scratch', 'Occupancy [waves/SIMD]': 'occupancy',
            'SGPRs Spill': 'sgpr_spill', 'VGPRs Spill': 'vgpr_spill', 'LDS Size [bytes/block]': 'lds', 'TotalSGPRs': 'sgprs'}
    for line in text.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+) \[-Rpass-analysis', line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


# which sources define the kernels a counter summary (profiles/*_pmc_*.json) was taken for: the summary carries the SHA-1 of these
# files as they were when the passes ran (scripts/pmc_summary.py), bench.py recomputes it and says when the kernel has changed since
PMC_SOURCES = {
    'stomp': ['mpb_stomp_fused.hip', 'mpb_stomp_fused.h', 'mpb_stomp_noise.h', 'mpb_geom.h', 'mpb_common.h', 'mpb_model_panda.h'],
    'stomp_c5': ['mpb_stomp_fused.hip', 'mpb_stomp_fused.h', 'mpb_stomp_noise.h', 'mpb_geom.h', 'mpb_common.h', 'mpb_model_panda.h'],
    'stomp_h128': ['mpb_stomp_fused_hx.hip', 'mpb_stomp_fused.h', 'mpb_stomp_noise.h', 'mpb_geom.h', 'mpb_common.h', 'mpb_model_panda.h'],
    'kernelA': ['mpb_kernels.hip', 'mpb_stomp_noise.h', 'mpb_geom.h', 'mpb_common.h', 'mpb_model_panda.h'],
    'solve': ['mpb_gpmp2.hip', 'mpb_gpmp2_lr.hip', 'mpb_gpmp2.h', 'mpb_geom.h', 'mpb_common.h'],
    'chomp': ['mpb_chomp.hip', 'mpb_geom.h', 'mpb_common.h', 'mpb_model_panda.h'],
    'mppi': ['mpb_mppi.hip', 'mpb_geom.h', 'mpb_common.h'],
}


def sources_sha1(files):
    """SHA-1 over the named csrc files (name and content, in the order given) -- the fingerprint of a kernel's sources."""
    import hashlib
    h = hashlib.sha1()
    for f in files:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_fingerprint(kind):
    files = PMC_SOURCES[kind]
    return {'sources': files, 'sources_sha1': sources_sha1(files), 'flags': FLAGS + EXTRA.get(files[0], [])}


def _stale():
    if not os.path.exists(OUT) or not os.path.exists(DEBUG_OUT) or not os.path.exists(RESOURCES):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(DEBUG_OUT))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))]
    deps += [os.path.join(os.path.dirname(HERE), 'include', h) for h in ('mpb.h', 'mpb_debug.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(out, extra_flags, verbose=False):
    """Tuning aid: build a variant of the library (extra -D flags) to another path -- never to the product path.  The
    wrong-result timing switches (GP_T_*) need -DMPB_TUNING_BUILD, added here; such a library reports itself through
    mpb_version() and _lib.lib() only loads it when asked to by MPB_LIB_PATH."""
    if os.path.abspath(out) == os.path.abspath(OUT):
        raise ValueError('build_variant must not overwrite the product library')
    extra_flags = list(extra_flags)
    if any('_T_' in f for f in extra_flags) and '-DMPB_TUNING_BUILD' not in extra_flags:
        extra_flags.append('-DMPB_TUNING_BUILD')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # per-file flags need per-file compiles
    objs = []
    tmp = out + '.objs'
    os.makedirs(tmp, exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(tmp, src.replace('.hip', '.o'))
        cmd = [hipcc, *FLAGS, *EXTRA.get(src, []), *extra_flags, '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', out])
    return out


def build(force=False, verbose=True):
    """The product library.  Compiles with FLAGS + the per-file scheduling flags only: extra defines (HIPCC_FLAGS-style
    environment hooks do not exist here) cannot reach it, and a tuning build is refused outright."""
    # the compile-time robot models (csrc/mpb_model_*.h) are generated from geometry.py, the single source of the numbers.
    # They are rewritten only when a build is going to run; on the fast path the committed header is only compared
    # (ranks starting together must not write into the package, and it may be installed read-only)
    from . import model_gen
    stale = force or _stale() or bool(model_gen.stale_headers())
    if not stale:
        return OUT
    for path in model_gen.write_headers():
        if verbose:
            print('regenerated', path, flush=True)
    if any('MPB_TUNING_BUILD' in f or '_T_' in f for f in FLAGS + [x for v in EXTRA.values() for x in v]):
        raise RuntimeError('the product build must not carry tuning switches')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in SOURCES + DEBUG_SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *EXTRA.get(src, []), *REMARKS, '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        # (stderr to a file, not a pipe: the resource remarks of the templated kernels overflow a 64 KB pipe buffer, and a
        # compile blocked in write() behind the one being drained serialises the build)
        log = tempfile.TemporaryFile(mode='w+')
        procs.append((cmd, subprocess.Popen(cmd, stderr=log, text=True), log))
    resources = {}
    for cmd, p, log in procs:
        p.wait()
        log.seek(0)
        err = log.read()
        log.close()
        rest = '\n'.join(l for l in err.splitlines() if 'kernel-resource-usage' not in l and not l.lstrip().startswith(('|', '^')) and l.strip()
                         and not __import__('re').match(r'^\s*\d+ \|', l))
        if p.returncode != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd) + '\n' + err[-4000:])
        if verbose and rest:
            print(rest, file=sys.stderr, flush=True)
        resources.update(parse_resource_remarks(err))
    import json
    with open(RESOURCES, 'w') as fh:
        json.dump(resources, fh, indent=0, sort_keys=True)
    n = len(SOURCES)
    for out, o in ((OUT, objs[:n]), (DEBUG_OUT, objs[n:])):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *o, '-o', out]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
