"""Large-B sweep of the STOMP iteration: B = P x 32 rollouts per iteration from C3's 4 096 up to 2^20, sample
tensors far beyond the 256 MB Infinity Cache.  Persistent launch (both layouts: workgroups per particle exchanging
partials / one workgroup per particle with two batches; MPB_STOMP_BATCHES forces one) and the two-kernel path; HBM and
VALU-issue fractions per size.    python scripts/bench_large_b.py > profiles/rNN_large_b_sweep.txt"""
import json, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def sweep():
    from motion_planning_baselines_amd import ops, workloads
    from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril
    dev = torch.device('cuda:0')
    S, H = 32, 64
    # VALU instructions per ROLLOUT-iteration by layout, from the newest counter summaries (round 6: until then ONE constant, 2 404.5
    # -- round 3's kernel A + B --, priced every line: the round-5 sweep printed 0.648 at C3 where the counters say 0.47):
    #   path 1 (exchange: wave = one rollout per iteration)       profiles/r*_pmc_stomp.json     per wave-iteration
    #   path 2 (two batches: a wave runs TWO rollouts per iteration) profiles/r*_pmc_stomp_c5.json per wave-iteration / 2
    #   two-kernel path: no counter summary of its own since round 2 -> no VALU column
    import bench
    mode = os.environ.get('MPB_MODE', 'persistent')

    def valu_per_rollout(path):
        if mode != 'persistent' or path not in (1, 2):
            return None, None
        pmc, f = bench.latest_profile('r*_pmc_stomp.json' if path == 1 else 'r*_pmc_stomp_c5.json')
        if not pmc or not pmc.get('SQ_INSTS_VALU_per_wave_iteration'):
            return None, None
        v = pmc['SQ_INSTS_VALU_per_wave_iteration'] / (1.0 if path == 1 else 2.0)
        fresh = bench.pmc_freshness(pmc).get('pmc_matches_sources')
        return v, '%s%s' % (f, '' if fresh else (' (STALE: the kernel sources changed since)' if fresh is False else ' (no fingerprint)'))
    for P in (128, 1024, 4096, 8192, 16384, 32768):
        wl = workloads.panda_spheres_stomp(min(P, 1024), dev, S=S, pos_only=False)
        m0 = wl['means0']
        means0 = m0.repeat((P + m0.shape[0] - 1) // m0.shape[0], 1, 1)[:P].contiguous()
        d = means0.shape[-1]
        cpu = dict(device='cpu', dtype=torch.float32)
        R = stomp_precision_matrix(H, wl['params']['dt'], 0.1, cpu)
        Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
        geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
        samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev); weights = torch.empty(P, S, device=dev)
        ws = ops.stomp_workspace(P, S, H, d, dev) if mode == 'persistent' else None
        path = ops.stomp_run_path(geom, ws, P, S, H, d) if ws is not None else 0
        n = 20 if P <= 4096 else 6
        def t(k):
            means = means0.clone()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, ws, n_iters=k)
            torch.cuda.synchronize(); return time.perf_counter() - t0
        t(n)
        dt_ = (min(t(2 * n) for _ in range(3)) - min(t(n) for _ in range(3))) / n
        B = P * S
        alg = 4 * (B * H * d + 2 * P * H * d + 2 * B)
        valu, src = valu_per_rollout(path)
        vtxt = (f'VALU issue {valu*B/dt_/1e9/1228.8:.3f} of peak ({valu:.0f} instr / rollout-iteration, {src})' if valu else 'VALU issue n/a (no counter summary for this path)')
        print(f'{mode:10s} path {path} batches={os.environ.get("MPB_STOMP_BATCHES", "auto"):4s} P={P:6d} B={B:8d} samples {B*H*d*4/1e6:8.1f} MB: '
              f'{dt_*1e6:9.1f} us/iter {B/dt_/1e6:7.1f} M rollouts/s  algorithmic {alg/dt_/1e9:7.1f} GB/s = {alg/dt_/8e12:.3f} of HBM peak, {vtxt}', flush=True)
        del samples

if __name__ == '__main__':
    if os.environ.get('MPB_SWEEP_CHILD'):
        sweep()
    else:     # one child per configuration: the layout override is read once per process
        for env in ({'MPB_MODE': 'persistent'}, {'MPB_MODE': 'persistent', 'MPB_STOMP_BATCHES': '1'},
                    {'MPB_MODE': 'persistent', 'MPB_STOMP_BATCHES': '2'}, {'MPB_MODE': 'two-kernel'}):
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, MPB_SWEEP_CHILD='1', **env), check=True)
