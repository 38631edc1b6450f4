"""Tuning aid: what the GPU's clocks and power read while the persistent STOMP kernel runs long launches at C3
(rocm-smi polled from a thread; needs no privileges)."""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops, workloads
from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril
dev = torch.device('cuda:0')
P, S, H = 128, 32, 64
wl = workloads.panda_spheres_stomp(P, dev, S=S, pos_only=False)
d = wl['means0'].shape[-1]
R = stomp_precision_matrix(H, wl['params']['dt'], 0.1, dict(device='cpu', dtype=torch.float32))
Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev); weights = torch.empty(P, S, device=dev)
ws = ops.stomp_workspace(P, S, H, d, dev)
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if any(k in l for k in ('sclk', 'mclk', 'Power', 'Temperature (Sensor junction)', 'fclk'))]
            print(' | '.join(keep)[:400], flush=True)
        except Exception as e:
            print('rocm-smi failed:', e, flush=True)
            return
        time.sleep(0.3)
print('idle:'); 
th = threading.Thread(target=poll); th.start(); time.sleep(1.0)
print('running:', flush=True)
t0 = time.time()
while time.time() - t0 < 4.0:
    means = wl['means0'].clone()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, ws, n_iters=20000)
    b.record(); torch.cuda.synchronize()
    print('20000 iterations: %.2f us / iteration' % (a.elapsed_time(b) * 1e3 / 20000), flush=True)
stop = True; th.join()
