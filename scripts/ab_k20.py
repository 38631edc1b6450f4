"""Tuning aid: the driver's protocol on the C3 planner -- wall clock of optimize(K) bracketed by synchronize (median of
N blocks, each from the initial means), the same launch by HIP events, and the steady per-iteration time
(t(400) - t(200)) / 200.  MPB_LIB_PATH selects the library build.    python scripts/ab_k20.py [K] [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 41
wl, cost, pl = bench.make_stomp(128, 32, dev, 0)
m0 = pl._particle_means.clone()
pl.optimize(opt_iters=500); torch.cuda.synchronize()
def block(k):
    pl._particle_means.copy_(m0); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record(); pl.optimize(opt_iters=k); b.record(); torch.cuda.synchronize()
    return time.perf_counter() - t0, a.elapsed_time(b) * 1e-3
for _ in range(5): block(K)
w = sorted(block(K) for _ in range(N))
wall = w[N // 2][0]; ev = sorted(x[1] for x in w)[N // 2]
t4 = min(block(400)[1] for _ in range(9)); t2 = min(block(200)[1] for _ in range(9))
# host-side cost of one optimize() call with nothing to run
t0 = time.perf_counter()
for _ in range(200): pl.optimize(opt_iters=0)
host0 = (time.perf_counter() - t0) / 200
print('K=%d wall %.2f us/step (%.1f us/block)  events %.2f us/step  steady %.2f us/iter  optimize(0) host %.1f us'
      % (K, wall / K * 1e6, wall * 1e6, ev / K * 1e6, (t4 - t2) / 200 * 1e6, host0 * 1e6))
