import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device('cuda:0')
wl, cost, pl = bench.make_stomp(128, 32, dev, 0, H=int(sys.argv[1]) if len(sys.argv) > 1 else 128)
m0 = pl._particle_means.clone()
pl.optimize(opt_iters=50); torch.cuda.synchronize()
def block(k):
    pl._particle_means.copy_(m0); torch.cuda.synchronize()
    t0 = time.perf_counter(); pl.optimize(opt_iters=k); torch.cuda.synchronize()
    w = (time.perf_counter() - t0) * 1e6
    return w, pl._status.device_span_ms() * 1e3
for k in (1, 2, 3, 5, 10, 20, 40):
    r = [block(k) for _ in range(15)]
    w = sorted(x[0] for x in r)[7]; d = sorted(x[1] for x in r)[7]
    print('K=%3d wall %7.1f us  device span %7.1f us  (%.2f us/step)' % (k, w, d, d / k))
