"""Tuning aid: where the host side of STOMP.optimize() goes (cProfile over optimize(0) / optimize(1) on the C3 planner)."""
import cProfile, pstats, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
wl, cost, pl = bench.make_stomp(128, 32, dev, 0)
pl.optimize(opt_iters=20); torch.cuda.synchronize()
for k in (0, 1):
    t0 = time.perf_counter()
    for _ in range(2000):
        pl.optimize(opt_iters=k)
    torch.cuda.synchronize()
    print('optimize(%d): %.2f us per call (wall, 2000 calls queued back to back)' % (k, (time.perf_counter() - t0) / 2000 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    pl.optimize(opt_iters=0)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(18)
