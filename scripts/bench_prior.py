import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops
from motion_planning_baselines_amd.planners.base import gp_prior_factor, gp_prior_scale_tril
dev = torch.device('cuda:0')
H, D, G_, n = 64, 7, 128, 32
f64 = lambda a: torch.as_tensor(a, dtype=torch.float64).to(dev).contiguous()
Ud, Uo = gp_prior_factor(H, 5.0 / H, 1e-3, 0.5, 1e-3)
tril = f64(gp_prior_scale_tril(Ud, Uo)); Ud, Uo = f64(Ud), f64(Uo)
means = torch.zeros(G_, H, 2 * D, dtype=torch.float64, device=dev)
for label, tr in (('dense', tril), ('chain', None)):
    fn = lambda: ops.gp_prior_sample(means, None, Ud, Uo, n, D, seed=1, scale_tril=tr)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{label}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call', flush=True)
