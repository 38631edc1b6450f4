import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
from conftest import load_golden, product_geometry_from_golden, ref_geometry_from_golden
from motion_planning_baselines_amd import ops
from oracle import planners_ref as O
T=torch.from_numpy
dev=torch.device('cuda:0')
for name in ['chomp_pm2d_dense','chomp_pm2d_soft','chomp_panda']:
    g=load_golden(name)
    robot, field = product_geometry_from_golden(g)
    geom=ops.DeviceGeometry(robot, field, dev)
    R=T(g['R']).to(dev)
    kw = dict(D=int(g['D']), k_sigma=1.0 / float(g['sigma_coll']) ** 2, weight=float(g['weight']), w_prior=float(g['w_prior']), lr=float(g['lr']), grad_clip=float(g['clip']))
    means = T(g['means0']).clone().to(dev)
    for it in range(g['means'].shape[0]):
        prev = means.clone()
        ops.chomp_step(means, R, geom, n_iters=1, **kw)
        torch.cuda.synchronize()
        ref = T(g['means'][it]); 
        d = (means.cpu()-ref).abs()
        idx = np.unravel_index(int(d.argmax()), d.shape)
        print(name, it, 'max abs err', float(d.max()), 'at', idx, 'gpu', float(means.cpu()[idx]), 'ref', float(ref[idx]), 'prev', float(prev.cpu()[idx]))
        means = ref.clone().to(dev)
    # gradient check at means0
    rr, rf = ref_geometry_from_golden(g)
    x = T(g['means0']).clone().requires_grad_(True)
    c = O.collision_cost(x, rr, rf, float(g['sigma_coll']), weight=float(g['weight'])); c.sum().backward()
    out, grad = ops.cost_collision_grad(T(g['means0']).to(dev), geom, kw['k_sigma'], weight=kw['weight'])
    print(name, 'grad err', float((grad.cpu()-x.grad).abs().max()), 'grad max', float(x.grad.abs().max()), 'cost err', float((out.cpu()-c.detach()).abs().max()))
