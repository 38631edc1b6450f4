"""Driver for rocprofv3: MPPI on 1024 independent point-mass problems, bench.py's own entry (50 iterations per launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.bench_mppi(torch.device('cuda:0'), 50)
print('MPPI NP=1024: %.2f us per iteration of the launch, %.3f us per problem-iteration' % (r['ms_per_step'] * 1e3, r['us_per_problem_iteration']))
