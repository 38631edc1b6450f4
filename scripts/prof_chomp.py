"""Driver for rocprofv3: CHOMP at C2 (B=1024, H=64, D=2), bench.py's own entry (500 iterations per launch, no CPU baseline)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.bench_c2(torch.device('cuda:0'), 500, with_cpu=False)
print('C2 us/iter %.3f' % (r['ms_per_step'] * 1e3))
