"""The `h128` and `c5` entries of the bench line on their own (tuning aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
for _ in range(2):
    r = bench.bench_h128(dev, 50, with_cpu=False)
    print('h128', r['value'], r['ms_per_step'])
