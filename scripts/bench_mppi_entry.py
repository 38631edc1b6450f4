import sys, json, torch
sys.path.insert(0,'/root/repo')
import bench
dev=torch.device('cuda:0')
for _ in range(3):
    r=bench.bench_mppi(dev,50,with_cpu=False)
    print('mppi', r['value'], r['ms_per_step'])
