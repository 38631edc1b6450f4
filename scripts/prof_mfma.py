"""The three MFMA users in one process for a PMC pass: STOMP kernel A (f32 16x16x4), GPMP2 solve (f64 16x16x4),
dense GP-prior sampling (f64 16x16x4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy
for s in ('prof_stomp.py', 'prof_gpmp2.py', 'bench_prior.py'):
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), s), run_name='__main__')
