cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity_gpmp2_mppi.py tests/test_gpu_generic_dof.py tests/test_gpu_full_size.py -q -x -k "gpmp2 or Gpmp2 or GPMP2" > gpurun_out/t_gp.log 2>&1; tail -3 gpurun_out/t_gp.log
python scripts/ab_gpmp2_kernels.py > gpurun_out/abk.log 2>&1; tail -1 gpurun_out/abk.log
rocprofv3 --kernel-trace --stats -d gpurun_out/full -o full -- python3 scripts/prof_gpmp2.py > gpurun_out/full.log 2>&1
python - <<'PY'
import sqlite3, collections
c=sqlite3.connect('gpurun_out/full/full_results.db')
d=collections.defaultdict(list)
for r in c.execute("select name, start, end from kernels"):
    if 'gpmp2' in r[0]: d[r[0][:50]].append(r[2]-r[1])
for k,v in d.items():
    v.sort(); print(k, len(v), 'min',v[0],'med',v[len(v)//2], 'max', v[-1])
PY
MPB_LIB_PATH=$PWD/build_variants/clk.so python scripts/prof_gpmp2.py 2>&1 | grep "cap clk" | head -2
