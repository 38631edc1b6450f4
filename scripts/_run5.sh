cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in pcr_16x2 pcr_8x2; do
  export MPB_LIB_PATH=$PWD/build_variants/$v.so
  python scripts/ab_gpmp2_kernels.py 2>&1 | tail -1
  rocprofv3 --kernel-trace --stats -d gpurun_out/$v -o full -- python3 scripts/prof_gpmp2.py > gpurun_out/$v.log 2>&1
  python - <<PY
import sqlite3, collections
c=sqlite3.connect('gpurun_out/$v/full_results.db')
d=collections.defaultdict(list)
for r in c.execute("select name, start, end from kernels"):
    if 'pcr' in r[0]: d[r[0][:40]].append(r[2]-r[1])
for k,v in d.items():
    v.sort(); print('$v', k, len(v), 'med',v[len(v)//2])
PY
done
