#!/bin/bash
# Everything kept under profiles/ for one round, in one go (run on the GPU box from the repo root; ~6 minutes):
#     bash scripts/profile_all.sh r03
#   * bench.py (the driver's command)                         -> profiles/<tag>_bench.json
#   * rocprofv3 stats + PMC of the headline (profile_round.sh) -> <tag>_kernel_stats_bench.csv, <tag>_pmc_per_wave.md, <tag>_pmc_stomp.json
#   * GPMP2 C4 (profile_gpmp2.sh), CHOMP C2, MPPI NP=1024      -> <tag>_gpmp2_*, <tag>_chomp_*, <tag>_mppi_*
#   * c5 (<14,1,2> at P = 4096) and H = 128 PMC passes        -> <tag>_pmc_stomp_c5.json, <tag>_pmc_stomp_h128.json
#   * large-B sweep                                             -> <tag>_large_b_sweep.txt
#   * shapes beyond H = 64 / gradient evaluators                -> <tag>_other_shapes.txt
# rocprofv3 counter passes run with --kernel-trace only (never with a sys / hip / hsa trace).
set -e
TAG=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/profile_round.sh $TAG 20 > gpurun_out/${TAG}_profile_round.log 2>&1
echo "headline passes done"
bash scripts/profile_gpmp2.sh $TAG > gpurun_out/${TAG}_profile_gpmp2.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/prof_${TAG}_gpmp2 ${TAG}_gpmp2 > /dev/null
python3 scripts/pmc_summary.py gpurun_out/prof_${TAG}_gpmp2_block ${TAG}_gpmp2_block > /dev/null
echo "gpmp2 passes done"
for drv in chomp mppi; do
  OUT=gpurun_out/prof_${TAG}_$drv
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 scripts/prof_$drv.py > $OUT/stats.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python3 scripts/prof_$drv.py > $OUT/pmc1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o p -- python3 scripts/prof_$drv.py > $OUT/pmc2.log 2>&1
  python3 scripts/pmc_summary.py $OUT ${TAG}_$drv > /dev/null
  echo "$drv passes done"
done
bash scripts/profile_c5.sh $TAG > gpurun_out/${TAG}_profile_c5.log 2>&1
bash scripts/profile_h128.sh $TAG > gpurun_out/${TAG}_profile_h128.log 2>&1
echo "c5 / h128 passes done"
python3 scripts/bench_large_b.py > profiles/${TAG}_large_b_sweep.txt 2> gpurun_out/${TAG}_large_b.err
echo "large-B sweep done"
( python3 scripts/bench_hx.py; python3 scripts/bench_grad.py ) > profiles/${TAG}_other_shapes.txt 2> gpurun_out/${TAG}_other.err
# the bench line LAST: it reads the instruction counts of the PMC summaries written above (profiles/<tag>_pmc_*.json)
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
echo "bench done"
# the two-rank rehearsal on this box's one GPU (gloo: ranks share the device; the 8-GPU node runs the same command on RCCL)
MPB_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 > profiles/${TAG}_bench_2ranks_one_gpu_gloo.json 2> gpurun_out/${TAG}_bench2.err || true
echo "two-rank rehearsal done"
# round 6: refuse to finish on stale counters -- every profiles/*_pmc_*.json that bench.py would read (the newest per kind) must carry the
# fingerprint of the kernel sources of THIS tree (scripts/pmc_summary.py stamps it; a file this run did not re-take, or one taken
# before a kernel edit, fails here instead of pricing live times with old instruction counts)
python3 - <<'PYEOF'
import glob, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
bad = []
for pat in ('r*_pmc_stomp.json', 'r*_pmc_stomp_c5.json', 'r*_pmc_stomp_h128.json', 'r*_pmc_solve.json', 'r*_pmc_gpmp2_lr.json', 'r*_pmc_chomp.json', 'r*_pmc_mppi.json'):
    pmc, f = bench.latest_profile(pat)
    if pmc is None:
        continue
    fresh = bench.pmc_freshness(pmc)['pmc_matches_sources']
    print('%-44s %s' % (f, {True: 'matches the kernel sources', False: 'STALE', None: 'no fingerprint (taken before round 6)'}[fresh]))
    if fresh is not True:
        bad.append(f)
if bad:
    sys.exit('profile_all.sh: counter summaries that do not belong to this tree\'s kernels: ' + ', '.join(bad))
PYEOF
# what travels back from the GPU box is gpurun_out/ (<= 64 MiB): the summaries, not the raw rocprofv3 databases
mkdir -p gpurun_out/profiles_${TAG}
cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
cp gpurun_out/prof_${TAG}/stats/*_results.db gpurun_out/profiles_${TAG}/ 2>/dev/null || true
rm -rf gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}_gpmp2 gpurun_out/prof_${TAG}_gpmp2_block gpurun_out/prof_${TAG}_chomp gpurun_out/prof_${TAG}_mppi
echo "all done"
