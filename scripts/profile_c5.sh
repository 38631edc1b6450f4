#!/bin/bash
# rocprofv3 PMC passes of the `c5` entry's kernel: stomp_fused_kernel<14,1,2> (two-batch layout: one workgroup per particle)
# at 4096 particles x S = 32 (BASELINE configs[4]'s per-GPU load) -> profiles/<tag>_pmc_stomp_c5.json, <tag>_c5_pmc_per_wave.md
#   bash scripts/profile_c5.sh r05
set -e
TAG=${1:-rXX}
export TMPDIR=/tmp MPB_P=4096 MPB_ITERS=20 MPB_FUSED=1 MPB_LAUNCHES=8 MPB_PMC_NAME=stomp_c5
export MPB_PMC_WORKLOAD="c5 P=4096 S=32 H=64 d=14 (131072 rollouts, two-batch layout)"
OUT=gpurun_out/prof_c5
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc1.log 2>&1
echo "pmc1 done"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc2.log 2>&1
echo "pmc2 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc4.log 2>&1
echo "pmc3/4 done"
python3 scripts/pmc_summary.py $OUT ${TAG}_c5 > $OUT/summary.log 2>&1 || true
mv profiles/${TAG}_c5_pmc_stomp_c5.json profiles/${TAG}_pmc_stomp_c5.json
mkdir -p gpurun_out/profiles_c5; cp profiles/${TAG}_pmc_stomp_c5.json profiles/${TAG}_c5_pmc_per_wave.md gpurun_out/profiles_c5/
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
