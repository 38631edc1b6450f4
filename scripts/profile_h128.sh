#!/bin/bash
# rocprofv3 PMC passes of the `h128` entry's kernel: stomp_fused_hx_kernel<14,1,2> at P = 128, S = 32, H = 128
#   -> profiles/<tag>_pmc_stomp_h128.json, <tag>_h128_pmc_per_wave.md       bash scripts/profile_h128.sh r05
set -e
TAG=${1:-rXX}
export TMPDIR=/tmp MPB_H=128 MPB_ITERS=20 MPB_FUSED=1 MPB_LAUNCHES=12 MPB_PMC_NAME=stomp_h128
export MPB_PMC_WORKLOAD="h128 P=128 S=32 H=128 d=14 (generalised persistent kernel)"
OUT=gpurun_out/prof_hx
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc4.log 2>&1
python3 scripts/pmc_summary.py $OUT ${TAG}_h128 > $OUT/summary.log 2>&1 || true
mv profiles/${TAG}_h128_pmc_stomp_h128.json profiles/${TAG}_pmc_stomp_h128.json
mkdir -p gpurun_out/profiles_h128; cp profiles/${TAG}_pmc_stomp_h128.json profiles/${TAG}_h128_pmc_per_wave.md gpurun_out/profiles_h128/
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
