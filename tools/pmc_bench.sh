#!/bin/bash
# Counter profile of the bench's kernels (rocprofv3, separate --pmc passes: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950;
# kernel-trace only, no sys/hip/hsa trace domains).  usage: bash tools/pmc_bench.sh [supervised|masked]
# Output: gpurun_out/pmc_bench_<objective>/summary.json (copy into profiles/ to be judged).
OBJ=${1:-supervised}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bench_$OBJ; rm -rf $O; mkdir -p $O
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-masked --objective $OBJ"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $ARGS > $O/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$T -- python3 $R/bench.py $ARGS > $O/$T.log 2>&1
done
python3 $R/tools/pmc_summary.py $O $OBJ
