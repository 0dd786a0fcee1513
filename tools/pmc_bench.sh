#!/bin/bash
# Counter profile of the bench's kernels (rocprofv3, separate --pmc passes: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950;
# kernel-trace only, no sys/hip/hsa trace domains).
#   usage: bash tools/pmc_bench.sh TAG [bench.py args that select the workload, e.g. --objective masked | --config small | --config large --patch 10 --dtype fp8 --batch 256]
# Output directory: gpurun_out/pmc_<TAG>_<kernel_source_sha16>/ -- named by the hash of the sources profiled, so a directory pulled back
# by an earlier call can never be mistaken for this one.  It holds summary.json AND kernel_stats.csv, the `--stats` table of the SAME
# trace pass the summary's durations come from: copy both into profiles/ (tools/check_profiles.py holds them against each other).
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
TAG=${1:?usage: pmc_bench.sh TAG [bench args]}; shift
R=$GRAFT_REPO_ROOT
# the profiled program must be the rank itself: bench.py would start child rank processes for these flags unless the rank environment is set
case " $* " in *" --single-rank-collectives "*|*" --gpus "[2-9]*) [ -n "$WORLD_SIZE" ] || { echo "pmc_bench.sh: set RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT for $* (no launcher hop behind the profiler)" >&2; exit 2; };; esac
SHA=$(cd $R && python3 -c "import bench; print(bench.kernel_source_hash())") || exit 1
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_${TAG}_${SHA}; rm -rf "$O"; mkdir -p "$O"
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-masked --no-bf16-saved --no-small --no-fp8-large $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $ARGS > $O/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES"; do
  T=$(echo $C | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$T -- python3 $R/bench.py $ARGS > $O/$T.log 2>&1
done
python3 $R/tools/pmc_summary.py $O "$TAG" $ARGS
