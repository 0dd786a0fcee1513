#!/bin/bash
# kernel trace of the plain step and of the same step with the N > 1 machinery on a 1-rank RCCL group (what data parallelism costs without a wire)
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dp_prof; rm -rf $O; mkdir -p $O
A="--steps 3 --warmup 1 --no-cpu-baseline --no-probe --no-masked --no-bf16-saved --no-small --no-fp8-large"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain -- python3 $R/bench.py $A > $O/plain.log 2>&1
# the 1-rank RCCL pass: the rank's environment is set HERE, so the program behind `--` is the rank itself (bench.py starts its own rank
# processes only when WORLD_SIZE is unset -- a launcher parent behind the profiler would be a process hop after the GPU is initialised)
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dp -- python3 $R/bench.py $A --single-rank-collectives > $O/dp.log 2>&1
