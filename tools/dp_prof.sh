: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dp_prof; rm -rf $O; mkdir -p $O
A="--steps 3 --warmup 1 --no-cpu-baseline --no-probe --no-masked --no-fp8-large"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain -- python3 $R/bench.py $A > $O/plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dp -- python3 $R/bench.py $A --single-rank-collectives > $O/dp.log 2>&1
