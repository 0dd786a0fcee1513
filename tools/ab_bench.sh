mkdir -p gpurun_out
T=ecg-representation-learning_amd/libecgvit_hip_tools.so
for r in 1 2; do
  for v in 0 16; do
    echo "== rep $r diag $v" 
    ECGVIT_HIP_LIB=$PWD/$T ECGVIT_NT_DIAG=$v python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"avg_launch_us": [0-9.]*' | tr '\n' ' '
    echo
  done
done
