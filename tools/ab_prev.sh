#!/bin/bash
# whole-line A/B on ONE device: csrc/build/libecgvit_hip_prev.so (`make prev` of the baseline sources) against the shipped library, alternating;
# identical final losses = the two builds computed the same steps bit for bit
# usage (on the GPU box): bash tools/ab_prev.sh [pairs]
PREV=ecg-representation-learning_amd/csrc/build/libecgvit_hip_prev.so
P='import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
print(sys.argv[1], "base", round(d["value"],1), round(d["ms_per_step"],2), "loss", d["final_loss"], "gemm_us", round(d["roofline"]["avg_launch_us"],1), "masked", round(d["masked"]["value"],1), d["masked"]["final_loss"], "small", round(d["small"]["value"],1), d["small"]["final_loss"], "fp8", round(d["fp8_large"]["value"],1), d["fp8_large"]["final_loss"], "large_bf16", round(d["fp8_large"]["bf16_same_config"]["value"],1))'
for i in $(seq 1 ${1:-2}); do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hip-lib $PREV 2>/dev/null | python -c "$P" prev
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P" new
done
