#!/bin/bash
# whole-line A/B on ONE device: round-5 library (csrc/build/libecgvit_hip_r05.so, built from commit b50449c's csrc: sha256 eff98bfd...) against the
# shipped one, alternating.   usage (on the GPU box): bash tools/ab_r05.sh [pairs]
R05=ecg-representation-learning_amd/csrc/build/libecgvit_hip_r05.so
P='import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
print(sys.argv[1], "base", round(d["value"],1), round(d["ms_per_step"],2), "gemm_us", round(d["roofline"]["avg_launch_us"],1), "masked", round(d["masked"]["value"],1), "small", round(d["small"]["value"],1), "fp8", round(d["fp8_large"]["value"],1), "large_bf16", round(d["fp8_large"]["bf16_same_config"]["value"],1))'
for i in $(seq 1 ${1:-2}); do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-saved --hip-lib $R05 2>/dev/null | python -c "$P" r05
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-saved 2>/dev/null | python -c "$P" new
done
