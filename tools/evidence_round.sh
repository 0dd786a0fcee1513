#!/bin/bash
# Every hash-tied piece of a round's evidence in ONE gpurun call on ONE device (tools/check_profiles.py holds them to each other):
#   /usr/local/graft/bin/gpurun --timeout 3300 -- 'bash tools/evidence_round.sh r06'
# Writes gpurun_out/evidence_<round>/: copy its files into profiles/ (same names) and add the prose headers by hand.
# Order matters: ALL FOUR counter passes first, their summaries copied into profiles/ ON THE BOX, then the driver-style bench line -- so that the line's
# roofline.traffic (headline AND the nested masked / small / fp8_large objects) comes from counter passes of the same sources on the same device.
: "${GRAFT_REPO_ROOT:?run through gpurun}"
RN=${1:-r06}
R=$GRAFT_REPO_ROOT; cd $R
T0=$(date +%s)
SHA=$(python3 -c "import bench; print(bench.kernel_source_hash())") || exit 1
E=$R/gpurun_out/evidence_$RN; rm -rf $E; mkdir -p $E
pmc() {   # tag, bench args
  local tag=$1; shift
  bash tools/pmc_bench.sh $tag "$@" > $E/pmc_$tag.log 2>&1
  cp gpurun_out/pmc_${tag}_$SHA/summary.json $E/${RN}_pmc_$tag.json && cp gpurun_out/pmc_${tag}_$SHA/kernel_stats.csv $E/${RN}_pmc_${tag}_kernel_stats.csv
  cp $E/${RN}_pmc_$tag.json $E/${RN}_pmc_${tag}_kernel_stats.csv profiles/
  rm -rf gpurun_out/pmc_${tag}_$SHA    # (the raw counter directories do not travel back: the summaries do)
}
pmc base_sup
pmc base_masked --objective masked
pmc small --config small
pmc large_fp8 --config large --patch 10 --dtype fp8 --batch 256
python bench.py > $E/${RN}_bench_line.json 2> $E/bench.err
L=$R/ecg-representation-learning_amd
python tools/gemm_ab.py --plain --nt4 --no-old 2>&1 | grep -v amdgpu.ids > $E/${RN}_gemm_shapes.txt
python tools/gemm_ab.py --m 64256 --dim 512 --plain --nt4 --no-old 2>&1 | grep -v amdgpu.ids > $E/${RN}_gemm_shapes_small.txt
python tools/gemm_ab.py --only ffn_ --aux8 --no-old --no-lib 2>&1 | grep -v amdgpu.ids > $E/${RN}_gemm_shapes_aux8.txt
# LayerNorm-fold pricing bodies (tools build) next to what ships
{ python tools/gemm_ab.py --only "fwd qkv" --rowaffine --no-old --no-lib --check; python tools/gemm_ab.py --only "fwd ffn_up" --rowaffine --aux8 --no-old --no-lib;
  python tools/gemm_ab.py --m 64256 --dim 512 --only "fwd qkv" --rowaffine --no-old --no-lib; python tools/gemm_ab.py --m 64256 --dim 512 --only "fwd ffn_up" --rowaffine --aux8 --no-old --no-lib; } 2>&1 | grep -v amdgpu.ids > $E/${RN}_ln_fold_pricing_raw.txt
# the attention kernels against round 5's library (same process, interleaved), base and large / patch-10 geometry; the forward's three forms in one build
python tools/attn_ab.py $L/csrc/build/libecgvit_hip_r05.so $L/libecgvit_hip.so 2>&1 | grep -v amdgpu.ids > $E/${RN}_attn_ab.txt
python tools/attn_ab.py $L/csrc/build/libecgvit_hip_r05.so $L/libecgvit_hip.so --n 501 --b 256 --h 16 2>&1 | grep -v amdgpu.ids >> $E/${RN}_attn_ab.txt
{ python tools/attn_fwd_ab.py; python tools/attn_fwd_ab.py --b 256 --h 16 --n 501; } 2>&1 | grep -v amdgpu.ids > $E/${RN}_attn_fwd_ab.txt
python tools/fp8_nt_ab.py 2>&1 | grep -v amdgpu.ids > $E/${RN}_fp8_nt_stores.txt
python tools/stress.py 150 2>&1 | grep -v amdgpu.ids > $E/${RN}_stress.txt
# whole-line A/B against the round-5 library, alternating on this device
bash tools/ab_r05.sh 3 2>&1 | grep -v amdgpu.ids > $E/${RN}_step_ab.txt
bash tools/steady_stats.sh base > /dev/null 2>&1; cp gpurun_out/steady_base.txt $E/${RN}_steady_base.txt
bash tools/steady_stats.sh small --config small > /dev/null 2>&1; cp gpurun_out/steady_small.txt $E/${RN}_steady_small.txt
bash tools/steady_stats.sh large_fp8 --config large --patch 10 --dtype fp8 --batch 256 > /dev/null 2>&1; cp gpurun_out/steady_large_fp8.txt $E/${RN}_steady_large_fp8.txt
rm -rf gpurun_out/steady_base gpurun_out/steady_small gpurun_out/steady_large_fp8
echo "evidence run: $(( ($(date +%s) - T0) / 60 )) minutes of box time" > $E/${RN}_evidence_minutes.txt
ls -la $E
python3 -c "
import json; o = json.load(open('$E/${RN}_bench_line.json'))
print(o['value'], o['ms_per_step'], o['roofline']['frac'], o['roofline']['traffic'], [(k, o.get(k, {}).get('value'), (o.get(k, {}).get('roofline') or {}).get('traffic')) for k in ('masked', 'small', 'fp8_large', 'bf16_saved_tensor')], o['kernel_source_sha16'])"
