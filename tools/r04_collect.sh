#!/bin/bash
# usage: tools/r04_collect.sh [part ...]   (run on the GPU box via gpurun): every number DESIGN.md section 6 quotes for round 4.
# parts: bench layers prof pmc  (default: all)
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
parts=${@:-bench layers prof pmc}
for part in $parts; do case $part in
bench)
  python bench.py > $O/bench_multimnist.json 2> $O/bench_multimnist.err || { tail -5 $O/bench_multimnist.err; exit 1; }
  python bench.py --loader --no-cpu-baseline --no-probe --steps 500 > $O/bench_multimnist_loader.json 2>/dev/null || exit 1
  python bench.py --workload celeba --no-cpu-baseline > $O/bench_celeba.json 2>/dev/null || exit 1
  python bench.py --workload coco --loader --no-cpu-baseline > $O/bench_coco.json 2>/dev/null || exit 1
  python bench.py --workload coco --batch 1024 --steps 40 --warmup 5 --no-cpu-baseline --no-probe > $O/bench_coco_b1024.json 2>/dev/null || exit 1
  python bench.py --graph --no-cpu-baseline --no-probe --steps 500 > $O/bench_multimnist_graph.json 2>/dev/null || exit 1
  python tools/dp_overhead.py 300 packed dp 2>&1 | grep "ms/step" > $O/dp_world1.txt || exit 1
  python tools/host_time.py 2>&1 | grep -v amdgpu > $O/host_time.txt
  echo benches done;;
layers)
  python tools/layer_bench.py > $O/layer_bench_b256.txt 2>&1 || exit 1
  python tools/step_parts.py > $O/step_parts.txt 2>&1 || exit 1
  python tools/step_ab.py full:dbg_skip_wgrad=0 none:dbg_skip_wgrad=1 noring:dbg_skip_wgrad=2 nostreamed:dbg_skip_wgrad=3 nogrouped:dbg_skip_wgrad=4 notext:dbg_skip_text=1 rounds=3 > $O/step_knockouts.txt 2>&1 || exit 1
  WR_TS=1 python tools/wgrad_check.py 256 dec_convT3_wgrad dec_convT2_wgrad enc_conv2_wgrad enc_conv3_wgrad dec_convT1_wgrad enc_conv4_wgrad 2>&1 | grep -v amdgpu > $O/wgrad_ring_check.txt || exit 1
  echo layer bench done;;
prof)
  bash tools/prof.sh r4mm && cp gpurun_out/prof_r4mm/*/*kernel_stats.csv $O/multimnist_kernel_stats.csv && cp gpurun_out/prof_r4mm.txt $O/multimnist_step_timeline.txt || exit 1
  bash tools/prof_bench.sh r4ca celeba --steps 20 --warmup 3 --no-probe && cp gpurun_out/prof_r4ca/*/*kernel_stats.csv $O/celeba_kernel_stats.csv && cp gpurun_out/prof_r4ca.txt $O/celeba_step_timeline.txt || exit 1
  bash tools/prof_bench.sh r4co coco --steps 12 --warmup 3 --no-probe && cp gpurun_out/prof_r4co/*/*kernel_stats.csv $O/coco_b128_kernel_stats.csv && cp gpurun_out/prof_r4co.txt $O/coco_b128_step_timeline.txt || exit 1
  echo profiles done;;
pmc)
  bash tools/pmc.sh r4 multimnist --no-probe > $O/pmc.log 2>&1; cp gpurun_out/pmc_r4.txt $O/multimnist_pmc_summary.txt 2>/dev/null; cp gpurun_out/pmc_r4/sq_counters.csv $O/multimnist_sq_counters.csv 2>/dev/null; cp gpurun_out/pmc_r4/traffic.json $O/multimnist_pmc_traffic_by_kernel.json 2>/dev/null
  bash tools/pmc.sh r4ca celeba --no-probe > $O/pmc_ca.log 2>&1; cp gpurun_out/pmc_r4ca.txt $O/celeba_pmc_summary.txt 2>/dev/null
  bash tools/layer_pmc.sh dec_convT3 dec_convT3_dgrad dec_convT3_wgrad dec_convT2_wgrad enc_conv2_wgrad enc_conv3_wgrad > $O/layer_pmc.log 2>&1
  python3 tools/r04_traffic.py > $O/traffic_summary.txt 2>&1; cat $O/traffic_summary.txt
  echo pmc done;;
esac; done
echo all done
