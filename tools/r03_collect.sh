#!/bin/bash
# usage: tools/r03_collect.sh   (run on the GPU box via gpurun): every number DESIGN.md section 6 quotes for round 3
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
python bench.py > $O/bench_multimnist.json 2> $O/bench_multimnist.err || { tail -5 $O/bench_multimnist.err; exit 1; }
python bench.py --loader --no-cpu-baseline --no-probe --steps 500 > $O/bench_multimnist_loader.json 2>/dev/null || exit 1
python bench.py --workload celeba --no-cpu-baseline > $O/bench_celeba.json 2>/dev/null || exit 1
python bench.py --workload coco --loader --no-cpu-baseline > $O/bench_coco.json 2>/dev/null || exit 1
python bench.py --workload coco --batch 1024 --steps 40 --warmup 5 --no-cpu-baseline --no-probe > $O/bench_coco_b1024.json 2>/dev/null || exit 1
echo benches done
python tools/layer_bench.py > $O/layer_bench_b256.txt 2>&1 || exit 1
python tools/step_parts.py > $O/step_parts.txt 2>&1 || exit 1
echo layer bench done
bash tools/prof.sh r3mm && cp gpurun_out/prof_r3mm/*/*kernel_stats.csv $O/multimnist_kernel_stats.csv && cp gpurun_out/prof_r3mm.txt $O/multimnist_step_timeline.txt || exit 1
bash tools/prof_bench.sh r3ca celeba --steps 20 --warmup 3 --no-probe && cp gpurun_out/prof_r3ca/*/*kernel_stats.csv $O/celeba_kernel_stats.csv && cp gpurun_out/prof_r3ca.txt $O/celeba_step_timeline.txt || exit 1
bash tools/prof_bench.sh r3co coco --steps 12 --warmup 3 --no-probe && cp gpurun_out/prof_r3co/*/*kernel_stats.csv $O/coco_b128_kernel_stats.csv && cp gpurun_out/prof_r3co.txt $O/coco_b128_step_timeline.txt || exit 1
echo profiles done
bash tools/pmc.sh r3 multimnist --no-probe > $O/pmc.log 2>&1; cp gpurun_out/pmc_r3.txt $O/multimnist_pmc_summary.txt 2>/dev/null; cp gpurun_out/pmc_r3/sq_counters.csv $O/multimnist_sq_counters.csv 2>/dev/null; cp gpurun_out/pmc_r3/traffic.json $O/multimnist_pmc_traffic_by_kernel.json 2>/dev/null
echo all done
