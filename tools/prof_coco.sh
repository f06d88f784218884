#!/bin/bash
# usage: tools/prof_coco.sh <tag> [batch] [steps] [warmup]   (run on the GPU box via gpurun; >= 30 steps: the host must be ahead of the GPU for the timeline to be the steady state)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload coco --batch ${2:-128} --steps ${3:-6} --warmup ${4:-2} --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$1.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/trace_step.py gpurun_out/prof_$1 > gpurun_out/prof_$1.txt
