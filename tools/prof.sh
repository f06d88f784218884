#!/bin/bash
# usage: tools/prof.sh <tag>   (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probe > $GRAFT_REPO_ROOT/gpurun_out/prof_$1.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/trace_step.py gpurun_out/prof_$1 > gpurun_out/prof_$1.txt
