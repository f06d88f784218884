#!/bin/bash
# usage: tools/prof_knobs.sh <tag> [--knob name=value ...]   (run on the GPU box via gpurun): kernel trace of bench.py under library knobs
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probe "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/trace_step.py gpurun_out/prof_$tag > gpurun_out/prof_$tag.txt
