#!/bin/bash
# usage: tools/prof_bench.sh <tag> <workload> [bench args]   (run on the GPU box via gpurun)
# rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/prof_<tag>/ + step timeline gpurun_out/prof_<tag>.txt
tag=$1; wl=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/trace_step.py gpurun_out/prof_$tag > gpurun_out/prof_$tag.txt
