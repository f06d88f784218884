#!/bin/bash
# usage: tools/prof_loader.sh <tag>   (run on the GPU box via gpurun): kernel + memory-copy trace of the loader-fed MultiMNIST loop
tag=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --loader --steps 40 --warmup 10 --no-cpu-baseline --no-probe > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
