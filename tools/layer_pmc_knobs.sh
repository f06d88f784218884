#!/bin/bash
# usage: tools/layer_pmc_knobs.sh <tag> <layer> [knob=value ...]   (run on the GPU box via gpurun)
# HBM-side traffic of ONE layer's kernels under library knobs: as tools/layer_pmc.sh, summary -> gpurun_out/layer_pmc/<tag>.json
tag=$1; L=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/layer_pmc/$tag
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/tools/layer_bench.py $L "$@" > $out/$c.log 2>&1 || { echo "$L $c failed"; tail -5 $out/$c.log; exit 1; }
done
(cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py --layer gpurun_out/layer_pmc/$tag $L 256 100)
