#!/bin/bash
# usage: tools/layer_pmc.sh <layer> [<layer> ...]   (run on the GPU box via gpurun)
# HBM-side traffic of ONE layer's kernels: two counter passes (FETCH_SIZE, WRITE_SIZE) over tools/layer_bench.py <layer>,
# which launches the layer's kernel(s) 23 times on the step's activations; summary -> gpurun_out/layer_pmc/<layer>.json
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/layer_pmc/$L
  mkdir -p $out
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/tools/layer_bench.py $L > $out/$c.log 2>&1 || { echo "$L $c failed"; tail -5 $out/$c.log; exit 1; }
  done
  (cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py --layer gpurun_out/layer_pmc/$L $L 256 100)
done
