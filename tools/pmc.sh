#!/bin/bash
# usage: tools/pmc.sh <tag> [workload] [extra bench args]   (run on the GPU box via gpurun)
# Three separate counter passes over the same bench command (rocprofv3 cannot fit FETCH_SIZE and WRITE_SIZE in one pass):
#   1. FETCH_SIZE   2. WRITE_SIZE   3. SQ instruction / busy counters
# Counters only go with --kernel-trace (no hip/hsa/sys trace domains in a counter run).
tag=$1; wl=${2:-multimnist}; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out.$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out.$name.log; exit 1; }
  echo "pass $name done"
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py gpurun_out/pmc_$tag $wl > gpurun_out/pmc_$tag.txt && tail -30 gpurun_out/pmc_$tag.txt
