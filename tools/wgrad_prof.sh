#!/bin/bash
# usage: tools/wgrad_prof.sh <tag> [wgrad_check.py arguments]   (run on the GPU box via gpurun): per-kernel durations of the weight-gradient
# kernels (ring kernel, its reduce, the streamed kernel and its reduce) out of rocprofv3 --kernel-trace --stats
T=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$T -- python3 $GRAFT_REPO_ROOT/tools/wgrad_check.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$T.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$T" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "wgrad" in n:
        short = n.split("(")[0][-110:]
        print(f'{short:110s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}  max {float(r["MaxNs"])/1e3:8.2f}')
PY
