#!/usr/bin/env python3
"""Builds profiles/r04_traffic.json (what bench.py's measured_traffic reads) from the PMC passes of tools/r04_collect.sh:
  * gpurun_out/layer_pmc/<layer>/traffic.json  (tools/layer_pmc.sh: one layer's kernels replayed alone)  -> "<layer>:256:100"
  * gpurun_out/pmc_r4ca/traffic.json           (tools/pmc.sh over bench.py --workload celeba)            -> "ca_<family>:512:100":
    HBM-side bytes of EVERY launch of a kernel family in one step = sum over the family's kernels of bytes per launch x launches,
    divided by the number of steps of the profiled run (= launches of adam_kernel).
FETCH_SIZE x2 (gfx950: MI355X_MICROARCH.md, HBM) + WRITE_SIZE, separate --pmc passes, KiB -> bytes."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for layer in ("dec_convT3", "dec_convT3_dgrad", "dec_convT3_wgrad", "dec_convT2_wgrad", "enc_conv2_wgrad", "enc_conv3_wgrad"):
    p = os.path.join(root, "gpurun_out", "layer_pmc", layer, "traffic.json")
    if os.path.exists(p):
        out["%s:256:100" % layer] = json.load(open(p))
p = os.path.join(root, "gpurun_out", "pmc_r4ca", "traffic.json")
if os.path.exists(p):
    t = json.load(open(p))["kernels"]
    steps = max(v["launches"] for k, v in t.items() if k.startswith("adam_kernel"))
    fams = {"ca_wgrad": lambda k: k.startswith("wgrad_"), "ca_gemm_gather": lambda k: k.startswith("gemm_gather_kernel"),
            "ca_convres": lambda k: k.startswith("convres_kernel")}
    for name, sel in fams.items():
        parts, total = [], 0.0
        for k, v in sorted(t.items()):
            if sel(k):
                b = v["hbm_bytes_per_launch"] * v["launches"] / steps
                total += b
                parts.append({"kernel": k[:100], "launches_per_step": v["launches"] / steps, "read_bytes_per_launch": v["read_bytes_per_launch"],
                              "write_bytes_per_launch": v["write_bytes_per_launch"], "bytes_per_step": b})
        out["%s:512:100" % name] = {"bytes": total, "parts": parts,
                                    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --workload celeba "
                                              "(tools/pmc.sh); every launch of the family in one step (%d profiled steps); FETCH_SIZE x2 "
                                              "(gfx950), KiB -> bytes" % steps}
json.dump(out, open(os.path.join(root, "gpurun_out", "r04", "traffic.json"), "w"), indent=1)
for k, v in out.items():
    print("%-28s %10.1f MB" % (k, v["bytes"] / 1e6))
