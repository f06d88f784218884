"""Short summary of bench.py JSON lines: python tools/printbench.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    wl = d.get("with_loader") or {}
    rf = d.get("roofline") or {}
    print(f"{f}: {d['config']['workload'][:40]}  {d['ms_per_step']:.4f} ms/step  {d['value']:.1f} steps/s"
          + (f"  loader {wl['ms_per_step']:.4f} ms ({wl['vs_resident']:.3f} of resident)" if wl else "")
          + (f"  step_mfma_frac {d['step_mfma_frac']:.4f}" if "step_mfma_frac" in d else "")
          + (f"  x{d['speedup_vs_cpu_baseline']:.0f} CPU" if "speedup_vs_cpu_baseline" in d else ""))
    if rf:
        print(f"    roofline: {rf['kernel'][:90]}  frac {rf['frac']:.4f}  us/step {rf.get('us_per_step', 0):.1f}")
