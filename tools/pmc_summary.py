#!/usr/bin/env python3
"""Summarise the three rocprofv3 counter passes of tools/pmc.sh.

usage: pmc_summary.py <dir with fetch/ write/ sq/> <workload>
Prints a per-kernel table and writes <dir>/traffic.json + <dir>/sq_counters.csv:
  * HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM)
    + WRITE_SIZE, both reported by rocprofv3 in KiB;
  * SQ counters summed over the launches of each kernel, divided by the launch count.
"""
import csv, glob, json, os, sys
from collections import defaultdict


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def read(d):
    """-> {kernel: {counter: [values per dispatch]}}, {kernel: [durations ns]}"""
    vals, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        names = {}
        for r in csv.DictReader(open(f)):
            k = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
            if "Start_Timestamp" in r and r["Start_Timestamp"]:
                names[(r["Dispatch_Id"], "t")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for did, cs in per_dispatch.items():
            for c, v in cs.items():
                vals[names[did]][c].append(v)
            if (did, "t") in names:
                dur[names[did]].append(names[(did, "t")])
    return vals, dur


def read_by_grid(d, counter):
    """-> {(kernel, grid): [value per dispatch, in dispatch order]}"""
    out = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
        per = defaultdict(float)
        meta = {}
        for r in rows:
            per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            meta[int(r["Dispatch_Id"])] = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        for did in sorted(per):
            out[meta[did]].append(per[did])
    return out


def layer_mode(root, layer, B, D):
    """tools/layer_pmc.sh: the layer's kernels are the (kernel, grid) groups launched >= 23 times; mean of the last 20."""
    fetch = read_by_grid(os.path.join(root, "FETCH_SIZE"), "FETCH_SIZE")
    write = read_by_grid(os.path.join(root, "WRITE_SIZE"), "WRITE_SIZE")
    total, parts = 0.0, []
    for key in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(key, []), write.get(key, [])
        if max(len(fv), len(wv)) < 23:
            continue
        per_iter = max(len(fv), len(wv)) // 23          # launches of this kernel per layer iteration
        rd = 2.0 * 1024.0 * sum(fv[-20 * per_iter:]) / 20.0
        wr = 1024.0 * sum(wv[-20 * per_iter:]) / 20.0
        total += rd + wr
        parts.append({"kernel": key[0], "grid": key[1], "launches_per_iteration": per_iter, "read_bytes": rd, "write_bytes": wr})
    res = {"bytes": total, "parts": parts,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/layer_bench.py %s; FETCH_SIZE x2 (gfx950), "
                     "KiB -> bytes, mean of 20 launches" % layer}
    json.dump(res, open(os.path.join(root, "traffic.json"), "w"), indent=1)
    print("%s:%d:%d" % (layer, B, D), json.dumps(res))


def main():
    if sys.argv[1] == "--layer":
        return layer_mode(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
    root, wl = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "multimnist"
    fetch, _ = read(os.path.join(root, "fetch"))
    write, _ = read(os.path.join(root, "write"))
    sq, dur = read(os.path.join(root, "sq"))
    kernels = sorted(set(fetch) | set(write) | set(sq))
    traffic = {}
    rows = []
    for k in kernels:
        fv = fetch.get(k, {}).get("FETCH_SIZE", [])
        wv = write.get(k, {}).get("WRITE_SIZE", [])
        n = max(len(fv), len(wv), 1)
        rd = 2.0 * 1024.0 * sum(fv) / max(len(fv), 1)       # bytes per launch (x2: gfx950 correction)
        wr = 1024.0 * sum(wv) / max(len(wv), 1)
        traffic[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
        s = sq.get(k, {})
        m = lambda c: sum(s.get(c, [0.0])) / max(len(s.get(c, [])), 1)
        rows.append((k, n, rd, wr, m("SQ_WAVES"), m("SQ_BUSY_CYCLES"), m("SQ_WAVE_CYCLES"), m("SQ_INSTS_VALU"), m("SQ_INSTS_MFMA"),
                     m("SQ_INSTS_LDS"), m("SQ_VALU_MFMA_BUSY_CYCLES"), m("SQ_WAIT_ANY")))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    json.dump({"workload": wl, "unit": "bytes", "note": "FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KiB -> bytes, mean per launch; separate --pmc passes",
               "kernels": traffic}, open(os.path.join(root, "traffic.json"), "w"), indent=1, sort_keys=True)
    hdr = ["kernel", "launches", "read_B", "write_B", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA",
           "SQ_INSTS_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY"]
    with open(os.path.join(root, "sq_counters.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(hdr)
        for r in rows:
            w.writerow([r[0], r[1]] + ["%.0f" % x for x in r[2:]])
    print("%-52s %6s %12s %12s %10s %12s %12s %10s" % ("kernel", "n", "read MB", "write MB", "waves", "insts_valu", "insts_mfma", "mfma_busy"))
    for r in rows:
        print("%-52s %6d %12.3f %12.3f %10.0f %12.0f %12.0f %10.0f" % (r[0][:52], r[1], r[2] / 1e6, r[3] / 1e6, r[4], r[7], r[8], r[10]))


if __name__ == "__main__":
    main()
