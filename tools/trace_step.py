import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
tot = 0
agg = {}
qsum = {}
for r in step:
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += dur
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:30]
    agg[nm] = agg.get(nm, 0) + dur
    qid = r.get('Queue_Id', '?')
    qsum[qid] = qsum.get(qid, 0) + dur
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {dur:8.1f}us q{qid} {nm:30s} grid {r['Grid_Size_X']:>8s},{r['Grid_Size_Y']},{r['Grid_Size_Z']}")
print('kernel us per queue', {k: round(v, 1) for k, v in qsum.items()})
print('sum kernel us', tot, 'span', (int(step[-1]['End_Timestamp']) - t0) / 1e3, 'launches', len(step))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"  {v:8.1f}us {k}")
