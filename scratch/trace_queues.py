import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
print('span', (int(step[-1]['End_Timestamp']) - t0) / 1e3, 'launches', len(step))
qs = {}
for r in step:
    qs.setdefault(r['Queue_Id'], []).append(r)
for q, rs in qs.items():
    print('queue', q, 'kernels', len(rs), 'busy', sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rs))
prev = {}
for r in step:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    q = r['Queue_Id']
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:30]
    print(f"q{q} {s:8.1f} {e-s:7.1f}us gap {s-prev.get(q,0):6.1f} {nm:30s} grid {r['Grid_Size_X']:>8s},{r['Grid_Size_Y']},{r['Grid_Size_Z']}")
    prev[q] = e
