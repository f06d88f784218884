import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
for r in step:
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:30]
    s = (int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    print(f"{s:8.1f} {e:8.1f} s{r['Stream_Id']} {nm} grid {r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}")
