"""Print the kernel timeline of one bench step from a rocprofv3 --kernel-trace csv."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'fps_kernel' in r['Kernel_Name'] or 'fps_query' in r['Kernel_Name']]
# a training step = a sampling launch followed (before the next one) by the optimiser's kernel; the passes after the
# timed region (bench.py's untimed row statistics, the CPU-baseline leg's device work) have no optimiser launch
steps = [(i, j) for i, j in zip(idx[:-1], idx[1:]) if any('adamw' in r['Kernel_Name'] for r in rows[i:j])]
a, b = steps[-2] if len(steps) >= 2 else (idx[-3], idx[-2])
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'\(.*', '', name)[:56]
    gap = (s - prev_end) / 1e3 if prev_end else 0
    wg = int(r['Workgroup_Size_X'])
    print(f"{(s-t0)/1e3:9.1f} dur {(e-s)/1e3:7.1f} gap {gap:5.1f} wgs {int(r['Grid_Size_X'])*int(r['Grid_Size_Y'])*int(r['Grid_Size_Z'])//max(wg*int(r['Workgroup_Size_Y']),1):>6} lds {r['LDS_Block_Size']:>6} vgpr {r['VGPR_Count']:>3} {name}")
    prev_end = e
    busy += e - s
print('step span us', (int(rows[b]['Start_Timestamp']) - t0) / 1e3, 'busy', busy / 1e3, 'n', b - a)
