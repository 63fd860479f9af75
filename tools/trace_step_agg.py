"""Aggregate the kernels of ONE step (between the last two fps_kernel launches) of a rocprofv3 --kernel-trace csv:
name, calls, total us, share -- plus the step's span and the sum of the gaps between consecutive kernels."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'fps_kernel' in r['Kernel_Name'] or 'fps_query_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
agg = collections.OrderedDict()
prev_end, gaps, busy = None, 0, 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)[:100]
    c = agg.setdefault(name, [0, 0])
    c[0] += 1; c[1] += e - s
    if prev_end is not None and s > prev_end:
        gaps += s - prev_end
    prev_end = max(prev_end or 0, e)
    busy += e - s
span = int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])
print(f"step span {span/1e6:.3f} ms, kernel time {busy/1e6:.3f} ms, gaps {gaps/1e6:.3f} ms, launches {b-a}")
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t/1e3:10.1f} us {100*t/span:5.1f}%  x{n:5d}  avg {t/n/1e3:8.1f}  {name}")
