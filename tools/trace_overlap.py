"""How the two streams of the pipelined step share the chip: per kernel name, the mean duration in the pipelined run
against the un-pipelined one (rocprofv3 --kernel-trace csv of each), over the steady-state steps, and the busy time
of each queue per step.      python tools/trace_overlap.py PIPELINED.csv PLAIN.csv"""
import csv, re, sys
from collections import defaultdict


def load(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ad = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
    # steady state: the last 20 optimiser launches that are ~one step apart
    ad = ad[-22:-1]
    return rows, ad


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return re.sub(r'\(.*', '', n)[:48]


def stats(rows, ad):
    per = defaultdict(list)
    a, b = ad[0], ad[-1]
    for r in rows[a + 1:b + 1]:
        per[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    steps = len(ad) - 1
    span = (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3 / steps
    q = defaultdict(float)
    for r in rows[a + 1:b + 1]:
        q[r.get('Queue_Id', '?')] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 / steps
    return per, steps, span, q


(rp, ap), (rn, an) = load(sys.argv[1]), load(sys.argv[2])
pp, sp, span_p, qp = stats(rp, ap)
pn, sn, span_n, qn = stats(rn, an)
print(f"step: pipelined {span_p:.1f} us, plain {span_n:.1f} us;  busy per queue and step: pipelined {dict((k, round(v, 1)) for k, v in qp.items())}  plain {dict((k, round(v, 1)) for k, v in qn.items())}")
print(f"{'kernel':48s} {'n/step':>6s} {'plain us':>9s} {'piped us':>9s} {'stretch':>8s} {'extra us/step':>13s}")
tot = 0.0
for k in sorted(pn, key=lambda k: -sum(pn[k])):
    if k not in pp:
        continue
    mn, mp = sum(pn[k]) / len(pn[k]), sum(pp[k]) / len(pp[k])
    extra = (mp - mn) * len(pp[k]) / sp
    tot += extra
    print(f"{k:48s} {len(pp[k]) / sp:6.1f} {mn:9.1f} {mp:9.1f} {mp / mn:8.2f} {extra:13.1f}")
print(f"sum of stretches per step: {tot:.1f} us")
