#!/usr/bin/env python
"""Print the kernel sequence of ONE training step from a rocprofv3 --kernel-trace CSV: the launches between two consecutive
optimizer launches (rmsprop_clip / adam / ...), with their durations and grids.  usage: step_trace.py <kernel_trace.csv> [nth]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r['Start_Timestamp']))
opt = [i for i, r in enumerate(rows) if 'rmsprop' in r['Kernel_Name'] or 'adam' in r['Kernel_Name'] or 'sgd' in r['Kernel_Name']]
# steps = the spans between optimizer launches; take the nth LONGEST-typical one (skip the batch-32 section: pick by span duration)
spans = []
for a, b in zip(opt[:-1], opt[1:]):
    if b - a < 3:
        continue
    t = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[a + 1:b + 1]) / 1e3
    spans.append((a + 1, b + 1, t))
if not spans:
    sys.exit('no steps found')
med = sorted(s[2] for s in spans)[len(spans) // 2]
big = [s for s in spans if s[2] > 0.5 * max(x[2] for x in spans)] if '--big' in sys.argv else spans
if '--median' in sys.argv:                      # a step of typical length (the batch-32 section of bench.py outnumbers the rest)
    big = [s for s in spans if abs(s[2] - med) < 0.02 * med]
if '--big' in sys.argv:                         # the typical full-batch step: the median of the big spans (not a span that holds a validation pass)
    bm = sorted(x[2] for x in big)[len(big) // 2]
    big = [x for x in big if abs(x[2] - bm) < 0.05 * bm] or big
a, b, t = big[min(nth, len(big) - 1)]
print('step span %.1f us of kernel time, %d launches (median span %.1f us, %d spans)' % (t, b - a, med, len(spans)))
tot = 0.0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print('%9.1f us  grid %-9s wg %-5s lds %-7s %s' % (d, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')),
                                                      r.get('LDS_Block_Size', '?'), name[:110]))
print('sum of kernel durations %.1f us' % tot)
