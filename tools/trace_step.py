"""Per-kernel durations and gaps of the batch-32 training step from a rocprofv3 kernel trace of bench.py:
  rocprofv3 --kernel-trace -d out -o bench --output-format csv -- python bench.py ...
  python tools/trace_step.py out/.../bench_kernel_trace.csv [marker-kernel-substring] [max kernels per step]
One step = the kernels between two consecutive launches of the marker kernel (default: the small-batch K-HEADS); intervals with
more kernels than the limit (default 16) are not steps (epoch ends, validation) and are skipped."""
import csv, collections, sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else 'heads_fused_small'
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
dur = collections.OrderedDict(); gaps = []; steps = []
skip = 10 if len(idx) > 40 else 2
for a, b in zip(idx[skip:-skip], idx[skip + 1:-skip + 1]):
    if b - a > limit:
        continue
    pe = None
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        dur.setdefault(r['Kernel_Name'][:70], []).append((e - s) / 1e3)
        if pe:
            gaps.append((s - pe) / 1e3)
        pe = e
    steps.append((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3)
tot = 0.0
for k, v in dur.items():
    per_step = sum(v) / len(steps)
    tot += per_step
    print('%-72s x%.0f  mean %6.1f us   per step %6.1f us' % (k, len(v) / len(steps), sum(v) / len(v), per_step))
steps.sort()
print('kernels per step %.1f us; step (start to start) median %.1f us over %d steps; mean gap inside a step %.2f us'
      % (tot, steps[len(steps) // 2], len(steps), sum(gaps) / max(len(gaps), 1)))
