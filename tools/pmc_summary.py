"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel (mean per launch) and derives the
utilisation figures DESIGN.md quotes.

  python tools/pmc_summary.py <dir with p*/..counter_collection.csv> [out.csv]

MfmaUtil  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs)        (gfx94x formula; busy cycles
            = 64 per v_mfma_f32_32x32x2_f32 -- the `from_insts` column assumes that instruction; the 16-bit
            v_mfma_f32_32x32x16_{bf16,f16} of rounds 2-6 occupy the pipe 32 cycles each: halve that column for them)
VALUBusy  = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024)                       (quad-cycle units)
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

KEEP = re.compile(r'heads_fused|heads_reduce|gemm_|transpose|splitk_reduce|zinb_nll|bn_|col_moments|rmsprop|moments_combine|enc0_|stack_|'
                  r'loss_finalize|step_end|relu_|dropout|optimizer|prelu|elempi')


def short(name):
    m = re.search(r'(?:anonymous namespace\)::)?([A-Za-z_0-9]+)(<[^>]*>)?\(', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
        per_dispatch = defaultdict(float)
        meta, wall = {}, {}
        for r in csv.DictReader(open(f)):
            if not KEEP.search(r['Kernel_Name']):
                continue
            key = (f, r['Dispatch_Id'], r['Counter_Name'])
            per_dispatch[key] += float(r['Counter_Value'])
            wall[(f, r['Dispatch_Id'])] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
            meta[(f, r['Dispatch_Id'])] = short(r['Kernel_Name'])
        for (ff, d, c), v in per_dispatch.items():
            acc[meta[(ff, d)]][c].append(v)
        for k, v in wall.items():
            acc[meta[k]]['wall_ns'].append(v)
    counters = sorted({c for k in acc.values() for c in k})
    rows = []
    for k, cs in sorted(acc.items()):
        row = {'kernel': k, 'launches': max(len(v) for v in cs.values())}
        for c in counters:
            row[c] = sum(cs[c]) / len(cs[c]) if c in cs else ''
        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (check: clock_GHz below = GRBM / 8 / wall
        # comes out at the 1.9-2.4 GHz the chip runs at): elapsed cycles = GRBM / 8, SIMD-cycles = x 1024
        g = (row.get('GRBM_GUI_ACTIVE') or 0) / 8.0
        if g:
            row['clock_GHz'] = g / row['wall_ns'] if row.get('wall_ns') else ''
            if row.get('SQ_VALU_MFMA_BUSY_CYCLES') != '':
                row['MfmaUtil_%'] = 100.0 * row['SQ_VALU_MFMA_BUSY_CYCLES'] / (g * 1024)
            if row.get('SQ_INSTS_MFMA') != '':
                row['MfmaUtil_from_insts_%'] = 100.0 * row['SQ_INSTS_MFMA'] * 64 / (g * 1024)
            if row.get('SQ_ACTIVE_INST_VALU') != '':
                row['VALUBusy_%'] = 100.0 * row['SQ_ACTIVE_INST_VALU'] * 4 / (g * 1024)
        rows.append(row)
    counters = [c for c in counters if c != 'wall_ns']
    cols = ['kernel', 'launches', 'wall_ns', 'clock_GHz', 'MfmaUtil_%', 'MfmaUtil_from_insts_%', 'VALUBusy_%'] + counters
    out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
    w = csv.DictWriter(out, fieldnames=cols, extrasaction='ignore')
    w.writeheader()
    for r in rows:
        w.writerow({c: (('%.4g' % r[c]) if isinstance(r.get(c), float) else r.get(c, '')) for c in cols})
    if len(sys.argv) > 3:
        # per-pipe instruction counts of the dominant kernels, stamped with the kernel sources' fingerprint: bench.py turns
        # them into the vector-issue / matrix-pipe floors of its `roofline.step` block (only when measured on ITS sources)
        import json
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        pick = {}
        for r in rows:
            for key, pat in (('heads_fused', 'heads_fused_h2_kernel'), ('gemm_heads_fwd', 'gemm_h2w_kernel')):
                if pat in r['kernel'] and r.get('SQ_INSTS_VALU') not in ('', None) and \
                        (key not in pick or r['wall_ns'] > pick[key]['wall_ns']):
                    pick[key] = {'kernel': r['kernel'], 'wall_ns': r['wall_ns'], 'clock_GHz': r.get('clock_GHz'),
                                 'SQ_INSTS_VALU': r['SQ_INSTS_VALU'], 'SQ_INSTS_MFMA': r.get('SQ_INSTS_MFMA'),
                                 'SQ_INSTS_LDS': r.get('SQ_INSTS_LDS'), 'MfmaUtil_pct': r.get('MfmaUtil_%'),
                                 'VALUBusy_pct': r.get('VALUBusy_%'), 'launches_averaged': r['launches']}
        json.dump({'source_sha': bench.source_sha(), 'bench_args': os.environ.get('BENCH_ARGS', ''),
                   'note': 'mean per launch over the launches of tools/gpu_pmc_bench.sh (full and partial batches of the bench '
                           'sequence); wall_ns is the profiled duration', 'kernels': pick}, open(sys.argv[3], 'w'), indent=1)


if __name__ == '__main__':
    main()
