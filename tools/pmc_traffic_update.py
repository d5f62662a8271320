"""HBM-side traffic of K-HEADS per launch from the two rocprofv3 --pmc passes of tools/gpu_pmc_traffic.sh.

    python tools/pmc_traffic_update.py gpurun_out/<tag>            (on the GPU box: writes <tag>/pmc_traffic_entry.json)
    python tools/pmc_traffic_update.py --merge gpurun_out/<tag>    (in the build container: folds it into profiles/pmc_traffic.json)

FETCH_SIZE is reported in KiB and counts HALF of the bytes of these access patterns on gfx950 (calibration kernels of known
size, profiles/r01_pmc/: doubled here, as MI355X_MICROARCH.md prescribes); WRITE_SIZE (KiB) is exact.  The entry carries the
fingerprint of the kernel sources it was measured on (bench.py::source_sha): bench.py attaches `roofline.traffic` only when
that fingerprint is the one of the tree being benched.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(root, counter):
    acc = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(root, 'heads_' + counter, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter or not re.search(r'heads_', r['Kernel_Name']):
                continue
            m = re.search(r'(heads_[a-z0-9_]+)', r['Kernel_Name'])
            acc[m.group(1)][r['Dispatch_Id']] += float(r['Counter_Value'])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}


def main():
    if sys.argv[1] == '--merge':
        d = sys.argv[2]
        ent = json.load(open(os.path.join(d, 'pmc_traffic_entry.json')))
        p = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        allj = json.load(open(p))
        allj['heads_fused'] = ent
        json.dump(allj, open(p, 'w'), indent=1)
        print('profiles/pmc_traffic.json: heads_fused <- %s (sources %s, %.3f GB per launch)' % (d, ent['source_sha'], ent['traffic_bytes'] / 1e9))
        return
    d = sys.argv[1]
    from bench import source_sha
    fe, wr = per_kernel(d, 'FETCH_SIZE'), per_kernel(d, 'WRITE_SIZE')
    fetch = 2.0 * 1024.0 * sum(fe.values())
    write = 1024.0 * sum(wr.values())
    B, G, hL = 4096, 20000, 64
    Gp, S, npart, NT = G, 2, 128, B // 32
    alg = {'byte counts': B * G, 'final dH partials written + re-read by the reduce': 2 * npart * NT * 32 * 64 * 4,
           'dW partials (2 batch splits) written + re-read': 2 * S * (hL + 2) * 3 * Gp * 4, 'head weights': (hL + 1) * 3 * Gp * 4,
           'split decoder output (both layouts, written + read)': 2 * 2 * NT * 3 * 32 * 64 * 2}
    ent = {'shape': {'B': B, 'G': G, 'hL': hL, 'ae_type': 'zinb-conddisp'}, 'source_sha': source_sha(),
           'raw': os.path.basename(os.path.normpath(d)) + ' (tools/gpu_pmc_traffic.sh; per-kernel KiB below)',
           'kernel': ' + '.join(sorted(set(fe) | set(wr))),
           'fetch_size_raw_kib': fe, 'write_size_raw_kib': wr, 'fetch_bytes_corrected': fetch, 'write_bytes': write,
           'traffic_bytes': fetch + write, 'algorithmic_hbm_bytes': float(sum(alg.values())), 'algorithmic_breakdown': alg,
           'note': 'FETCH_SIZE / WRITE_SIZE tally requests on the fabric side of the L2 and count Infinity-Cache hits as traffic: '
                   'the in-kernel read-modify-write of the dH partials (134 MB set, ~5 visits each way) is the bulk of the excess '
                   'over the algorithmic bytes'}
    json.dump(ent, open(os.path.join(d, 'pmc_traffic_entry.json'), 'w'), indent=1)
    print(json.dumps({k: ent[k] for k in ('source_sha', 'fetch_bytes_corrected', 'write_bytes', 'traffic_bytes', 'algorithmic_hbm_bytes')}))


if __name__ == '__main__':
    main()
