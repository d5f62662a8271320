#!/usr/bin/env python
"""Kernel-stat summary (name, calls, total/avg/min/max ns, %) from a rocprofv3 rocpd .db
(ROCm 7.2 writes sqlite by default); the CSV it prints is what we commit under profiles/."""
import csv
import sqlite3
import sys


def main(db, out=None, top=40):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by %s order by 3 desc" % (name, name)).fetchall()
    tot = float(sum(r[2] for r in rows)) or 1.0
    f = open(out, 'w', newline='') if out else sys.stdout
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows[:top]:
        w.writerow([r[0][:160], r[1], r[2], '%.1f' % r[3], '%.3f' % (100.0 * r[2] / tot), r[4], r[5]])
    if out:
        f.close()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
