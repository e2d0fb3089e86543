"""Summarise a rocprofv3 --pmc run stored as rocpd (SQLite): per kernel (name filter) median counter values per
counter row and the median duration.  python tools/pmc_db.py <results.db> [name-substring]"""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else ''
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
ev = [t for t in tabs if 'pmc_event' in t][0]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
pm = [t for t in tabs if 'info_pmc' in t][0]
ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
q = ("select s.kernel_name, p.name, e.value, k.end - k.start, k.id from %s e join %s k on e.event_id = k.event_id "
     "join %s p on e.pmc_id = p.id join %s s on k.kernel_id = s.id" % (ev, kd, pm, ks))
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
dur = collections.defaultdict(dict)
for name, c, v, dt, kid in cur.execute(q):
    if filt in name:
        short = name.split('(')[0][-70:]
        acc[short][c][kid] += v          # summed over the counter's instances (XCDs / SEs) of one dispatch
        dur[short][kid] = dt
for k, cs in acc.items():
    d = sorted(dur[k].values())
    print(k, 'dispatches=%d' % len(d), 'median_us=%.1f' % (d[len(d) // 2] / 1e3))
    for c, per in cs.items():
        v = sorted(per.values())
        print('   %-28s %.5g' % (c, v[len(v) // 2]))
