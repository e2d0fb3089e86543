"""Timeline of one bench step from a rocprofv3 kernel trace: python tools/step_timeline.py <kernel_trace.csv> [step_index]
Prints start offset, duration, queue and (short) name of every kernel between two consecutive front-end launches."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'frontend_kernel' in r['Kernel_Name']]
a, b = starts[which], starts[which + 1]
t0 = int(rows[a]['Start_Timestamp'])
busy_until = {}
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    q = r.get('Queue_Id', '?')
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    gap = s - busy_until.get(q, 0)
    busy_until[q] = e
    print("%9.1f us  +%8.1f us  (gap on queue %6.1f)  q%-3s %s" % (s / 1e3, (e - s) / 1e3, gap / 1e3, q, name))
print("step: %.1f us" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
