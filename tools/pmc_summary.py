"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (name filter), mean counter values and duration.
python tools/pmc_summary.py <csv> [name-substring]"""
import csv, sys, collections
path = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ''
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(path)):
    name = r['Kernel_Name']
    if filt not in name:
        continue
    short = name.split('(')[0][-60:]
    acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[short][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, cs in acc.items():
    d = sorted(dur[k].values())
    print(k, 'n=%d' % len(d), 'median_us=%.1f' % d[len(d) // 2])
    for c, v in cs.items():
        v = sorted(v)
        print('   %-32s median %.4g' % (c, v[len(v) // 2]))
