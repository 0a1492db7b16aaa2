"""Summarises rocprofv3 csv output of tools/prof.sh: per-kernel durations and PMC counters (per dispatch averages)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def short(n):
    n = n.replace('void ', '')
    return n[:110]
for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    print('== kernel stats', f)
    for row in list(csv.DictReader(open(f)))[:8]:
        print('  %-110s calls=%s avg_ns=%s total_ns=%s pct=%s' % (short(row['Name']), row['Calls'], row['AverageNs'], row['TotalDurationNs'], row['Percentage']))
for sub in sorted(glob.glob(os.path.join(out, 'pmc*'))):
    if not os.path.isdir(sub): continue
    for f in glob.glob(os.path.join(sub, '**', '*counter_collection.csv'), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        print('== counters', f)
        for k, cs in acc.items():
            if 'step_kernel' not in k: continue
            print('  ', short(k))
            for c, v in cs.items():
                print('      %-24s n=%d mean=%.6g' % (c, len(v), sum(v) / len(v)))
