"""GPU timeline of ONE fit from a rocprofv3 --kernel-trace csv: busy time, idle gaps and what runs on either side of the big ones.
usage: python tools/gpu_gaps.py <kernel_trace.csv> [min_gap_us]        (the LAST fit of the trace is analysed: kernels after the last
gap longer than 5 ms)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
mingap = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]) for r in rows)
# split into fits at gaps > 5 ms
cuts = [0] + [i for i in range(1, len(ev)) if ev[i][0] - max(e[1] for e in ev[max(0, i - 8):i]) > 5e6] + [len(ev)]
a, b = cuts[-2], cuts[-1]
fit = ev[a:b]
t0, t1 = fit[0][0], max(e[1] for e in fit)
busy = 0; cur_s, cur_e = fit[0][0], fit[0][1]; gaps = []
last_name = fit[0][2]
for s, e, n in fit[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n, (cur_e - t0) / 1e6))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = n
busy += cur_e - cur_s
print('fit: %d kernels, span %.2f ms, GPU busy %.2f ms (union), idle %.2f ms' % (len(fit), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
big = [g for g in gaps if g[0] >= mingap * 1e3]
print('%d gaps >= %.0f us: total %.2f ms;  %d smaller gaps: total %.2f ms' % (len(big), mingap, sum(g[0] for g in big) / 1e6, len(gaps) - len(big), sum(g[0] for g in gaps if g[0] < mingap * 1e3) / 1e6))
for g in big:
    print('  at %7.2f ms: %7.1f us  after %-40s before %s' % (g[3], g[0] / 1e3, g[1][:40], g[2][:40]))
c = collections.Counter()
for s, e, n in fit: c[n] += e - s
for n, v in c.most_common(8): print('  %-60s %.2f ms' % (n, v / 1e6))
