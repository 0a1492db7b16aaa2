"""Per-kernel means of every PMC counter collected by tools/prof.sh -> JSON (profiles/r01_<tag>_pmc.json).

usage: python tools/pmc_json.py gpurun_out/prof_c4 > profiles/r01_c4_pmc.json
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; hbm_* fields apply the gfx950 correction (FETCH_SIZE x 2)."""
import csv, glob, json, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
res = {}
for k, cs in sorted(acc.items()):
    d = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    d['dispatches'] = max(len(v) for v in cs.values())
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        d['hbm_read_bytes_corrected'] = d['FETCH_SIZE'] * 1024 * 2
        d['hbm_write_bytes'] = d['WRITE_SIZE'] * 1024
        d['hbm_bytes_per_launch'] = d['hbm_read_bytes_corrected'] + d['hbm_write_bytes']
    res[k] = d
json.dump(res, sys.stdout, indent=1)
