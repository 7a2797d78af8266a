"""median / min / p90 device time per kernel (template arguments kept) from a rocprofv3 --kernel-trace csv"""
import csv, collections, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r['Start_Timestamp'])):
    n = r['Kernel_Name']
    m = re.match(r'(?:void )?([A-Za-z_0-9]+)(?:<(.*?)>\()?', n)
    key = m.group(1) + ('<' + m.group(2).replace('HIP_vector_type<float, 2u>', 'c32') + '>' if m.group(2) else '')
    agg.setdefault(key, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    v = sorted(v)
    print(f"{k:60s} n={len(v):6d} median {v[len(v)//2]/1e3:8.2f} us  min {v[0]/1e3:8.2f}  p90 {v[int(len(v)*.9)]/1e3:8.2f}")
