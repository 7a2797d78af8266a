"""median / min / p90 / total device time per kernel (template arguments kept) from a rocprofv3 --kernel-trace csv
(`rocprofv3 --kernel-trace --output-format csv ...`)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def key(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '').replace('HIP_vector_type<float, 2u>', 'c32')
    return name.split('(')[0][:64]


agg = collections.OrderedDict()
for r in rows:
    agg.setdefault(key(r['Kernel_Name']), []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    v = sorted(v)
    print(f"{k:64s} n={len(v):6d} median {v[len(v)//2]/1e3:8.2f} us  min {v[0]/1e3:8.2f}  p90 {v[int(len(v)*.9)]/1e3:8.2f}  total {sum(v)/1e6:9.3f} ms")
