"""Per-kernel averages of one rocprofv3 --pmc pass of SQ counters over tools/pmc_probe.py (where the waves' cycles go):
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
usage: python tools/pmc_sq_summarize.py <rocprof output dir> profiles/r03_pmc_sq.csv"""
import collections, csv, glob, sys

d, out = sys.argv[1:3]
f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip().replace(",", ";")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out, "w") as fo:
    fo.write("kernel,launches," + ",".join(names) + ",wait_share,issue_stall_share,issuing_share\n")
    for k, c in agg.items():
        n = max(len(v) for v in c.values())
        vals = [sum(c[x]) / len(c[x]) if c.get(x) else 0.0 for x in names]
        wc = vals[0] or 1.0
        fo.write(f"{k},{n}," + ",".join(repr(v) for v in vals) + f",{vals[1]/wc:.3f},{vals[2]/wc:.3f},{vals[3]/wc:.3f}\n")
print(open(out).read()[:3000])
