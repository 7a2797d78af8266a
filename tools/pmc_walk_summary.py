"""bytes per launch of the slab kernels from two rocprofv3 --pmc passes over tools/pmc_probe_walk.py:
python tools/pmc_walk_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>   (FETCH_SIZE doubled, KB -> bytes: the guide's
gfx950 corrections, as bench.py's live_pmc does)"""
import csv, glob, os, sys
def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"]) * 1024.0))
    return rows
F, W = load(sys.argv[1]), load(sys.argv[2])
def key(name, grid):
    short = name.split("(anonymous namespace)::")[-1].split("(")[0]
    return f"{short} grid={grid}"
agg = {}
for rows, which in ((F, 0), (W, 1)):
    for _, name, grid, v in rows:
        if "cgnr_pipe_a_kernel" not in name and "cgnr_pipe_r_kernel" not in name: continue
        agg.setdefault(key(name, grid), [[], []])[which].append(v)
for k, (f, w) in sorted(agg.items()):
    fm, wm = sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)
    print(f"{k}: {len(f)} launches, FETCH_SIZE x 2 = {2 * fm / 1e6:.1f} MB, WRITE_SIZE = {wm / 1e6:.1f} MB, traffic = {(2 * fm + wm) / 1e6:.1f} MB per launch")
