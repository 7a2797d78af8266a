"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_probe.py) into per-kernel HBM
bytes per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
FETCH_SIZE (KB) reports exactly half of a wide coalesced 16-B/lane stream on gfx950 -> doubled (checked here
on the two-pass GEMV kernels, whose byte count is known: 67.16 MB); WRITE_SIZE (KB) is exact.
usage: python tools/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01"""
import collections, csv, glob, json, shutil, sys

fetch_dir, write_dir, out_prefix = sys.argv[1:4]


def per_kernel(d):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return f, {k: sum(v) / len(v) * 1024.0 for k, v in agg.items()}  # KB -> bytes


ff, fetch = per_kernel(fetch_dir)
wf, write = per_kernel(write_dir)
shutil.copy(ff, out_prefix + "_pmc_fetch_size.csv")
shutil.copy(wf, out_prefix + "_pmc_write_size.csv")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/pmc_probe.py",
       "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE x1", "kernels": {}}
for k in fetch:
    short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0].strip()
    if short in ("skinny_t_kernel", "skinny_v_kernel", "skinny_pack_rows_kernel"):  # half (<= 8 right-hand sides) / full operand layout
        targs = [a.strip() for a in k.split("(")[0].split("<", 1)[1].rsplit(">", 1)[0].replace("HIP_vector_type<float, 2u>", "c32").split(",")]
        half = (targs[3] if short != "skinny_pack_rows_kernel" else targs[1]) == "true"
        short += " [half layout]" if half else " [full layout]"
        if short.startswith("skinny_t_kernel") and len(targs) > 5 and targs[5] == "true":
            short += " [V = AHA P: the Gram-mode product]"
    if short == "cgnr_gramk_resident_kernel":
        short += " [32 iterations per launch]"
    entry = {"fetch_size_raw_bytes": fetch[k], "write_size_bytes": write.get(k, 0.0),
             "hbm_bytes_per_launch": 2.0 * fetch[k] + write.get(k, 0.0)}
    # several instantiations share a short name (the probe also runs BASELINE configs[0], whose kernels are tiny): keep the largest
    if short not in out["kernels"] or entry["hbm_bytes_per_launch"] > out["kernels"][short]["hbm_bytes_per_launch"]:
        out["kernels"][short] = entry
g = out["kernels"].get("gemv_t_kernel")
if g:
    out["calibration"] = {"kernel": "gemv_t_kernel", "known_bytes": 4096 * 2048 * 8 + 6144 * 8,
                          "fetch_raw": g["fetch_size_raw_bytes"], "ratio_known_over_raw": (4096 * 2048 * 8 + 6144 * 8) / g["fetch_size_raw_bytes"]}
json.dump(out, open(out_prefix + "_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
