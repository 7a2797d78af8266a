"""Markdown table of the parity gate's records (tests/conftest.py writes gpurun_out/parity_errors.jsonl; a copy of the
round's full GPU run is kept under profiles/): per case the device error against the float64 oracle and the error of
the oracle's own Float32 run.  usage: python tools/parity_table.py profiles/r02_parity_errors.jsonl"""
import collections, json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
print(f"{len(rows)} recorded comparisons; {sum(r['err_gpu_vs_f64'] <= r['tol'] for r in rows)} within 1e-5 outright, "
      f"{sum(r['err_gpu_vs_f64'] > r['tol'] for r in rows)} through the Float32 bound")
groups = collections.OrderedDict()
def key(tag):
    for pre in ("BASELINE config 3", "BASELINE config 4", "BASELINE config 5", "config5_schedule", "cgnr_resident_4096x2048_complex64", "fista_resident_4096x2048_complex64",
                "fista_gram_4096x2048", "cgnr_gram_4096x2048", "admm_tv_8192x4096", "admm_tv_gram_8192x4096", "admm_batched", "fista_batched", "batched_cgnr", "fista_rowsharded", "admm_rowsharded",
                "kaczmarz", "golden_next_tier", "OptISTA", "POGM", "splitbregman", "multisolve", "cgnr_weighted", "cgnr_normalized", "fista_normalized"):
        if pre in tag:
            return pre
    return tag.split("_it")[0]
for r in rows:
    groups.setdefault(key(r["tag"]), []).append(r)
print("| case (all recorded comparisons of the group) | n | max device error vs float64 oracle | Float32 oracle vs float64 (max, where evaluated) |")
print("|---|---|---|---|")
for k, v in sorted(groups.items(), key=lambda kv: -max(r["err_gpu_vs_f64"] for r in kv[1])):
    e = max(r["err_gpu_vs_f64"] for r in v)
    e32 = [r["err_f32_oracle_vs_f64"] for r in v if r["err_f32_oracle_vs_f64"] is not None]
    print(f"| {k} | {len(v)} | {e:.2e} | {max(e32):.2e} |" if e32 else f"| {k} | {len(v)} | {e:.2e} | — |")
