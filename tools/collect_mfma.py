#!/usr/bin/env python3
"""profiles/rNN_mfma_busy.json from one rocprofv3 PMC pass over bench.py:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/rNN_pmc_mfma -- python3 bench.py ...
    python tools/collect_mfma.py rNN
SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles summed over the 1024 SIMDs (16 per 16x16x32 bf16 MFMA); GRBM_GUI_ACTIVE counts active
cycles per XCD (8 XCDs summed): mfma_busy_frac = MFMA / (GRBM / 8 * 1024).  Per kernel SYMBOL as well as for the whole run."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{rnd}_pmc_mfma", "*", "*_counter_collection.csv")), key=os.path.getmtime)
if not files:
    raise SystemExit("no PMC output")
per = {}
for r in csv.DictReader(open(files[-1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:120]
    d = per.setdefault(name, {"dispatches": set()})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    d["dispatches"].add(r["Dispatch_Id"])


def frac(d):
    g = d.get("GRBM_GUI_ACTIVE", 0.0)
    return d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g / 8 * 1024) if g else 0.0


tot = {}
gemm = {}
for name, d in per.items():
    for k, v in d.items():
        if k == "dispatches":
            continue
        tot[k] = tot.get(k, 0.0) + v
        if "gemm_" in name:
            gemm[k] = gemm.get(k, 0.0) + v
out = {"note": __doc__.split("\n", 1)[1].strip(), "all": tot, "gemm_kernels": gemm, "mfma_busy_frac_all": frac(tot), "mfma_busy_frac_gemm_kernels": frac(gemm),
       "per_kernel": {name: {"dispatches": len(d["dispatches"]), "mfma_busy_frac": round(frac(d), 4),
                             "GRBM_GUI_ACTIVE_per_dispatch": round(d.get("GRBM_GUI_ACTIVE", 0.0) / max(len(d["dispatches"]), 1), 1)}
                      for name, d in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)) if "gemm_" in name}}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_mfma_busy.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("mfma_busy_frac_all", "mfma_busy_frac_gemm_kernels")}))
for k, v in list(out["per_kernel"].items())[:12]:
    print(f"{v['mfma_busy_frac']:.3f}  x{v['dispatches']:4d}  {k[:100]}")
