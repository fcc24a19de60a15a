#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes into profiles/rNN_traffic.json (HBM bytes per launch of the hot GEMMs).

On the GPU box, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with other traces):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01_pmc_fetch_<tag> -- python3 tools/run_dominant.py <tag>
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01_pmc_write_<tag> -- python3 tools/run_dominant.py <tag>
then here:  python tools/collect_traffic.py r01 <tag> [<tag> ...]
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-byte stores and f32/f64 atomics."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tags = sys.argv[1], sys.argv[2:]
out_path = os.path.join(ROOT, "profiles", f"{rnd}_traffic.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {"kernels": {}}
out["note"] = ("HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/run_dominant.py, B=65536, inputs "
               "rotated over 4 buffers); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE as read")


def per_launch(kind, tag):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"{rnd}_pmc_{kind}_{tag}", "*", "*_counter_collection.csv"))
    if not files:
        raise SystemExit(f"no PMC output for {kind} {tag}")
    files.sort(key=os.path.getmtime)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[-1])) if (("gemm_" in r["Kernel_Name"] or "vae_loss" in r["Kernel_Name"]) and "mm" in r["Kernel_Name"])
            and r["Counter_Name"] == ("FETCH_SIZE" if kind == "fetch" else "WRITE_SIZE")]
    vals = vals[2:]                      # the first launches also page the inputs in
    return sum(vals) / len(vals) * 1024.0, len(vals)


for tag in tags:
    f, n = per_launch("fetch", tag)
    w, _ = per_launch("write", tag)
    out["kernels"][tag] = {"fetch_bytes": 2.0 * f, "write_bytes": w, "total_bytes": 2.0 * f + w, "launches_averaged": n}
    print(tag, out["kernels"][tag])
json.dump(out, open(out_path, "w"), indent=1)
