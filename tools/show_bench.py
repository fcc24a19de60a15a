#!/usr/bin/env python3
"""Print the headline figures and the per-launch table of a bench.py JSON line:  python tools/show_bench.py gpurun_out/bench_now.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("checks"))
for key in ("roofline", "roofline_gemm"):
    r = d.get(key) or {}
    print(key, {k: r.get(k) for k in ("kernel", "launch_ms", "frac", "achieved", "traffic", "algorithmic_bytes")})
for k in sorted(d.get("kernels", []), key=lambda k: -k.get("ms", 0))[:45]:
    print(f"{k['tag']:26s} {k['ms'] * 1e3:7.1f}")
print(len(d.get("kernels", [])), "tagged launches; probed sum", d.get("probed_ms_per_step"))
