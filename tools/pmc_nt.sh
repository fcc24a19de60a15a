cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2_k
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r2_k/sq1 -- python3 tools/run_dominant.py DecoderB.L2.fwd > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r2_k/sq2 -- python3 tools/run_dominant.py DecoderB.L2.fwd > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d gpurun_out/r2_k/ta -- python3 tools/run_dominant.py DecoderB.L2.fwd > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("sq1","sq2","ta"):
    for f in glob.glob(f"gpurun_out/r2_k/{d}/*/*counter_collection.csv"):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_nt" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d, k, sum(v[2:])/max(1,len(v[2:])), len(v))
    for f in glob.glob(f"gpurun_out/r2_k/{d}/*/*kernel_trace.csv"):
        ds=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "gemm_nt" in r["Kernel_Name"]]
        print(d, "kernel us", [x/1e3 for x in ds])
PY
