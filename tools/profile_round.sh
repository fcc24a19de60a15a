cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03_prof
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_prof/stats -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 > gpurun_out/r03_prof/bench_under_rocprof.json 2> gpurun_out/r03_prof/bench_under_rocprof.err
for tag in DecoderB.L2.fwd DecoderA.L1.fwd EncoderB.L0.dW EncoderB.L0.fwd EncoderA.L0.fwd DecoderB.L2.dW; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r03_pmc_fetch_$tag -- python3 tools/run_dominant.py $tag > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r03_pmc_write_$tag -- python3 tools/run_dominant.py $tag > /dev/null 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r03_pmc_mfma -- python3 bench.py --cpu-steps 0 --no-probe --steps 5 --warmup 3 > gpurun_out/r03_prof/bench_pmc.json 2> gpurun_out/r03_prof/bench_pmc.err
python3 bench.py > gpurun_out/r03_prof/bench.json 2> gpurun_out/r03_prof/bench.err
ls gpurun_out/r03_prof/stats/*/ | head; tail -c 300 gpurun_out/r03_prof/bench.json
python3 bench.py --batch 4096 > gpurun_out/r03_prof/bench_b4096.json 2> /dev/null
python3 bench.py --batch 32 > gpurun_out/r03_prof/bench_b32.json 2> /dev/null
MMVAE_FORCE_DP=1 python3 bench.py --cpu-steps 0 > gpurun_out/r03_prof/bench_dp1.json 2> gpurun_out/r03_prof/bench_dp1.err
python3 tools/bench_infer.py > gpurun_out/r03_prof/infer.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_prof/scaled_stats -- python3 tools/bench_scaled.py > gpurun_out/r03_prof/scaled.log 2>&1
