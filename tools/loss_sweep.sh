cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wg in 8 4 3 2; do
  rm -rf gpurun_out/loss_wg$wg
  MMVAE_LOSS_WG=$wg timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/loss_wg$wg -- python3 bench.py --cpu-steps 0 --no-probe --steps 20 --warmup 5 > gpurun_out/loss_wg$wg.json 2>/dev/null || exit 1
  echo "wg=$wg" $(grep vae_loss gpurun_out/loss_wg$wg/*/*_kernel_stats.csv | awk -F, '{print $(NF-4)}') $(python3 -c "import json;print(json.loads(open('gpurun_out/loss_wg$wg.json').readline())['ms_per_step'])")
done
