#!/usr/bin/env python3
"""Benchmark of the MultiModalVAE training hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1 without WORLD_SIZE in the environment: starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full reference-shaped training step (optimize_hyperparameters.py:104-113) on
one synthetic minibatch per GPU: forward -> vae_loss (incl. its host read of the three loss
floats) -> zero_grad -> backward [-> gradient all-reduce] -> AdamW.  Inputs are resident in
HBM as fp32 (B,782)/(B,572) + int64 (B,), as the reference's Dataset yields them; weights are
random-init (seeded).  Prints ONE JSON line on rank 0.

The `roofline` object is measured live: HIP events on the launch stream bracket every tagged
GEMM launch of the timed steps; the dominant launch (largest share of a step) is reported with
its algorithmic bytes (operands read once + outputs written once) over its
mean duration.  `cpu_baseline` times oracle/torch_ref.py (stock PyTorch, fp32) on the host cores
for a bounded number of steps of the SAME workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd"), os.path.join(ROOT, "oracle")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

A, D, S, L = 782, 572, 24, 20           # BASELINE.json: RNA=782, DNA=572, latent=20; 24 sites (prepare_data.py:70)
FLOPS_PER_SAMPLE = 5_665_280           # 2 * (1 075 200 fwd + 1 757 440 bwd) MAC
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}


def synth_batch(B, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    a = torch.randn(B, A, generator=g).abs_()
    b = torch.rand(B, D, generator=g)
    site = torch.randint(0, S, (B,), generator=g, dtype=torch.int64)
    return a.to(device), b.to(device), site.to(device)


def launch_bytes_flops(meta):
    """Algorithmic HBM bytes and FLOPs of one tagged launch (GEMM, or a streaming kernel: bytes given by the wrapper)."""
    if meta["kind"] == "stream":
        return meta["bytes"], 0.0
    M, N, K = meta["M"], meta["N"], meta["K"]
    if meta["kind"] == "nt":
        byts = M * K * meta["a_bytes"] + (M * K if meta["pro_mask"] else 0) + M * N * meta["c_bytes"]
        if meta["epi"] in (3, 4):                # loss epilogue: the fp32 target is read, the bf16 gradient written (c_bytes)
            byts += M * N * 4
        elif meta["epi"] != 0:                   # epilogue operand (saved activation / pre-BN output) + mask
            byts += M * N * meta["act_bytes"] + (M * N if meta["epi_mask"] else 0)
        byts += N * K * meta["act_bytes"]
    else:
        byts = M * N * meta["p_bytes"] + M * K * meta["q_bytes"] + (M * K if meta["pro_mask"] else 0) + N * K * 4
    return byts, 2.0 * M * N * K


def pmc_traffic(tag, B):
    """HBM bytes per launch of `tag` from the committed rocprofv3 PMC passes (profiles/r*_traffic.json, collected at
    B=65536 with tools/run_dominant.py as MI355X_MICROARCH.md prescribes); None when that launch was not profiled."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files or B != 65536:
        return None
    k = json.load(open(files[-1])).get("kernels", {}).get(tag)
    return None if k is None else k["total_bytes"]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(B, steps):
    """oracle/torch_ref.py (stock PyTorch fp32: the reference's own op set) on the host cores.  `value` is the headline batch;
    SURVEY 8(d) also asks for the reference's native batch 32 and for 4096: `points` (a bounded number of steps each)."""
    import torch_ref
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("MMVAE_CPU_THREADS", "16"))))   # the box grants a 16-CPU share per GPU

    def run(bsz, n, warm):
        tr = torch_ref.CpuTrainer(A, D, S, L, seed=0)
        a, b, site = synth_batch(bsz, 0, "cpu")
        for _ in range(warm):
            tr.step(a, b, site)
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step(a, b, site)
        return (time.perf_counter() - t0) / n

    dt = run(B, steps, 1)
    points = {str(B): dict(ms_per_step=dt * 1e3, samples_per_s=B / dt, steps=steps)}
    for bsz, n, warm in ((32, 100, 10), (4096, 10, 2)):
        if bsz != B:
            d = run(bsz, n, warm)
            points[str(bsz)] = dict(ms_per_step=d * 1e3, samples_per_s=bsz / d, steps=n)
    return dict(value=B / dt, unit="samples/s", cores=torch.get_num_threads(), cpu=cpu_model(), kind="port", points=points,
                sample=f"{steps} full training steps (after 1 warm-up) of oracle/torch_ref.py, stock PyTorch fp32, batch {B}, "
                       f"same synthetic workload; {dt * 1e3:.0f} ms/step; `points`: the same step at batch 32 (100 steps) and 4096 (10 steps)")


def spawn_ranks(n):
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, the same command line), as
    `python -m torch.distributed.run --nproc-per-node n` would start them; rank 0's stdout (the ONE JSON line) is ours.  Returns the
    exit status: non-zero as soon as any rank fails (the others are then terminated -- they would wait in a collective for ever)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
               MASTER_PORT=os.environ.get("MASTER_PORT", str(port)), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    status = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                print(f"bench.py: rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr)
                for o in pending:
                    procs[o].terminate()
        time.sleep(0.05)
    return status


def load_expectations():
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_expect.json")))
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536, help="rows per GPU (weak scaling)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed steps of the CPU baseline (0 = skip)")
    ap.add_argument("--no-probe", action="store_true", help="do not bracket GEMM launches with events")
    ap.add_argument("--eager", action="store_true", help="issue every launch from Python instead of replaying the captured hipGraph")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: put every rank on cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This process has not touched the GPU yet
        # (torch is imported, no HIP call was made) and never will: it only waits for its children.
        raise SystemExit(spawn_ranks(args.gpus))
    # stdout carries ONE line, the JSON: everything else a library prints there (RCCL writes a five-line version banner to stdout when
    # its communicator is created) is sent to stderr by pointing file descriptor 1 at it until the line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dp = os.environ.get("MMVAE_FORCE_DP") == "1"          # rehearsal: the data-parallel code path with ONE rank
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from mmvae import ops, parallel
    from mmvae.optim import FusedAdamW
    from src.models import MultiModalVAE
    from src.utils import vae_loss

    torch.manual_seed(0)
    model = MultiModalVAE(A, D, S, L).to(dev).set_precision(args.precision)
    dp = world > 1 or force_dp
    if dp:
        parallel.broadcast_parameters(model)
        parallel.attach(model, overlap=True)
    opt = FusedAdamW(model.parameters(), lr=5e-4, weight_decay=1e-5)
    B = args.batch
    a, b, site = synth_batch(B, rank, dev)
    model.train()

    def eager_step():
        ra, rb, rc, mu, lv = model(a=a, b=b, site=site)
        loss, rec, cls, kld = vae_loss(ra, a, rb, b, rc, site, mu, lv, beta=1e-3, gamma=1.0)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return rec

    # Step 0, eager and CHECKED: fresh seed-0 weights, Philox offset 0 -> the total loss must be the stored expectation
    # (tests/golden/bench_expect.json; tied to the CPU oracle by tests/test_fullsize_gpu.py).  A wrong kernel cannot hide
    # behind "the loss is not NaN".
    expect = load_expectations() if (B == 65536 and args.precision == "bf16" and not dp and (A, D, S, L) == (782, 572, 24, 20)) else None
    steps_done = 0
    checks = {}
    if expect is not None:
        ra, rb, rc, mu, lv = model(a=a, b=b, site=site)
        loss0, rec0, _, _ = vae_loss(ra, a, rb, b, rc, site, mu, lv, beta=1e-3, gamma=1.0)
        opt.zero_grad(); loss0.backward(); opt.step()
        steps_done += 1
        first = float(loss0.item())
        rel = abs(first - expect["first_step_total"]) / expect["first_step_total"]
        checks["first_step_total"] = dict(got=first, recon=rec0, expected=expect["first_step_total"], rel_err=rel, rtol=expect["rtol_first"])
        if not rel <= expect["rtol_first"]:
            raise SystemExit(f"first-step loss {first} differs from the stored expectation {expect['first_step_total']} by {rel:.2e}")
        del ra, rb, rc, mu, lv, loss0

    # Default on one GPU: the SAME step captured once as a hipGraph (mmvae.graphs) and replayed; the loss floats of every
    # step are still read on the host, as the reference's loop does to log them -- one step behind the launches.
    # N > 1: three graphs, [forward, loss, decoder backward] | [encoder backward] | [AdamW]; the RCCL all-reduce of the decoder half
    # of the flat gradient arena runs beside the second graph, the encoder half after it (nothing of RCCL is captured).
    graphed = None
    if not args.eager:
        from mmvae.graphs import GraphedTrainStep
        reduce = (lambda flat, async_op=False: dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)) if dp else None
        # a capture failure ends the run (every rank runs the same code): no silent change of the launch form -- `--eager` is the
        # explicit way to measure Python-issued launches
        graphed = GraphedTrainStep(model, opt, a, b, site, beta=1e-3, gamma=1.0, warmup=2, reduce=reduce)
        steps_done += 2

    def graph_step():
        # every step's loss floats reach the host, one step behind the launches (GraphedTrainStep.step_logged): the host never
        # stalls the GPU between two replays.  MMVAE_SYNC_LOSS=1: read them right after each replay instead (A/B switch).
        if os.environ.get("MMVAE_SYNC_LOSS") == "1":
            graphed()
            return graphed.losses()[1]
        prev = graphed.step_logged()
        return None if prev is None else prev[1]

    step = graph_step if graphed is not None else eager_step

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    steps_done += args.warmup
    # untimed survey pass: every tagged GEMM launch bracketed by events, to find the dominant one
    probe = survey = None
    if not args.no_probe:
        survey = ops.PROBE = ops.KernelProbe()
        # events need eager launches (a captured graph has no Python in it): the captured work itself, issued from Python
        # (GraphedTrainStep.run_eager -- incl. the reconstruction losses inside the decoder GEMMs, which the public
        # model() + vae_loss() call sequence of eager_step cannot use because it hands the reconstructions to the caller)
        probe_step = graphed.run_eager if (graphed is not None and not dp) else eager_step
        for _ in range(3):
            probe_step()
        steps_done += 3
        torch.cuda.synchronize()
        ssum = survey.summary()
        dom_tag = max(ssum, key=lambda t: ssum[t]["mean_ms"] * ssum[t]["calls"])
        gemm_tags = [t for t in ssum if ssum[t]["meta"]["kind"] != "stream"]
        dom_gemm = max(gemm_tags, key=lambda t: ssum[t]["mean_ms"] * ssum[t]["calls"])
        # timed region: only the dominant launch is bracketed (2 events per step); under graph replay no Python runs,
        # so the dominant launch is timed in a separate eager pass right after the timed region instead
        probe = ops.PROBE = ops.KernelProbe(only={dom_tag, dom_gemm}) if graphed is None else None
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    if graphed is not None and os.environ.get("MMVAE_SYNC_LOSS") != "1":
        last = graphed.flush_logged()[1]          # the last step's losses: read inside the timed region
    fence()
    dt = time.perf_counter() - t0
    steps_done += args.steps
    # the last timed step's reconstruction loss against the stored trajectory value for exactly this many steps (default flags)
    if expect is not None and graphed is not None:
        want = expect.get("after_steps", {}).get(str(steps_done))
        if want is not None:
            rel = abs(last - want["recon"]) / want["recon"]
            checks["last_step_recon"] = dict(got=last, expected=want["recon"], rel_err=rel, rtol=want["rtol"], after_steps=steps_done)
            if not rel <= want["rtol"]:
                raise SystemExit(f"reconstruction loss after {steps_done} steps: {last}, stored expectation {want['recon']} (rel {rel:.2e})")
        else:
            checks["last_step_recon"] = dict(got=last, expected=None, after_steps=steps_done)
        if not last < checks["first_step_total"]["recon"]:
            raise SystemExit("the loss did not go down during the benchmark")
    if graphed is not None and survey is not None:
        probe = ops.PROBE = ops.KernelProbe(only={dom_tag, dom_gemm})
        for _ in range(args.steps):
            probe_step()
        torch.cuda.synchronize()
    ops.PROBE = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not (last == last):
        raise SystemExit("loss became NaN during the benchmark")

    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    out = {
        "metric": "training samples/sec, MultiModalVAE batch 65536", "value": value, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"MultiModalVAE full training step (fwd + vae_loss + bwd + AdamW), RNA={A} DNA={D} sites={S} latent={L}, "
                               f"batch {B} per GPU, fp32 inputs resident in HBM, random-init weights (seed 0)",
                   "global_batch": world * B, "parallelism": f"dp{world}" if world > 1 else "single",
                   "grad_allreduce": (None if not dp else "RCCL SUM over the flat fp32 gradient arena: decoder half asynchronously beside the encoder-backward graph, encoder half after it" if graphed is not None
                                      else "RCCL SUM over flat fp32 arena, decoder bucket overlapped with encoder backward"),
                   "launch": ("eager (Python-issued launches)" if graphed is None else
                              "hipGraph replay (1 launch/step)" if not dp else "3 hipGraph replays/step ([fwd, loss, decoder bwd] | [encoder bwd] | [AdamW]) around the eager all-reduces"),
                   "loss_logging": ("every step's [total, recon, class, kld] reaches the host, as optimize_hyperparameters.py:113 reads it, but ONE STEP "
                                    "LATE (pinned 20-byte copy behind each replay; the last step's is read inside the timed region)"
                                    if (graphed is not None and os.environ.get("MMVAE_SYNC_LOSS") != "1") else "read right after each step")},
        "checks": checks,
        "step_tflops": FLOPS_PER_SAMPLE * B / (ms * 1e-3) / 1e12,
        "step_mfma_frac": FLOPS_PER_SAMPLE * B / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[args.precision],
    }
    if rank == 0:
        if probe is not None:
            summ = ssum
            summ.update(probe.summary())         # the dominant launches: timed-region measurement
            kernels = []
            for tag, s in summ.items():
                byts, flops = launch_bytes_flops(s["meta"])
                per_step_ms = s["mean_ms"] * s["calls"] / (args.steps if tag in (dom_tag, dom_gemm) else 3)
                kernels.append(dict(tag=tag, ms=s["mean_ms"], per_step_ms=per_step_ms, GBs=byts / s["mean_ms"] / 1e6,
                                    TFLOPs=flops / s["mean_ms"] / 1e9, bytes=byts, flops=flops))
            kernels.sort(key=lambda k: -k["per_step_ms"])

            def roof(dom, definition):
                t_hbm = dom["bytes"] / (HBM_PEAK_GBS * 1e9)
                t_mfma = dom["flops"] / (MFMA_PEAK_TFLOPS[args.precision] * 1e12)
                traffic = pmc_traffic(dom["tag"], B) if args.precision == "bf16" else None
                if t_hbm >= t_mfma:
                    r = {"bound": "hbm", "achieved": dom["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["GBs"] / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": dom["tag"], "launch_ms": dom["ms"], "algorithmic_bytes": dom["bytes"]}
                else:
                    r = {"bound": "mfma", "achieved": dom["TFLOPs"], "peak": MFMA_PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                         "frac": dom["TFLOPs"] / MFMA_PEAK_TFLOPS[args.precision], "traffic": traffic, "kernel": dom["tag"],
                         "launch_ms": dom["ms"], "algorithmic_flops": dom["flops"]}
                r["definition"] = definition
                return r
            how = ("duration = HIP events on the launch stream around that launch over the timed steps (eager pass), bytes = algorithmic "
                   "operand + result bytes of the launch")
            out["roofline"] = roof(next(k for k in kernels if k["tag"] == dom_tag), "largest single launch of the step, any kind; " + how)
            out["roofline_gemm"] = roof(next(k for k in kernels if k["tag"] == dom_gemm), "largest single GEMM launch of the step; " + how)
            # the same figure per kernel SYMBOL (what rocprofv3 --stats lists): launches grouped by the template instantiation they select
            fam = {}
            for k in kernels:
                m = summ[k["tag"]]["meta"]
                if m["kind"] == "stream":
                    name = k["tag"].split(".")[0]
                elif m["kind"] == "nt":
                    name = (f"gemm_nt<{'f32' if m['a_bytes'] == 4 else 'bf16'} A{'+BN' if m.get('pro') else ''}, "
                            f"{('store', 'relu-mask', 'bn-bwd', 'mse-loss', 'bce-loss')[m['epi']]}, {'f32' if m['c_bytes'] == 4 else 'bf16'} C, {'128x256' if m['N'] % 256 == 0 else '128x128'}>")
                else:
                    name = f"gemm_tn<{'f32' if m['p_bytes'] == 4 else 'bf16'} P, {'f32' if m['q_bytes'] == 4 else 'bf16'} Q{'+BN' if m['pro_mask'] else ''}>"
                f = fam.setdefault(name, dict(symbol=name, launches_per_step=0, ms_per_step=0.0, bytes=0.0, flops=0.0, tags=[]))
                f["launches_per_step"] += 1; f["ms_per_step"] += k["per_step_ms"]; f["bytes"] += k["bytes"]; f["flops"] += k["flops"]; f["tags"].append(k["tag"])
            top = max(fam.values(), key=lambda f: f["ms_per_step"])
            out["roofline_by_symbol"] = {"symbol": top["symbol"], "launches_per_step": top["launches_per_step"], "tags": top["tags"],
                                         "avg_launch_ms": top["ms_per_step"] / top["launches_per_step"], "bound": "hbm",
                                         "achieved": top["bytes"] / top["ms_per_step"] / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": top["bytes"] / top["ms_per_step"] / 1e6 / HBM_PEAK_GBS,
                                         "TFLOPs": top["flops"] / top["ms_per_step"] / 1e9}
            out["gemm_ms_per_step"] = sum(k["per_step_ms"] for k in kernels if summ[k["tag"]]["meta"]["kind"] != "stream")
            out["probed_ms_per_step"] = sum(k["per_step_ms"] for k in kernels)
            out["kernels"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in kk.items() if k not in ("bytes", "flops")}
                              for kk in kernels[:40]]
        if world == 1 and args.cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(B, args.cpu_steps)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
