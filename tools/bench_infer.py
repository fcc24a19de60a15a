#!/usr/bin/env python3
"""Inference throughput of the drop-in modules: eval-mode forward under torch.no_grad(), B = 65 536, bf16.

Call shapes of the reference's downstream scripts:
  all modalities            model(a=, b=, site=)            /root/reference/downstream_task.py:29-33 (latent extraction + all decoders)
  RNA only                  model(a=)                       /root/reference/downstream_task.py:45-49
  DNA2RNAVAE, site=None     model(dna=)                     /root/reference/reconstruct_unmatched.py:188-194
  RNA2DNAVAE, site=None     model(rna=)                     same script, other direction
Each is timed eagerly (Python-issued launches, HIP events on the launch stream, 30 calls after 5 warm-up) next to the SAME call of
oracle/torch_ref.py (stock PyTorch fp32) on the host cores (a bounded number of calls).  Prints one JSON object; meant to be run on the
GPU box:  python tools/bench_infer.py > gpurun_out/infer.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd"), os.path.join(ROOT, "oracle")]
import torch  # noqa: E402

A, D, S, L = 782, 572, 24, 20
B = int(os.environ.get("B", "65536"))
dev = "cuda"


def gpu_time(f, n=30, warm=5):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def cpu_time(f, n=3):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from src.models import MultiModalVAE, DNA2RNAVAE, RNA2DNAVAE
    import np_oracle as O
    import torch_ref as T
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    a = torch.randn(B, A, generator=g).abs_()
    b = torch.rand(B, D, generator=g)
    site = torch.randint(0, S, (B,), generator=g)
    da, db, ds = a.to(dev), b.to(dev), site.to(dev)
    mm = MultiModalVAE(A, D, S, L).to(dev).set_precision("bf16").eval()
    d2r = DNA2RNAVAE(A, D, S, L).to(dev).set_precision("bf16").eval()
    r2d = RNA2DNAVAE(A, D, S, L).to(dev).set_precision("bf16").eval()
    threads = min(os.cpu_count() or 1, int(os.environ.get("MMVAE_CPU_THREADS", "16")))
    torch.set_num_threads(threads)
    P, Bf = O.make_params(0, A, D, S, L, 32)
    p, bufs = T.to_torch(P, Bf, requires_grad=False)

    def cpu_directional(x_is_dna):
        # the directional models are sub-graphs: one encoder, reparameterise, one decoder (directional_vae.py:12-111)
        if x_is_dna:
            mu, lv = T._encoder(b, p, bufs, "encoder_b", O.ENC_B_IDX, False, None)
        else:
            mu, lv = T._encoder(a, p, bufs, "encoder_a", O.ENC_A_IDX, False, None)
        z = mu + torch.randn_like(mu) * torch.exp(0.5 * lv)
        return T._decoder(z, p, "decoder_a", [0, 2], False) if x_is_dna else T._decoder(z, p, "decoder_b", [0, 2, 4], True)

    cases = [
        ("MultiModalVAE(a, b, site)", lambda: mm(a=da, b=db, site=ds), lambda: T.forward(p, bufs, a, b, site, False)),
        ("MultiModalVAE(a)", lambda: mm(a=da), lambda: T.forward(p, bufs, a, None, None, False)),
        ("DNA2RNAVAE(dna, site=None)", lambda: d2r(dna=db), lambda: cpu_directional(True)),
        ("RNA2DNAVAE(rna, site=None)", lambda: r2d(rna=da), lambda: cpu_directional(False)),
    ]
    out = {"batch": B, "dtype": "bf16", "cpu": {"threads": threads, "kind": "port (oracle/torch_ref.py, stock PyTorch fp32)"}, "cases": []}
    with torch.no_grad():
        for name, gf, cf in cases:
            ms = gpu_time(gf)
            cms = cpu_time(cf)
            out["cases"].append(dict(call=name, gpu_ms=round(ms, 4), gpu_samples_per_s=round(B / ms * 1e3), cpu_ms=round(cms, 1),
                                     cpu_samples_per_s=round(B / cms * 1e3)))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
