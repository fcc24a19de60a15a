#!/bin/bash
# Experimental libraries of the wave-specialised NT kernel: libmmvae_<name>.so = the product objects with gemm_ntp.hip rebuilt under the
# given flags (see the switches at the top of gemm_ntp.h), selected with MMVAE_LIB_PATH / tools/bench_ntp.py ABL=<name>.
#   tools/abl_ntp.sh abl1 -DNTP_ABL=1  abl2 -DNTP_ABL=2  nt '-DNTP_A_POLICY=" nt"'
# NTP_ABL libraries are timing-only: their results are WRONG by construction.
set -e
cd "$(dirname "$0")/../vae-los-angeles_amd/csrc"
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc $f -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-result -munsafe-fp-atomics -c gemm_ntp.hip -o gemm_ntp.$n.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gemm_nt.o gemm_ntp.$n.o gemm_tn.o gemm_tn_wide.o elementwise.o -o ../mmvae/libmmvae_$n.so
done
