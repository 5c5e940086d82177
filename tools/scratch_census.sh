#!/bin/bash
# Scratch (VGPR spill) instructions in the body of every multi-wave kernel variant.  About a dozen belong to the two
# out-of-line calls; more means spill traffic inside the proposal loop, which costs far more than it looks.
cd $(dirname $0)/../flag_complex_mcmc_amd/csrc
for t in m2 m3 m4 m5 m6 n2 n3 n4 n5 n6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DFCM_TAG=${t}_0 -DFCM_MAXT=${t:1} -DFCM_EXACT=1 -DFCM_PC=$([ ${t:0:1} = m ] && echo 1 || echo 2) -DFCM_CLIQUE=0 -S --cuda-device-only -o /tmp/census_$t.s fcm_step_variant.hip 2>/dev/null
  echo "$t: $(awk '/^_Z18fcm_step_mw_kernel/,/s_endpgm/' /tmp/census_$t.s | grep -c scratch_)"
done
