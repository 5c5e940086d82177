#!/bin/bash
# For every multi-wave kernel variant: scratch instructions in the kernel body under each setting of the code-shape
# knobs (fcm_step_mw.hpp: MW_K_ZERO, MW_K_LANE, MW_K_EVLOOP).  Prints one line per variant with the best setting.
cd $(dirname $0)/../flag_complex_mcmc_amd/csrc
for t in ${@:-m2 m3 m4 m5 m6 n2 n3 n4 n5 n6}; do
  best=""; bestn=99999; all=""
  for z in 1 0; do for l in 0 1; do for e in 1 0; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DMW_K_ZERO=$z -DMW_K_LANE=$l -DMW_K_EVLOOP=$e -DFCM_TAG=${t}_0 -DFCM_MAXT=${t:1} -DFCM_EXACT=1 -DFCM_PC=$([ ${t:0:1} = m ] && echo 1 || echo 2) -DFCM_CLIQUE=0 -S --cuda-device-only -o /tmp/tune_${t}_$z$l$e.s fcm_step_variant.hip 2>/dev/null ) &
  done; done; done
  wait
  for z in 1 0; do for l in 0 1; do for e in 1 0; do
    n=$(awk '/^_Z18fcm_step_mw_kernel/,/s_endpgm/' /tmp/tune_${t}_$z$l$e.s | grep -c scratch_)
    all="$all $z$l$e:$n"
    if [ $n -lt $bestn ]; then bestn=$n; best="$z$l$e"; fi
  done; done; done
  echo "$t best(ZERO LANE EVLOOP)=$best scratch=$bestn |$all"
done
