#!/bin/bash
# census1.sh <tag e.g. m5> [extra hipcc flags]: assembly of one multi-wave variant -> /tmp/census_<tag>.s, block census -> /tmp/bb_<tag>.txt
t=$1; shift
cd $(dirname $0)/../flag_complex_mcmc_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DFCM_TAG=${t}_0 -DFCM_MAXT=${t:1} -DFCM_EXACT=1 -DFCM_PC=$([ ${t:0:1} = m ] && echo 1 || echo 2) -DFCM_CLIQUE=0 "$@" -S --cuda-device-only -o /tmp/census_$t.s fcm_step_variant.hip 2>&1 | grep -v hip-link | head -20
python3 ../../tools/bb_census.py /tmp/census_$t.s > /tmp/bb_$t.txt
tail -1 /tmp/bb_$t.txt
grep -n "vgpr_spill_count\|sgpr_spill_count" /tmp/census_$t.s | tail -2
echo "scratch ops in kernel: $(awk '/^_Z18fcm_step_mw_kernel/,/s_endpgm/' /tmp/census_$t.s | grep -c scratch_)"
