#!/bin/bash
# Builds tools/_stamp/abl<v>/libfcm.so for the given MW_ABL values: only the headline kernel's object (m5_0) is compiled with the
# probe flag, the rest is the product build's objects.  usage: bash tools/abl_build.sh 1 3 7 ...
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$ROOT/flag_complex_mcmc_amd/csrc
make -s -C $SRC -j8 >/dev/null 2>&1
for v in "$@"; do
  W=$ROOT/tools/_stamp/abl$v; rm -rf $W; mkdir -p $W
  ( cd $SRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DFCM_TAG=m5_0 -DFCM_MAXT=5 -DFCM_EXACT=1 -DFCM_PC=1 -DFCM_CLIQUE=0 -DMW_ABL=$v -c fcm_step_variant.hip -o $W/stepk_m5_0.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $W/libfcm.so $(ls stepk_*.o | grep -v stepk_m5_0.o) $W/stepk_m5_0.o fcm_count.o fcm_host.o ) &
done
wait
for v in "$@"; do rm -f $ROOT/tools/_stamp/abl$v/stepk_m5_0.o; ls -la $ROOT/tools/_stamp/abl$v/libfcm.so | awk '{print $5, $9}'; done
