#!/bin/bash
# Builds tools/_stamp/knob_<name>/libfcm.so: the headline kernel's object (m5_0) compiled with the given extra flags, the rest
# the product build's objects.  usage: bash tools/knob_build.sh <name> <flags...>
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$ROOT/flag_complex_mcmc_amd/csrc
NAME=$1; shift
W=$ROOT/tools/_stamp/knob_$NAME; rm -rf $W; mkdir -p $W
cd $SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DFCM_TAG=m5_0 -DFCM_MAXT=5 -DFCM_EXACT=1 -DFCM_PC=1 -DFCM_CLIQUE=0 "$@" -c fcm_step_variant.hip -o $W/stepk_m5_0.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DFCM_TAG=m5_0 -DFCM_MAXT=5 -DFCM_EXACT=1 -DFCM_PC=1 -DFCM_CLIQUE=0 "$@" -S --cuda-device-only -o $W/m5.s fcm_step_variant.hip 2>/dev/null
echo "$NAME: scratch $(awk '/^_Z18fcm_step_mw_kernel/,/s_endpgm/' $W/m5.s | grep -c scratch_) lines $(awk '/^_Z18fcm_step_mw_kernel/,/s_endpgm/' $W/m5.s | wc -l)"
rm -f $W/m5.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $W/libfcm.so $(ls stepk_*.o | grep -v stepk_m5_0.o) $W/stepk_m5_0.o fcm_count.o fcm_host.o
rm -f $W/stepk_m5_0.o
