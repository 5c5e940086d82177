#!/bin/bash
# Builds tools/_stamp/knob_<name>/libfcm.so: one kernel object compiled with the given extra flags, the rest the product build's
# objects.  The object: the headline kernel (m5_0), or KNOB_TAG=<tag> (Makefile tags: x5_1 = the one-wave kernel with clique moves, c5_1 =
# the cooperative one).  usage: [KNOB_TAG=x5_1] bash tools/knob_build.sh <name> <flags...>
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$ROOT/flag_complex_mcmc_amd/csrc
NAME=$1; shift
TAG=${KNOB_TAG:-m5_0}
W=$ROOT/tools/_stamp/knob_$NAME; rm -rf $W; mkdir -p $W
cd $SRC
head=${TAG%%_*}; CLQ=${TAG##*_}
MAXT=$(echo $head | sed 's/^[a-z]//')
case $head in m*) PC=1; EX=1;; n*) PC=2; EX=1;; c*) PC=3; EX=1;; s*) PC=5; EX=1;; x*) PC=0; EX=1;; *) PC=0; EX=0;; esac
FL="-DFCM_TAG=$TAG -DFCM_MAXT=$MAXT -DFCM_EXACT=$EX -DFCM_PC=$PC -DFCM_CLIQUE=$CLQ"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $FL "$@" -c fcm_step_variant.hip -o $W/stepk_$TAG.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $FL "$@" -S --cuda-device-only -o $W/k.s fcm_step_variant.hip 2>/dev/null
echo "$NAME ($TAG): scratch $(awk '/^_Z[0-9]*fcm_step_/,/s_endpgm/' $W/k.s | grep -c scratch_) lines $(awk '/^_Z[0-9]*fcm_step_/,/s_endpgm/' $W/k.s | wc -l) $(grep -E 'sgpr_spill_count' $W/k.s | tail -1 | tr -s ' ')"
rm -f $W/k.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $W/libfcm.so $(ls stepk_*.o | grep -v stepk_$TAG.o) $W/stepk_$TAG.o fcm_count.o fcm_host.o
rm -f $W/stepk_$TAG.o
