#!/bin/bash
# Builds libfcm with extra compiler flags into a scratch directory (the product libfcm.so is not touched) and prints
# its path: use it through FCM_LIB_PATH.  usage: variant_lib.sh <name> <extra flags...>
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
WORK=$ROOT/tools/_stamp/$NAME   # git-ignored; travels with gpurun pushes, so only the built library is kept there (sources removed below)
rm -rf $WORK; mkdir -p $WORK/pkg/csrc $WORK/include
cp -r $ROOT/flag_complex_mcmc_amd/csrc/. $WORK/pkg/csrc/
cp $ROOT/include/fcm.h $WORK/include/
( cd $WORK/pkg/csrc && rm -f *.o && make -s -j8 EXTRA="$*" OUT=$WORK/libfcm.so $WORK/libfcm.so >/dev/null 2>&1 )
rm -rf $WORK/pkg $WORK/include
echo $WORK/libfcm.so
