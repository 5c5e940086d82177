#!/bin/bash
# Builds the stamped diagnostic library (-DFCM_STAMP) in a scratch copy of csrc/ and points
# run_stamps.py at it through FCM_LIB_PATH, so the product libfcm.so and its objects are never
# touched (GPU box).  Usage: bash tools/run_stamps.sh flips|clique
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WORK=$(mktemp -d /tmp/fcm_stamp.XXXXXX)
trap 'rm -rf "$WORK"' EXIT
mkdir -p $WORK/pkg/csrc $WORK/include
cp -r $ROOT/flag_complex_mcmc_amd/csrc/. $WORK/pkg/csrc/
cp $ROOT/include/fcm.h $WORK/include/
( cd $WORK/pkg/csrc && rm -f *.o && make -s -j8 EXTRA=-DFCM_STAMP OUT=$WORK/libfcm_stamp.so $WORK/libfcm_stamp.so >/dev/null )
cd $ROOT && FCM_LIB_PATH=$WORK/libfcm_stamp.so python tools/run_stamps.py $1
