#!/bin/bash
# Builds the stamped diagnostic library (-DMW_STAMP) in a scratch copy of csrc/ (the product libfcm.so is not touched)
# and runs tools/mw_stamps.py with it.  usage: bash tools/mw_stamps.sh <config> <chains> [W]   (here or on the GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WORK=${MW_STAMP_DIR:-$ROOT/gpurun_out/mw_stamp_build}
if [ ! -f $WORK/libfcm_stamp.so ]; then
  mkdir -p $WORK/pkg/csrc $WORK/include
  cp -r $ROOT/flag_complex_mcmc_amd/csrc/. $WORK/pkg/csrc/
  cp $ROOT/include/fcm.h $WORK/include/
  ( cd $WORK/pkg/csrc && rm -f *.o && make -s -j8 EXTRA=-DMW_STAMP OUT=$WORK/libfcm_stamp.so $WORK/libfcm_stamp.so >/dev/null 2>&1 )
fi
[ "${BUILD_ONLY:-0}" = "1" ] && exit 0
cd $ROOT && FCM_LIB_PATH=$WORK/libfcm_stamp.so python tools/mw_stamps.py "$@"
