#!/bin/bash
# Builds the stamped diagnostic library (-DMW_STAMP) in a scratch copy of csrc/ (the product libfcm.so is not touched)
# and runs tools/mw_stamps.py with it.  usage: bash tools/mw_stamps.sh <config> <chains> [W]   (here or on the GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WORK=${MW_STAMP_DIR:-$ROOT/gpurun_out/mw_stamp_build}
# rebuilt whenever a source is newer than the stamped library (a stale one would time the wrong kernel)
if [ ! -f $WORK/libfcm_stamp.so ] || [ -n "$(find $ROOT/flag_complex_mcmc_amd/csrc $ROOT/include -newer $WORK/libfcm_stamp.so \( -name '*.hpp' -o -name '*.hip' -o -name '*.cpp' -o -name '*.h' -o -name Makefile \) -print -quit)" ]; then
  rm -rf $WORK
  mkdir -p $WORK/pkg/csrc $WORK/include
  cp -r $ROOT/flag_complex_mcmc_amd/csrc/. $WORK/pkg/csrc/
  cp $ROOT/include/fcm.h $WORK/include/
  ( cd $WORK/pkg/csrc && rm -f *.o && make -s -j8 EXTRA=-DMW_STAMP OUT=$WORK/libfcm_stamp.so $WORK/libfcm_stamp.so >/dev/null 2>&1 )
  rm -rf $WORK/pkg $WORK/include   # only the library stays: no copy of the sources is left behind
fi
[ "${BUILD_ONLY:-0}" = "1" ] && exit 0
cd $ROOT && FCM_LIB_PATH=$WORK/libfcm_stamp.so python tools/mw_stamps.py "$@"
