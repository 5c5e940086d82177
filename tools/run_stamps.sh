#!/bin/bash
# Builds the stamped diagnostic library next to the product one and prints phase shares (GPU box).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT/flag_complex_mcmc_amd/csrc
cp ../libfcm.so /tmp/libfcm_product.so
make -s clean >/dev/null; make -s -j8 EXTRA=-DFCM_STAMP >/dev/null
cd $ROOT && python tools/run_stamps.py $1
cp /tmp/libfcm_product.so flag_complex_mcmc_amd/libfcm.so
