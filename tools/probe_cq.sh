#!/bin/bash
# Instruction cost of the parts of a clique move by difference: SQ counters of probe builds of the clique-move kernel
# (CQ_PROBE: 1 no patch, 2 no evaluations, 3 rows requested but not consumed, 4 no pair loop) against the product build.
# Build here:   for v in 1 2 3 4; do bash tools/variant_lib.sh cqp$v -DCQ_PROBE=$v; done
# GPU box:      bash tools/probe_cq.sh "base cqp1 cqp2 cqp3 cqp4" --weights 0,0,1,0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=$1; shift
for V in $LIBS; do
  [ $V = base ] && unset FCM_LIB_PATH || export FCM_LIB_PATH=$ROOT/tools/_stamp/$V/libfcm.so
  echo "== $V"
  FCM_BENCH_PROBE=1 bash $ROOT/tools/sq_quick.sh --proposals 1024 "$@" 2>&1 | tail -1
done
