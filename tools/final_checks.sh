#!/bin/bash
# GPU box: the whole -m gpu suite, two fuzz campaigns with the seeds given, and the smoke -- all output under gpurun_out/.
# usage: bash tools/final_checks.sh <small seed> <medium seed>
S1=${1:-41}; S2=${2:-42}
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/gpu_suite_final.txt 2>&1; echo "suite rc=$?" >> gpurun_out/gpu_suite_final.txt
FCM_FUZZ_CASES=1200 FCM_FUZZ_SEED=$S1 timeout -k 10 400 python -m pytest tests/test_fuzz_parity.py -q -m gpu -x -p no:cacheprovider > gpurun_out/fuzz_small_s$S1.txt 2>&1; echo "small rc=$?" >> gpurun_out/fuzz_small_s$S1.txt
FCM_FUZZ_MEDIUM=1 FCM_FUZZ_CASES=${FCM_MEDIUM_CASES:-40} FCM_FUZZ_SEED=$S2 timeout -k 10 1000 python -m pytest tests/test_fuzz_parity.py -q -m gpu -x -p no:cacheprovider --durations=5 > gpurun_out/fuzz_medium_s$S2.txt 2>&1; echo "medium rc=$?" >> gpurun_out/fuzz_medium_s$S2.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_final.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_final.txt
for f in gpurun_out/gpu_suite_final.txt gpurun_out/fuzz_small_s$S1.txt gpurun_out/fuzz_medium_s$S2.txt gpurun_out/smoke_final.txt; do echo "== $f"; tail -c 250 $f; echo; done
