#!/bin/bash
# GPU box: quick check of a kernel change -- the parity tests that exercise the step kernels, then the headline bench and the
# default-mix bench (short runs).  usage: bash tools/quick_ab.sh <tag> [tests: 1/0]
TAG=${1:-x}; TESTS=${2:-1}
mkdir -p gpurun_out
if [ "$TESTS" = 1 ]; then
  timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_clique_cooperative.py tests/test_baseline_configs.py tests/test_sparse_state.py -q -m gpu -x -p no:cacheprovider > gpurun_out/qab_${TAG}_suite.txt 2>&1; echo "suite rc=$?" >> gpurun_out/qab_${TAG}_suite.txt
  tail -2 gpurun_out/qab_${TAG}_suite.txt
fi
python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/qab_${TAG}_head.json 2> gpurun_out/qab_${TAG}_head.err
python bench.py --steps 4 --warmup 1 --moves default > gpurun_out/qab_${TAG}_def.json 2> gpurun_out/qab_${TAG}_def.err
python - <<PY
import json
for f in ("head", "def"):
    try:
        d = json.load(open("gpurun_out/qab_${TAG}_%s.json" % f)); print("${TAG}", f, "%.4g" % d["value"], "%.2f ms" % d["kernel_ms_per_launch"])
    except Exception as e:
        print("${TAG}", f, "FAILED", e)
PY
