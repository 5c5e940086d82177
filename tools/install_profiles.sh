#!/bin/bash
# Here, after `bash tools/collect_profiles.sh <tag>` (and, once its pmc_summary.json is in profiles/, `bash
# tools/refresh_bench_lines.sh <tag>`) ran on the GPU box: copy what is to be judged from gpurun_out/prof_<tag>/ into profiles/.
set -eu
TAG=${1:-r03}
P=gpurun_out/prof_$TAG
for f in $P/bench*.json $P/count_kernel.json $P/sq_counters*.json; do cp $f profiles/${TAG}_$(basename $f); done
cp $P/pmc_summary.json profiles/pmc_summary.json
cp $P/pmc_summary.json profiles/${TAG}_configs_pmc.json
# the kernel-trace summary of the bench command: the newest one that holds the step kernel
S=$(ls -t $P/stats/*/*_kernel_stats.csv | while read f; do grep -q fcm_step_ $f && { echo $f; break; }; done)
cp $S profiles/${TAG}_kernel_stats.csv
[ -f $P/ablation.txt ] && cp $P/ablation.txt profiles/${TAG}_ablation.txt || true
[ -f $P/mw_stamps_256chains.txt ] && cp $P/mw_stamps_256chains.txt profiles/${TAG}_mw_stamps_256chains.txt || true
grep -o '"lib_sha16": "[0-9a-f]*"' profiles/pmc_summary.json | sort | uniq -c
