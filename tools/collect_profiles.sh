#!/bin/bash
# Copies what tools/profile_round.sh <tag> left under gpurun_out/ into profiles/ (tracked): kernel stats of the bench command,
# PMC traffic (the file bench.py reads), SQ counters, knn stats, the build identity.  Usage: bash tools/collect_profiles.sh r03a
T=${1:?tag}
set -e
cp gpurun_out/prof_$T/stats/*/*kernel_stats.csv profiles/${T}_kernel_stats.csv
cp gpurun_out/prof_$T/knn/*/*kernel_stats.csv profiles/${T}_knn2_kernel_stats.csv
cp gpurun_out/pmc_$T/summary.json profiles/${T}_pmc_traffic.json
cp gpurun_out/pmc_$T/summary.json profiles/r04_pmc_traffic.json
cp gpurun_out/sq_$T/summary.json profiles/r04_sq_counters.json
[ -f gpurun_out/pmcph_$T/summary.json ] && cp gpurun_out/pmcph_$T/summary.json profiles/${T}_pmc_traffic_phases.json
cp gpurun_out/sq_$T/summary.json profiles/${T}_sq_counters.json
cp gpurun_out/prof_$T/build.json profiles/${T}_build.json
[ -f gpurun_out/bench_$T.json ] && cp gpurun_out/bench_$T.json profiles/${T}_bench_driver_cmd.json
ls -la profiles/${T}_* profiles/r04_pmc_traffic.json
