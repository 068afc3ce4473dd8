#!/bin/bash
# Copy one tools/profile_all.sh pass (gpurun_out/prof_<tag>) into profiles/ under the round's names:
#   tools/copy_profiles.sh r02e r02
set -e
S=gpurun_out/prof_$1; R=$2; P=profiles
cp $S/bench_default_run.json $P/${R}_bench_default_run.json
cp $S/bench_under_rocprof.json $P/${R}_bench_under_rocprof.json
cp $S/bench_tin_b128.json $P/${R}_bench_tin_b128.json
cp $S/bench_tin_b512.json $P/${R}_bench_tin_b512.json
cp $S/bench_tin_under_rocprof.json $P/${R}_bench_tin_b128_under_rocprof.json
cp $S/family_summary.txt $P/${R}_family_summary.txt
cp $S/family_summary_tin_b128.txt $P/${R}_family_summary_tin_b128.txt
cp $S/kt/k_kernel_stats.csv $P/${R}_kernel_stats_bench_hipgraph_B1024.csv
cp $S/kt_tin/k_kernel_stats.csv $P/${R}_kernel_stats_tin_b128_hipgraph.csv
cp $S/mfma_util.txt $P/${R}_mfma_util.txt
cp $S/traffic.txt $P/${R}_traffic.txt
cp $S/mfma_util.json $P/mfma_util.json
cp $S/traffic.json $P/traffic.json
cp $S/pmc_fetch/f_counter_collection.csv $P/${R}_pmc_fetch_size_eager2steps.csv
cp $S/pmc_write/w_counter_collection.csv $P/${R}_pmc_write_size_eager2steps.csv
gzip -c $S/pmc_mfma/m_counter_collection.csv > $P/${R}_pmc_mfma_eager2steps.csv.gz
