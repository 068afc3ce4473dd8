#!/bin/bash
# Copy one tools/profile_all.sh pass (gpurun_out/prof_<tag>) into profiles/ under the round's names:
#   tools/copy_profiles.sh r03 r03
set -e
S=gpurun_out/prof_$1; R=$2; P=profiles
for f in bench_default_run bench_under_rocprof bench_tin_b128 bench_tin_b512 bench_tin_b128_under_rocprof bench_tin_b512_under_rocprof bench_q32_eval bench_q32_eval_v2 bench_q32_under_rocprof \
         bench_force_ddp bench_force_ddp_all_tags bench_force_ddp_no_tags bench_lateral_one_stream; do
  [ -f $S/$f.json ] && cp $S/$f.json $P/${R}_$f.json
done
for f in family_summary family_summary_tin_b128 family_summary_tin_b512 family_summary_q32_eval family_summary_force_ddp overlap step_timeline mfma_util traffic bench_branch branch_mfma_pmc tn_census tn_stream bench_tl chain_stamps; do
  [ -f $S/$f.txt ] && cp $S/$f.txt $P/${R}_$f.txt
done
cp $S/kt/k_kernel_stats.csv $P/${R}_kernel_stats_bench_hipgraph_B1024.csv
cp $S/kt_tin128/k_kernel_stats.csv $P/${R}_kernel_stats_tin_b128_hipgraph.csv
cp $S/kt_tin512/k_kernel_stats.csv $P/${R}_kernel_stats_tin_b512_hipgraph.csv
cp $S/kt_q32/k_kernel_stats.csv $P/${R}_kernel_stats_q32_eval_hipgraph.csv
cp $S/mfma_util.json $P/mfma_util.json
cp $S/traffic.json $P/traffic.json
cp $S/pmc_fetch/f_counter_collection.csv $P/${R}_pmc_fetch_size_eager2steps.csv
cp $S/pmc_write/w_counter_collection.csv $P/${R}_pmc_write_size_eager2steps.csv
gzip -c $S/pmc_mfma/m_counter_collection.csv > $P/${R}_pmc_mfma_eager2steps.csv.gz
gzip -c $S/pmc_branch/m_counter_collection.csv > $P/${R}_pmc_branch_mfma.csv.gz
grep -h "NCCL INFO" $S/bench_force_ddp_nccl_info.log | grep -vi "Channel [0-9]\|Trees \[" | cut -c1-300 > $P/${R}_rccl_one_rank_info.txt || true
[ -f $S/kt_ddp/k_kernel_stats.csv ] && cp $S/kt_ddp/k_kernel_stats.csv $P/${R}_kernel_stats_force_ddp_hipgraph.csv
