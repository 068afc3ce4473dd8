#!/bin/bash
# One pass over everything profiles/ holds for a round (run on the GPU box through gpurun):
#   tools/profile_all.sh r03          -> gpurun_out/prof_<tag>/ ; tools/copy_profiles.sh copies what should be judged into profiles/
# rocprofv3 runs the program directly (python3 ...), never through env / bash -c; counter passes are separate runs.
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o k -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python3 tools/kfamily.py $OUT/kt/k_kernel_stats.csv 31 > $OUT/family_summary.txt
# stream overlap and the per-kernel timeline of one replayed step: a trace WITHOUT bench.py's instrumented eager step (its bracketing kernels are not the product)
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_plain -o k -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/bench_under_rocprof_plain.json 2> $OUT/bench_under_rocprof_plain.err
python3 tools/overlap_report.py $OUT/kt_plain/k_kernel_trace.csv > $OUT/overlap.txt
python3 tools/step_timeline.py $OUT/kt_plain/k_kernel_trace.csv 0 100000 > $OUT/step_timeline.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 tools/prof_step.py 1024 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 tools/prof_step.py 1024 2 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- python3 tools/prof_step.py 1024 2 > $OUT/pmc_mfma.log 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv $OUT/traffic.json > $OUT/traffic.txt
python3 tools/pmc_mfma.py $OUT/pmc_mfma/m_counter_collection.csv $OUT/mfma_util.json 1 > $OUT/mfma_util.txt
# the fused attention branch on its own: device time (inference form, training form with saves, forward + backward) and its MFMA counters
python3 tools/bench_branch.py 1024 30 > $OUT/bench_branch.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_branch -o m -- python3 tools/bench_branch.py 1024 2 > $OUT/pmc_branch.log 2>&1
python3 tools/pmc_branch.py $OUT/pmc_branch/m_counter_collection.csv > $OUT/branch_mfma_pmc.txt
# BASELINE config 5's model (Tiny-ImageNet 64x64), single GPU: per-GPU batch 128 (the reference default) and 512
python3 bench.py --config tin --batch 128 --no-cpu-baseline > $OUT/bench_tin_b128.json 2> $OUT/bench_tin_b128.err
python3 bench.py --config tin --batch 512 --no-cpu-baseline > $OUT/bench_tin_b512.json 2> $OUT/bench_tin_b512.err
for B in 128 512; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_tin$B -o k -- python3 bench.py --config tin --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/bench_tin_b${B}_under_rocprof.json 2> $OUT/bench_tin_b${B}_under_rocprof.err
  python3 tools/kfamily.py $OUT/kt_tin$B/k_kernel_stats.csv 28 > $OUT/family_summary_tin_b$B.txt
done
# BASELINE config 2: QA-ViT @32 forward only, eval, B = 512 (v1 = QAViT.py, v2 = QAViTv2.py blocks)
python3 bench.py --config q32 --eval > $OUT/bench_q32_eval.json 2> $OUT/bench_q32_eval.err
python3 bench.py --config q32 --eval --variant v2 --no-cpu-baseline > $OUT/bench_q32_eval_v2.json 2> $OUT/bench_q32_eval_v2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_q32 -o k -- python3 bench.py --config q32 --eval --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/bench_q32_under_rocprof.json 2> $OUT/bench_q32_under_rocprof.err
python3 tools/kfamily.py $OUT/kt_q32/k_kernel_stats.csv 28 > $OUT/family_summary_q32_eval.txt
# the data-parallel step on ONE rank over RCCL (what one GPU can show of the N-GPU step): default sync points, all seven, none
# (librccl prints its version banner on stdout: keep the JSON line only)
QAVIT_FORCE_DDP=1 python3 bench.py --no-cpu-baseline --no-kernel-timing 2> $OUT/bench_force_ddp.err | grep '^{' > $OUT/bench_force_ddp.json
QAVIT_FORCE_DDP=1 QAVIT_DDP_TAGS=all python3 bench.py --no-cpu-baseline --no-kernel-timing 2> $OUT/bench_force_ddp_all_tags.err | grep '^{' > $OUT/bench_force_ddp_all_tags.json
QAVIT_FORCE_DDP=1 QAVIT_DDP_TAGS= python3 bench.py --no-cpu-baseline --no-kernel-timing 2> $OUT/bench_force_ddp_no_tags.err | grep '^{' > $OUT/bench_force_ddp_no_tags.json
QAVIT_FORCE_DDP=1 NCCL_DEBUG=INFO python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-graph > $OUT/bench_force_ddp_nccl_info.log 2>&1
QAVIT_FORCE_DDP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_ddp -o k -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/bench_force_ddp_under_rocprof.log 2>&1
python3 tools/kfamily.py $OUT/kt_ddp/k_kernel_stats.csv 31 > $OUT/family_summary_force_ddp.txt
# the lateral path on its stream (default) against everything on one stream
QAVIT_LATERAL_STREAM=0 python3 bench.py --no-cpu-baseline --no-kernel-timing > $OUT/bench_lateral_one_stream.json 2> $OUT/bench_lateral_one_stream.err
# the one-launch weight-gradient kernel: the step's own problem list in isolation, homogeneous streaming sets per tile class; the fused TokenLearner;
# the step's chain stamps (host-recorded events at the phase boundaries of one eager step)
python3 tools/tn_census.py > $OUT/tn_census.txt 2>&1
python3 tools/bench_tn_stream.py > $OUT/tn_stream.txt 2>&1
python3 tools/bench_tl.py > $OUT/bench_tl.txt 2>&1
python3 tools/chain_stamps.py > $OUT/chain_stamps.txt 2>&1
rm -f $OUT/kt_plain/k_kernel_trace.csv $OUT/kt_ddp/k_kernel_trace.csv $OUT/kt/k_kernel_trace.csv $OUT/kt_tin128/k_kernel_trace.csv $OUT/kt_tin512/k_kernel_trace.csv $OUT/kt_q32/k_kernel_trace.csv
tail -n 20 $OUT/family_summary.txt $OUT/mfma_util.txt $OUT/traffic.txt $OUT/branch_mfma_pmc.txt
