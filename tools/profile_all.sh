#!/bin/bash
# One pass over everything profiles/ holds for a round (run on the GPU box through gpurun):
#   tools/profile_all.sh r02          -> gpurun_out/<tag>_* ; copy what should be judged into profiles/
# rocprofv3 runs the program directly (python3 ...), never through env / bash -c; counter passes are separate runs.
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o k -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python3 tools/kfamily.py $OUT/kt/k_kernel_stats.csv 31 > $OUT/family_summary.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 tools/prof_step.py 1024 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 tools/prof_step.py 1024 2 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- python3 tools/prof_step.py 1024 2 > $OUT/pmc_mfma.log 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv $OUT/traffic.json > $OUT/traffic.txt
python3 tools/pmc_mfma.py $OUT/pmc_mfma/m_counter_collection.csv $OUT/mfma_util.json 1 > $OUT/mfma_util.txt
# BASELINE config 5's model (Tiny-ImageNet 64x64), single GPU: per-GPU batch 128 (the reference default) and 512
python3 bench.py --config tin --batch 128 --no-cpu-baseline > $OUT/bench_tin_b128.json 2> $OUT/bench_tin_b128.err
python3 bench.py --config tin --batch 512 --no-cpu-baseline > $OUT/bench_tin_b512.json 2> $OUT/bench_tin_b512.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_tin -o k -- python3 bench.py --config tin --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/bench_tin_under_rocprof.json 2> $OUT/bench_tin_under_rocprof.err
python3 tools/kfamily.py $OUT/kt_tin/k_kernel_stats.csv 28 > $OUT/family_summary_tin_b128.txt
rm -f $OUT/kt/k_kernel_trace.csv $OUT/kt_tin/k_kernel_trace.csv
tail -n 20 $OUT/family_summary.txt $OUT/mfma_util.txt $OUT/traffic.txt
