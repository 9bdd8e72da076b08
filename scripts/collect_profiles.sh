#!/bin/bash
# Collects the judged evidence on a GPU box (run from the repo root through gpurun):
#   bench JSON lines, rocprofv3 kernel stats of the default bench command, PMC passes (traffic, SQ counters)
#   for the dominant kernel, the MFMA shape lab.  Everything lands in gpurun_out/prof/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_default_n1e7_b2097152_mixed.json 2> $OUT/bench_default.err
python bench.py --force-type 0 --batch 262144 --cpu-seconds 0 --no-e2e > $OUT/bench_type0_n1e7_b262144.json 2>/dev/null
HVS_I8_SHAPE=32 python bench.py --cpu-seconds 0 --no-e2e > $OUT/bench_i8_32x32x32_n1e7_b2097152_mixed.json 2>/dev/null
python bench.py --engine 2 --cpu-seconds 0 --no-e2e > $OUT/bench_bf16_filter_n1e7_b2097152_mixed.json 2>/dev/null
python bench.py --engine 1 --batch 16384 --cpu-seconds 0 --no-e2e > $OUT/bench_exact_engine_n1e7_b16384_mixed.json 2>/dev/null
python bench.py --n 1000000 --batch 10000 --force-type 0 --steps 5 --warmup 2 --cpu-seconds 0 > $OUT/bench_config1_n1e6_q1e4_type0.json 2>/dev/null
python bench.py --n 1000000 --batch 10000 --steps 5 --warmup 2 --cpu-seconds 0 > $OUT/bench_config2_n1e6_q1e4_mixed.json 2>/dev/null
python bench.py --n 100000000 --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/bench_n1e8_b2097152_mixed.json 2>/dev/null
python bench.py --force-dist --cpu-seconds 0 --no-e2e > $OUT/bench_force_dist_rehearsal.json 2>/dev/null
echo "bench lines done"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write /tmp/p_sq
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $ROOT/bench.py --cpu-seconds 0 --no-e2e > $OUT/bench_under_rocprof.json 2>/dev/null
python3 $ROOT/scripts/prof_summary.py /tmp/p_stats > $OUT/summary.txt
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find /tmp/p_stats -name "*domain_stats.csv" | head -1) $OUT/domain_stats.csv 2>/dev/null || true
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-e2e > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-e2e > /dev/null 2>&1
echo "traffic passes done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/p_sq -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-e2e > /dev/null 2>&1 || echo "sq pass failed"
echo "sq pass done"
python3 $ROOT/scripts/pmc_summary.py /tmp/p_fetch /tmp/p_write /tmp/p_sq $OUT
cd $ROOT
./scripts/shape_lab.out > $OUT/mfma_shape_lab.txt 2>&1 || true
echo "lab done"
ls -la $OUT
