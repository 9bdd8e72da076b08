#!/bin/bash
# Collects the judged evidence on a GPU box (run from the repo root through gpurun):
#   bench JSON lines, rocprofv3 kernel stats of the default bench command, PMC passes (traffic, SQ counters)
#   for the dominant kernel, the MFMA shape lab.  Everything lands in gpurun_out/prof/.
# Usage: scripts/collect_profiles.sh [part ...]   parts: bench stats pmc lab (default: all)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd $ROOT
PARTS="${*:-bench stats pmc lab}"
B="python bench.py --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12"
if [[ " $PARTS " == *" bench "* ]]; then
python bench.py --steps 20 --warmup 5 > $OUT/bench_default_n1e7_b2097152_mixed_steps20.json 2> $OUT/bench_default.err
for t in 0 1 2 3; do $B --force-type $t > $OUT/bench_type${t}_n1e7_b2097152.json 2>/dev/null; done
$B --force-type 0 --batch 262144 > $OUT/bench_type0_n1e7_b262144.json 2>/dev/null
HVS_I8_SHAPE=32 $B > $OUT/bench_i8_32x32x32_n1e7_b2097152_mixed.json 2>/dev/null
$B --engine 2 > $OUT/bench_bf16_filter_n1e7_b2097152_mixed.json 2>/dev/null
$B --engine 4 > $OUT/bench_f16_filter_n1e7_b2097152_mixed.json 2>/dev/null
$B --engine 1 --batch 16384 > $OUT/bench_exact_engine_n1e7_b16384_mixed.json 2>/dev/null
$B --engine 1 --batch 16384 --force-type 0 > $OUT/bench_exact_engine_n1e7_b16384_type0.json 2>/dev/null
for p in 2 3 4 5; do $B --profile $p > $OUT/bench_profile${p}_n1e7_b2097152_mixed.json 2>/dev/null; done
python bench.py --only-configs12 > $OUT/bench_configs12.json 2>/dev/null
python bench.py --n 100000000 --steps 2 --warmup 1 --cpu-seconds 0 --no-configs12 > $OUT/bench_n1e8_b2097152_mixed.json 2>/dev/null
python bench.py --force-dist --steps 4 --warmup 1 --cpu-seconds 0 --no-e2e --no-configs12 --in-library --in-library-devices 0,0 > $OUT/bench_force_dist_rehearsal.json 2>/dev/null
HVS_LANES=0 $B > $OUT/bench_one_lane.json 2>/dev/null
$B --per-step-calls > $OUT/bench_per_step_calls.json 2>/dev/null
echo "bench lines done"
fi
cd /tmp && export TMPDIR=/tmp
if [[ " $PARTS " == *" stats "* ]]; then
rm -rf /tmp/p_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $ROOT/bench.py --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > $OUT/bench_under_rocprof.json 2>/dev/null
python3 $ROOT/scripts/prof_summary.py /tmp/p_stats > $OUT/summary.txt
python3 $ROOT/scripts/timeline.py /tmp/p_stats 48 > $OUT/timeline_last_step.txt
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find /tmp/p_stats -name "*domain_stats.csv" | head -1) $OUT/domain_stats.csv 2>/dev/null || true
echo "kernel stats done"
fi
if [[ " $PARTS " == *" pmc "* ]]; then
rm -rf /tmp/p_fetch /tmp/p_write /tmp/p_sq
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > /dev/null 2>&1
echo "traffic passes done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/p_sq -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > /dev/null 2>&1 || echo "sq pass failed"
echo "sq pass done"
python3 $ROOT/scripts/pmc_summary.py /tmp/p_fetch /tmp/p_write /tmp/p_sq $OUT
python3 $ROOT/scripts/rescore_traffic.py /tmp/p_fetch > $OUT/rescore_traffic_per_level.txt 2>&1 || true
fi
cd $ROOT
if [[ " $PARTS " == *" lab "* ]]; then
./scripts/shape_lab.out > $OUT/mfma_shape_lab.txt 2>&1 || true
./scripts/shape_lab.out bits > $OUT/operand_bits_lab.txt 2>&1 || true
./scripts/h16_lab.out > $OUT/h16_shape_lab.txt 2>&1 || true   # hipcc --offload-arch=gfx950 -O3 scripts/h16_shape_lab.hip -o scripts/h16_lab.out
./scripts/shape_lab_fold.out bits > $OUT/operand_bits_lab_folded_thresholds.txt 2>&1 || true
echo "lab done"
fi
ls -la $OUT
