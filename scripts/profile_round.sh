#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  scripts/profile_round.sh r02 [quad|maze|walker|bench ...]
# kernel trace + stats, then PMC counters in SEPARATE passes (no trace domains combined with --pmc),
# everything into gpurun_out/<round>/; scripts/summarize_profiles.py <round> condenses it into profiles/<round>/.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=${1:-r03}; shift
PARTS=${@:-quad north maze walker a1 bench}
OUT=gpurun_out/$R
mkdir -p $OUT
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has quad; then
  # eager launches: one traced dispatch per env.step() (the graph replay runs the same kernel nodes)
  B="python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --launch eager"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quad_trace -o q -- $B > $OUT/quad_trace.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU \
            --output-format csv -d $OUT/quad_pmc_sq -o q -- $B > $OUT/quad_pmc_sq.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/quad_pmc_fetch -o q -- $B > $OUT/quad_pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/quad_pmc_write -o q -- $B > $OUT/quad_pmc_write.log 2>&1
  # the default (hipGraph replay) run under the tracer as well: same kernel, same average expected
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quad_graph_trace -o q -- \
          python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > $OUT/quad_graph_trace.log 2>&1
fi
if has north; then
  # north_star's own batch sizes on one GPU: 2^17 (its per-GPU share on 8 GPUs) and 2^20 (the whole batch), eager launches
  for NN in 131072 1048576; do
    B="python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --launch eager --envs-per-gpu $NN"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quad_${NN}_trace -o q -- $B > $OUT/quad_${NN}_trace.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU \
              --output-format csv -d $OUT/quad_${NN}_pmc_sq -o q -- $B > $OUT/quad_${NN}_pmc_sq.log 2>&1
  done
fi
if has maze; then
  for V in discrete continuous; do
    M="python scripts/bench_maze.py --skip2d --no-u8 --res 256 --steps 30 --warmup 5 --only $V"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/maze_${V}_trace -o m -- $M > $OUT/maze_${V}_trace.log 2>&1
  done
  M="python scripts/bench_maze.py --skip2d --no-u8 --res 256 --steps 30 --warmup 5 --only discrete"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/maze_pmc_fetch -o m -- $M > $OUT/maze_pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/maze_pmc_write -o m -- $M > $OUT/maze_pmc_write.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
            --output-format csv -d $OUT/maze_pmc_sq -o m -- $M > $OUT/maze_pmc_sq.log 2>&1
  python scripts/bench_maze.py > $OUT/bench_maze.jsonl 2> $OUT/bench_maze.err
fi
if has walker; then
  python scripts/bench_walker.py > $OUT/bench_walker.jsonl 2> $OUT/bench_walker.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/walker_trace -o w -- python scripts/bench_walker.py > $OUT/walker_trace.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU \
            --output-format csv -d $OUT/walker_pmc_sq -o w -- python scripts/bench_walker.py humanoid > $OUT/walker_pmc_sq.log 2>&1
  # the contact-rich batch (bench.py C4_grounded_*): everybody lying on the ground
  python scripts/bench_walker.py humanoid --grounded > $OUT/bench_walker_grounded.jsonl 2> $OUT/bench_walker_grounded.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/walker_grounded_trace -o w -- python scripts/bench_walker.py humanoid --grounded > $OUT/walker_grounded_trace.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU \
            --output-format csv -d $OUT/walker_grounded_pmc_sq -o w -- python scripts/bench_walker.py humanoid --grounded > $OUT/walker_grounded_pmc_sq.log 2>&1
fi
if has a1; then
  # every secondary workload of bench.py under the tracer: the a1_* kernels (Quadrupedal actuation / wrappers) among them
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a1_trace -o a -- \
          python bench.py --no-cpu-baseline --steps 50 --warmup 5 > $OUT/a1_trace.log 2>&1
fi
if has bench; then
  python bench.py --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err      # the driver's command
  python bench.py > $OUT/bench.json 2> $OUT/bench.err
  python bench.py --launch eager --no-secondary --no-cpu-baseline > $OUT/bench_eager.json 2> $OUT/bench_eager.err
  BENCH_FORCE_DIST=1 python bench.py --no-secondary --no-cpu-baseline > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err
  BENCH_FORCE_DIST=1 python bench.py --workload mixed --no-cpu-baseline > $OUT/bench_mixed_force_dist.json 2> $OUT/bench_mixed_force_dist.err
fi
ls -R $OUT | head -80
