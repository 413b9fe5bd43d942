#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun): kernel trace + stats, then PMC
# counters in separate passes (no trace domains combined with --pmc), all into gpurun_out/r01/.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r01
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quad_trace -o q -- $B > $OUT/quad_trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU \
          --output-format csv -d $OUT/quad_pmc_sq -o q -- $B > $OUT/quad_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/quad_pmc_fetch -o q -- $B > $OUT/quad_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/quad_pmc_write -o q -- $B > $OUT/quad_pmc_write.log 2>&1
M="python scripts/bench_maze.py --skip2d --no-u8 --res 256 --steps 10 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/maze_trace -o m -- $M > $OUT/maze_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/maze_pmc_fetch -o m -- $M > $OUT/maze_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/maze_pmc_write -o m -- $M > $OUT/maze_pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
          --output-format csv -d $OUT/maze_pmc_sq -o m -- $M > $OUT/maze_pmc_sq.log 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python scripts/bench_maze.py > $OUT/bench_maze.jsonl 2> $OUT/bench_maze.err
python scripts/bench_walker.py > $OUT/bench_walker.jsonl 2> $OUT/bench_walker.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/walker_trace -o w -- python scripts/bench_walker.py > $OUT/walker_trace.log 2>&1
W="python scripts/bench_walker.py"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU \
          --output-format csv -d $OUT/walker_pmc_sq -o w -- $W > $OUT/walker_pmc_sq.log 2>&1
ls -R $OUT | head -60
