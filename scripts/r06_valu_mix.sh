#!/bin/bash
# Round 6: the dynamic VALU instruction MIX of the three main kernels (rocprofv3 --pmc, class counters, two passes each; GPU box
# through gpurun). With scripts/ubench/f64_rates.hip's issue costs this gives each kernel's VALU issue-cycle budget.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06/valu_mix
mkdir -p $OUT
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
B="SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_WAVE_CYCLES"
run() {  # name, command...
  name=$1; shift
  timeout 300 rocprofv3 --pmc $A --output-format csv -d $OUT/${name}_a -o x -- "$@" > $OUT/${name}_a.log 2>&1
  timeout 300 rocprofv3 --pmc $B --output-format csv -d $OUT/${name}_b -o x -- "$@" > $OUT/${name}_b.log 2>&1
}
run quad python bench.py --no-cpu-baseline --no-secondary --no-pmc --steps 40 --warmup 5 --launch eager
run maze python scripts/bench_maze.py --skip2d --no-u8 --res 256 --steps 12 --warmup 3 --only discrete
run maze64 python scripts/bench_maze.py --skip2d --no-u8 --res 64 --envs 65536 --steps 12 --warmup 3 --only discrete
run walker python scripts/bench_walker.py humanoid
run walker_grounded python scripts/bench_walker.py humanoid --grounded
find $OUT -name "*.csv" -size +4M -delete
ls $OUT
