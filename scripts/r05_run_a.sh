#!/bin/bash
# Round 5, GPU call A: the new tests + the driver's bench command (reference CPU baseline from oracle/_ref) + launch-shape sweep
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05a; mkdir -p $OUT
ls oracle/_ref/metagym/quadrotor > $OUT/ref_listing.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err
for v in default quad_b64 quad_b128 quad_b512; do
  if [ $v = default ]; then timeout 120 python scripts/quad_variants.py 0.01; else METAGYM_HIP_LIB=metagym_amd/lib/variants/$v.so timeout 120 python scripts/quad_variants.py 0.01; fi
done > $OUT/quad_launch_shapes.txt 2>&1
timeout 120 python scripts/probe_maze3d_64.py 64 > $OUT/maze3d_64_baseline.txt 2>&1
timeout 200 python scripts/bench_walker.py humanoid > $OUT/bench_walker.jsonl 2> $OUT/bench_walker.err
tail -5 $OUT/pytest_gpu.log; cat $OUT/quad_launch_shapes.txt $OUT/maze3d_64_baseline.txt
