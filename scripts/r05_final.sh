#!/bin/bash
# Round 5, final validation on a fresh box: what the driver runs (smoke, pytest -m gpu, bench at --steps 20 and at the defaults), the examples
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05final; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err ) 2> $OUT/bench_20steps.time
python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python examples/random_rollouts.py > $OUT/examples.log 2>&1; echo "examples rc $?" >> $OUT/examples.log
tail -3 $OUT/smoke.log; tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/examples.log; cat $OUT/bench_20steps.time; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_20steps.json $OUT/bench.json
