export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-.}"
bash scripts/fuzz_round.sh r05 61 > gpurun_out/fuzz_r05_61.log 2>&1
bash scripts/fuzz_round.sh r05b 62 > gpurun_out/fuzz_r05_62.log 2>&1
for i in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-pmc; done > gpurun_out/r05/bench_20steps_spin_x3.jsonl 2>/dev/null
tail -n 3 gpurun_out/r05/fuzz_*.txt gpurun_out/r05b/fuzz_*.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r05/bench_20steps_spin_x3.jsonl
