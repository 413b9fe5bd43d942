#!/bin/bash
# Round 5, GPU call E: maze3d SMALL with the column records parked in LDS (with / without the deferred store)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_maze_gpu.py tests/test_mixed_gpu.py -m gpu -x -q > $OUT/pytest_maze.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_maze.log
METAGYM_HIP_LIB=metagym_amd/lib/variants/maze_small_nopipe.so timeout 900 python -m pytest tests/test_maze_gpu.py -m gpu -x -q -k "small_frame or batch_matches or ragged" > $OUT/pytest_maze_nopipe.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_maze_nopipe.log
for r in 64 32; do
  echo "== SMALL (records in LDS) + deferred store $r"; timeout 120 python scripts/probe_maze3d_64.py $r
  echo "== SMALL (records in LDS), direct store $r"; METAGYM_HIP_LIB=metagym_amd/lib/variants/maze_small_nopipe.so timeout 120 python scripts/probe_maze3d_64.py $r
  echo "== general kernel (MG_MAZE3D_NO_SMALL=1) $r"; MG_MAZE3D_NO_SMALL=1 timeout 120 python scripts/probe_maze3d_64.py $r
done > $OUT/maze3d_small_frames.txt 2>&1
P="python scripts/probe_maze3d_64.py 64"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS \
        --output-format csv -d $OUT/pmc_small_lds -o m -- $P > $OUT/pmc_small_lds.log 2>&1
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r05e/pmc_small_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "maze3d_step" in r.get("Kernel_Name",""): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: sum(v)/len(v) for k,v in agg.items()})
PY
tail -2 $OUT/pytest_maze.log; tail -2 $OUT/pytest_maze_nopipe.log; grep -v amdgpu.ids $OUT/maze3d_small_frames.txt
