#!/bin/bash
# Small measurements DESIGN.md quotes, each into its own file (GPU box, through gpurun):  scripts/evidence_round.sh r03
# (compile scripts/ubench/quad_io first:  hipcc --offload-arch=gfx950 -O3 scripts/ubench/quad_io.hip -o scripts/ubench/quad_io)
cd "${GRAFT_REPO_ROOT:-.}"
R=${1:-r03}; OUT=gpurun_out/$R; mkdir -p $OUT
# I/O floor of the quadrotor step's access pattern: variant 0 SoA loads + stores, 2 empty kernel, 3 loads only, 5 stores only, 6 loads + nt stores
scripts/ubench/quad_io > $OUT/quad_io_ubench.txt 2>&1
python scripts/write_ceiling.py > $OUT/write_ceiling.json 2>/dev/null
python scripts/quad_rollout.py 2>/dev/null | grep "^n=" > $OUT/quad_rollout.txt
python scripts/quad_sizes.py 2>/dev/null | grep "^{" > $OUT/quad_sizes.json
cat $OUT/quad_io_ubench.txt $OUT/write_ceiling.json $OUT/quad_rollout.txt $OUT/quad_sizes.json
