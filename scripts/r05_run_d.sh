#!/bin/bash
# Round 5, GPU call D: what bounds maze3d at 64x64? counters for the general kernel and the SMALL instantiation
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05d; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
P="python scripts/probe_maze3d_64.py 64"
for mode in small general; do
  if [ $mode = general ]; then export MG_MAZE3D_NO_SMALL=1; else unset MG_MAZE3D_NO_SMALL; fi
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU \
        --output-format csv -d $OUT/pmc_${mode}_a -o m -- $P > $OUT/pmc_${mode}_a.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS \
        --output-format csv -d $OUT/pmc_${mode}_b -o m -- $P > $OUT/pmc_${mode}_b.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH \
        --output-format csv -d $OUT/pmc_${mode}_c -o m -- $P > $OUT/pmc_${mode}_c.log 2>&1
  timeout 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum \
        --output-format csv -d $OUT/pmc_${mode}_d -o m -- $P > $OUT/pmc_${mode}_d.log 2>&1
done
unset MG_MAZE3D_NO_SMALL
python - <<'PY'
import csv, glob, collections, os
out="gpurun_out/r05d"
for d in sorted(glob.glob(out+"/pmc_*_?")):
    agg=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "maze3d_step" in r.get("Kernel_Name",""):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: (sum(v)/len(v), len(v)) for k,v in agg.items()})
PY
