#!/bin/bash
# Wider PMC sweep of one bench command (GPU box, through gpurun):  scripts/pmc_probe.sh NAME KERNEL_SUBSTR -- cmd ...
# Four separate --pmc passes (no trace domains), per-kernel averages printed as JSON and saved to gpurun_out/pmc_NAME.json.
export TMPDIR=/tmp
name=$1; kern=$2; shift 3
out=gpurun_out/pmc_$name; mkdir -p $out
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TD_TD_BUSY_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_ACCESSES_sum SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- "$@" > $out/p$i.log 2>&1
done
python3 - "$out" "$kern" "$name" <<'PY'
import csv, glob, json, sys, collections
out, kern, name = sys.argv[1:4]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
res["launches_seen"] = max((len(v) for v in acc.values()), default=0)
json.dump(res, open("gpurun_out/pmc_%s.json" % name, "w"), indent=1)
print(json.dumps(res))
PY
