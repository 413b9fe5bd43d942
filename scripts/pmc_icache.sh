export TMPDIR=/tmp
mkdir -p gpurun_out/icache
rocprofv3 --list-avail 2>/dev/null | grep -i -o "Name:[[:space:]]*[A-Z_a-z0-9]*\(ICACHE\|IFETCH\|INST_LEVEL\|WAIT_IFETCH\)[A-Z_a-z0-9]*" | sort -u > gpurun_out/icache/avail.txt
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|ifetch" | head -40 >> gpurun_out/icache/avail.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/icache/p$i -o p -- python scripts/bench_walker.py humanoid > gpurun_out/icache/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/icache/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "walker_step_wave" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: sum(v)/len(v) for k, v in acc.items()}))
PY
cat gpurun_out/icache/avail.txt | head -30
tail -3 gpurun_out/icache/p1.log
