#!/bin/bash
# Round 6, VERDICT items 3 and 7 (GPU box, through gpurun): uint8 frames with the packed store against the byte stores and against
# int32 frames, and Continuous-3D against Discrete-3D at 256 x 256 — kernel trace, then SQ and texture-path counters in SEPARATE passes.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06/maze_pmc
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -o "TA_[A-Z_0-9a-z]*\|TCP_[A-Z_0-9a-z]*\|TD_[A-Z_0-9a-z]*" $OUT/counters_list.txt | sort -u > $OUT/counters_ta_tcp.txt
M="python scripts/bench_maze.py --skip2d --res 256 --steps 20 --warmup 3"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
SQ2="SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
for V in discrete continuous; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${V}_trace -o m -- $M --no-u8 --only $V > $OUT/${V}_trace.log 2>&1
  timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $OUT/${V}_sq -o m -- $M --no-u8 --only $V > $OUT/${V}_sq.log 2>&1
  timeout 300 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/${V}_sq2 -o m -- $M --no-u8 --only $V > $OUT/${V}_sq2.log 2>&1
  timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum \
          --output-format csv -d $OUT/${V}_ta -o m -- $M --no-u8 --only $V > $OUT/${V}_ta.log 2>&1
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${V}_tcc -o m -- $M --no-u8 --only $V > $OUT/${V}_tcc.log 2>&1
done
# uint8: byte stores (default) and the packed store (MG_MAZE3D_U8_PACKED=1); bench_maze prints int32 first, then uint8 — both kernels are in each trace
MG_MAZE3D_U8_PACKED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/u8_packed_trace -o m -- $M --only discrete > $OUT/u8_packed_trace.log 2>&1
MG_MAZE3D_U8_PACKED=1 timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $OUT/u8_packed_sq -o m -- $M --only discrete > $OUT/u8_packed_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/u8_bytes_trace -o m -- $M --only discrete > $OUT/u8_bytes_trace.log 2>&1
timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $OUT/u8_bytes_sq -o m -- $M --only discrete > $OUT/u8_bytes_sq.log 2>&1
python scripts/bench_maze.py --skip2d --res 256 --only discrete > $OUT/bench_u8_bytes_256.jsonl 2>/dev/null
python scripts/bench_maze.py --skip2d --res 64 --envs 65536 --only discrete > $OUT/bench_u8_bytes_64.jsonl 2>/dev/null
MG_MAZE3D_U8_PACKED=1 python scripts/bench_maze.py --skip2d --res 256 --only discrete > $OUT/bench_u8_packed_256.jsonl 2>/dev/null
MG_MAZE3D_U8_PACKED=1 python scripts/bench_maze.py --skip2d --res 64 --envs 65536 --only discrete > $OUT/bench_u8_packed_64.jsonl 2>/dev/null
# keep what travels back small: the per-dispatch counter CSVs and the stats, not the traces' event dumps
find $OUT -name "*.csv" -size +3M -delete
du -sh $OUT; ls $OUT
