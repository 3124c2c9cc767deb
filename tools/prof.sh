#!/bin/bash
# GPU-box profiling helper: kernel-trace stats + PMC passes (separate runs, as the guide prescribes).
# usage: tools/prof.sh <tag> [bench args...]   -> writes CSVs under gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH=${PROF_CMD:-"python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-long --latency-frames 5 $*"}  # PROF_CMD: profile another command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
i=0
MAXG=${PROF_GROUPS:-99}
for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY" \
             "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_WAVE32_LDS" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  [ $i -gt $MAXG ] && break
  rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
done
cd $R
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
python tools/traffic_json.py $OUT > $OUT/traffic.json 2>/dev/null   # (with the library's device-code hash: bench.py quotes it only for that library)
# (the per-dispatch traces are large and gpurun_out/ is capped at 64 MiB: keep the summaries)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
cat $OUT/summary.txt
