#!/bin/bash
# Round 5's evidence from the GPU box in one call (results under gpurun_out/r05_end/, copied into profiles/ by hand):
#   the -m gpu suite on the product build (+ the fused-conversion tests on the laboratory build), smoke(), the default bench line
#   (long loop, CPU baseline, target and one-rank sharded children), the driver's command twice, the other configs, one frame in flight;
#   kernel trace + PMC passes of configs[2] (tools/prof.sh) -> traffic / issue JSON carrying the library's device-code hash;
#   FETCH_SIZE / WRITE_SIZE passes of configs[3]; the converter's wave stamps (timing build); host enqueue rate.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_end
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "from smelter_amd import build as B; print('device code', B.kernels_sha256())" | tee $O/lib_identity.txt
bash tools/gpu_tests.sh r05_end
[ -f smelter_amd/variants/libsmr_hip.lab.so ] && { SMR_LIB=$R/smelter_amd/variants/libsmr_hip.lab.so timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernel_selection.py -m gpu -q > $O/pytest_lab.log 2>&1; echo "lab pytest rc $?"; tail -2 $O/pytest_lab.log; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# counters first: the bench line quotes them only if they carry this library's hash
bash tools/prof.sh r05 --inflight 1 --no-target > $O/prof.log 2>&1
cp gpurun_out/prof_r05/summary.txt $O/rocprofv3_summary.txt; cp gpurun_out/prof_r05/stats/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || find gpurun_out/prof_r05/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cp gpurun_out/prof_r05/traffic.json $O/traffic.json
python tools/issue_json.py $O/rocprofv3_summary.txt > $O/issue.json
cp $O/traffic.json profiles/r05_traffic.json; cp $O/issue.json profiles/r05_issue.json   # (for the bench runs below, on this box)
for G in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/prof_r05_c3/pmc_$G -o p -- python $R/bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline --no-long --latency-frames 5 --inflight 1 > $O/prof_c3_$G.log 2>&1)
done
python tools/traffic_json.py gpurun_out/prof_r05_c3 > $O/traffic_configs3.json; cp $O/traffic_configs3.json profiles/r05_traffic_configs3.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05_c3/stats -o s -- python $R/bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline --no-long --latency-frames 5 --inflight 1 > $O/prof_c3_stats.log 2>&1)
find gpurun_out/prof_r05_c3/stats -name "*kernel_stats.csv" -exec cp {} $O/configs3_kernel_stats.csv \;
find gpurun_out/prof_r05_c3 gpurun_out/prof_r05 -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r05_c3 gpurun_out/prof_r05 -name "*counter_collection.csv" -delete
# the lines
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-target --long-seconds 2 > $O/bench_driver_like_$i.json 2>/dev/null; done
for a in "c1:--config 1" "c3:--config 3" "c4:--config 4" "valu:--ingest valu" "inflight1:--inflight 1" "inflight3:--inflight 3"; do
  n=${a%%:*}; f=${a#*:}
  timeout 300 python bench.py --no-cpu-baseline --no-target --long-seconds 2 $f > $O/bench_$n.json 2>/dev/null
done
timeout 300 python bench.py --force-sharded --no-cpu-baseline --no-target --steps 100 --latency-frames 50 --no-long > $O/bench_sharded_1rank.json 2>$O/bench_sharded.err
[ -f smelter_amd/variants/libsmr_hip.lab.so ] && SMR_LIB=$R/smelter_amd/variants/libsmr_hip.lab.so timeout 300 python bench.py --no-cpu-baseline --no-target --long-seconds 2 --ingest fused > $O/bench_fused_lab.json 2>/dev/null
[ -f smelter_amd/variants/libsmr_hip.timing.so ] && for k in 4 6; do SMR_LIB=$R/smelter_amd/variants/libsmr_hip.timing.so SMR_CONVERT_WG_PER_CU=$k timeout 200 python tools/r05/conv_timing.py > $O/conv_timing_k$k.txt 2>&1; done
timeout 200 python tools/host_rate.py > $O/host_rate.txt 2>&1
rocm-smi --showclocks --showpower > $O/rocm_smi_idle.txt 2>&1
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), r["value"], "fps; long", r.get("value_long",{}).get("frames_per_s"), "; 1 in flight", r["config"].get("frames_per_s_one_in_flight"), {k:v["avg_us"] for k,v in r.get("kernels",{}).items()}, "p50", r.get("latency_ms",{}).get("p50"), "roofline", (r.get("roofline") or {}).get("frac"), "traffic x", (r.get("roofline") or {}).get("traffic_over_algorithmic"))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
du -sh gpurun_out
