#!/bin/bash
# Round 6's evidence from the GPU box in one call (results under gpurun_out/r06_end/, copied into profiles/ by hand):
#   the -m gpu suite on the product build (+ the fused-conversion tests on the laboratory build), smoke(), the default bench line
#   (long loop, CPU baseline, target and one-rank sharded children), the driver's command twice, the other configs, one frame in flight;
#   kernel trace + PMC passes of configs[2] (tools/prof.sh) -> traffic / issue JSON carrying the library's device-code hash;
#   FETCH_SIZE / WRITE_SIZE passes of configs[3]; the converter's wave stamps (timing build); host enqueue rate.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_end
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "from smelter_amd import build as B; print('device code', B.kernels_sha256())" | tee $O/lib_identity.txt
bash tools/gpu_tests.sh r06_end
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# counters first: the bench line quotes them only if they carry this library's hash
bash tools/prof.sh r06 --inflight 1 --no-target > $O/prof.log 2>&1
cp gpurun_out/prof_r06/summary.txt $O/rocprofv3_summary.txt; cp gpurun_out/prof_r06/stats/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || find gpurun_out/prof_r06/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cp gpurun_out/prof_r06/traffic.json $O/traffic.json
python tools/issue_json.py $O/rocprofv3_summary.txt > $O/issue.json
cp $O/traffic.json profiles/r06_traffic.json; cp $O/issue.json profiles/r06_issue.json   # (for the bench runs below, on this box)
for G in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/prof_r06_c3/pmc_$G -o p -- python $R/bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline --no-long --latency-frames 5 --inflight 1 > $O/prof_c3_$G.log 2>&1)
done
python tools/traffic_json.py gpurun_out/prof_r06_c3 > $O/traffic_configs3.json; cp $O/traffic_configs3.json profiles/r06_traffic_configs3.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06_c3/stats -o s -- python $R/bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline --no-long --latency-frames 5 --inflight 1 > $O/prof_c3_stats.log 2>&1)
find gpurun_out/prof_r06_c3/stats -name "*kernel_stats.csv" -exec cp {} $O/configs3_kernel_stats.csv \;
find gpurun_out/prof_r06_c3 gpurun_out/prof_r06 -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r06_c3 gpurun_out/prof_r06 -name "*counter_collection.csv" -delete
# the lines
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-target --long-seconds 2 > $O/bench_driver_like_$i.json 2>/dev/null; done
for a in "c1:--config 1" "c3:--config 3" "c4:--config 4" "valu:--ingest valu" "inflight1:--inflight 1" "inflight3:--inflight 3"; do
  n=${a%%:*}; f=${a#*:}
  timeout 300 python bench.py --no-cpu-baseline --no-target --long-seconds 2 $f > $O/bench_$n.json 2>/dev/null
done
timeout 300 python bench.py --force-sharded --no-cpu-baseline --no-target --steps 100 --latency-frames 50 --no-long > $O/bench_sharded_1rank.json 2>$O/bench_sharded.err
# the sharded driver with two ranks as two contexts of this device (code-path evidence; kernel trace -> timeline), the plane-source route with its counters,
# the reference-shaped capacity mode
timeout 300 python bench.py --same-device 2 --steps 100 --warmup 10 > $O/bench_same_device_2.json 2> $O/bench_same_device_2.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r06_sd2 -o t -- python $R/bench.py --same-device 2 --steps 40 --warmup 5 > $O/prof_sd2.log 2>&1)
python tools/timeline.py $(find gpurun_out/prof_r06_sd2 -name "*kernel_trace.csv" | head -1) > $O/timeline_same_device_2.txt 2>&1
find gpurun_out/prof_r06_sd2 -name "*kernel_trace.csv" -delete
timeout 300 python bench.py --no-cpu-baseline --no-target --long-seconds 2 --plane-source > $O/bench_plane_source.json 2>/dev/null
for G in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/prof_r06_ps/pmc_$G -o p -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-long --no-target --latency-frames 5 --inflight 1 --plane-source > $O/prof_ps_$G.log 2>&1)
done
python tools/traffic_json.py gpurun_out/prof_r06_ps > $O/traffic_plane_source.json
find gpurun_out/prof_r06_ps -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r06_ps -name "*counter_collection.csv" -delete
# configs[4] frozen half-way through a transition (tools/r06/c4probe.py): what the kernels of a frame in motion take, from the kernel trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r06_c4m -o t -- python $R/tools/r06/c4probe.py 200 0.5 > $O/prof_c4m.log 2>&1)
python - > $O/configs4_motion_kernels.txt <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_r06_c4m/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[(r["Kernel_Name"][:64], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("configs[4] frozen half-way through a transition, one frame in flight, the same presentation time 200 times (tools/r06/c4probe.py): kernel, workgroups, launches, median / min us")
for k, v in sorted(d.items(), key=lambda x: -sum(x[1])):
    if len(v) < 50: continue
    v2 = sorted(v)
    print("  %-66s %6d  n %4d  %7.2f  %7.2f" % (k[0], k[1], len(v), v2[len(v2) // 2] / 1e3, v2[0] / 1e3))
PY
find gpurun_out/prof_r06_c4m -name "*kernel_trace.csv" -delete
timeout 300 python bench.py --config 4 --inflight 2 --no-cpu-baseline --no-target --long-seconds 2 > $O/bench_c4_inflight2.json 2>/dev/null
timeout 900 python bench.py --mode capacity > $O/bench_capacity.json 2> $O/bench_capacity.err
timeout 200 python tools/host_rate.py > $O/host_rate.txt 2>&1
rocm-smi --showclocks --showpower > $O/rocm_smi_idle.txt 2>&1
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), r["value"], "; long", (r.get("value_long") or {}).get("frames_per_s"), "; 1 in flight", r["config"].get("frames_per_s_one_in_flight"), {k:v["avg_us"] for k,v in (r.get("kernels") or {}).items()}, "p50", (r.get("latency_ms") or {}).get("p50"), "roofline", (r.get("roofline") or {}).get("frac"), "traffic x", (r.get("roofline") or {}).get("traffic_over_algorithmic"), "target", ((r.get("target") or {}).get("frames_per_s")))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
du -sh gpurun_out
