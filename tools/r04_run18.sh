# round 4, run 18: the converter's workgroups walked XCD by XCD (neighbours in x and y on the same L2)
mkdir -p gpurun_out/r04_18
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c2', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
for c in 3 4 1; do
timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --no-target --no-long --steps 40 --warmup 10 --latency-frames 5 --inflight 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --no-target --no-long --steps 40 --warmup 10 --latency-frames 5 --inflight 1 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
for tag,d in (('FETCH_SIZE','/tmp/pf'),('WRITE_SIZE','/tmp/pw')):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name')==tag: acc[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if len(v)>20: print(tag,k,len(v),'avg KB',round(sum(v)/len(v),1))
PY
