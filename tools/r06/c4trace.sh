#!/bin/bash
# kernel-trace durations of configs[4] frozen in motion for laboratory builds:  tools/r06/c4trace.sh NAME [NAME ...]
cd "$(dirname "$0")/../.."
R=$PWD
for v in "$@"; do
O=gpurun_out/c4trace_$v; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && SMR_LIB=$R/smelter_amd/variants/libsmr_hip.$v.so timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/tools/r06/c4probe.py 200 0.5 > $R/$O/trace.log 2>&1)
echo "== $v"
python - $O <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[(r['Kernel_Name'][:48], r['Grid_Size_X'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(d.items(), key=lambda x: -sum(x[1])):
    if len(v) < 50: continue
    v2 = sorted(v)
    print('  ', k, 'n', len(v), 'median us', v2[len(v2)//2] / 1e3, 'min', v2[0] / 1e3)
PY
find $O/trace -name "*kernel_trace.csv" -delete
done
