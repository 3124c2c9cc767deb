#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/c4probe; mkdir -p $O
R=$PWD
L=$PWD/smelter_amd/variants/libsmr_hip.lab0.so
for ab in 0 8192 4096 16384 12288 28672; do
  SMR_LIB=$L SMR_ABLATE=$ab timeout 300 python tools/r06/c4probe.py 200 0.5 2>&1 | tail -1
done
(cd /tmp && export TMPDIR=/tmp && SMR_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o t -- python $R/tools/r06/c4probe.py 200 0.5 > $R/$O/trace.log 2>&1)
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/c4probe/trace/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[(r['Kernel_Name'][:60], r['Grid_Size_X'], r['Workgroup_Size_X'], r.get('VGPR_Count'), r.get('LDS_Block_Size'))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(d.items(), key=lambda x: -sum(x[1])):
    v2 = sorted(v)
    print(k, 'n', len(v), 'median us', v2[len(v2)//2] / 1e3, 'min', v2[0] / 1e3)
PY
find $O/trace -name "*kernel_trace.csv" -delete
