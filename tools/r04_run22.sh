# round 4, run 22: what bounds the converter — ablation builds (tools/variant_convert.sh cablN -DCV_ABL=N: 1 no stores | 2 no chroma loads | 4 no luma loads |
# 8 no per-pixel arithmetic), HIP stage timer of the `ingest` stage (converter alone), configs[2] and configs[3]
for name in base cabl1 cabl2 cabl4 cabl6 cabl7 cabl8 cabl9 cabl15; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  for c in 2 3; do
  SMR_LIB=$PWD/$lib timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 100 --warmup 20 --latency-frames 20 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$name c$c ingest', r['kernels']['ingest']['avg_us'], 'us')"
  done
done
