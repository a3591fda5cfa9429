# A/B of esm_amd/libesmk.so against a saved build (esm_amd/libesmk.so.base) on one box: bash tools/ab_lib.sh
set -e
B="timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
cp esm_amd/libesmk.so /tmp/new.so
$B > gpurun_out/ab_new1.log 2>&1
cp esm_amd/libesmk.so.base esm_amd/libesmk.so
$B > gpurun_out/ab_base1.log 2>&1
cp /tmp/new.so esm_amd/libesmk.so
$B > gpurun_out/ab_new2.log 2>&1
for f in new1 base1 new2; do python - <<PY
import json
l=[x for x in open("gpurun_out/ab_$f.log") if x.startswith("{")][-1]
d=json.loads(l); print("$f", round(d["ms_per_step"],3), round(d["value"]), round(d["roofline"]["launch_ms"]*1e3,1))
PY
done
