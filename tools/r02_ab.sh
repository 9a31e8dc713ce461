#!/bin/bash
# one A/B point of the ipm_kernel work: GPU parity subset, PMC traffic summary, short bench.   usage: r02_ab.sh <tag> [full]
TAG=$1; ROOT=$PWD; OUT=$ROOT/gpurun_out/ab_$TAG; mkdir -p $OUT
if [ "$2" == "full" ]; then SEL=""; else SEL='-k "twin or scvx or batch256 or stream or rocket2d"'; fi
eval timeout -k 5 600 python -m pytest tests -m gpu -x -q $SEL > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
bash tools/pmc_hbm.sh $TAG 4096 > $OUT/pmc.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/pmc_$TAG/summary.json"))
print("bytes/inst-iter %.3e (raw %.3e)  per solve %.3e  L2 hit %.3f  MFMA/iter %.0f" % (d["ipm_bytes_per_instance_iteration"], d["ipm_bytes_per_instance_iteration_raw"], d["ipm_bytes_per_instance_solve"], d.get("ipm_l2_hit_rate",0), d["ipm_mfma"]["SQ_INSTS_MFMA"]/d["ipm_iterations"]))
k=d["kernels"]["ipm_kernel"]; print("fetch %.3e write %.3e per iter" % (k["fetch_bytes"]/d["ipm_iterations"], k["write_bytes"]/d["ipm_iterations"]))
sq=d["sq"]["ipm_kernel"]; wc=sq["SQ_WAVE_CYCLES"]; print({c: round(v/wc,3) for c,v in sq.items()})
PY
timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('$TAG value', d['value'], 'ms/step', d['ms_per_step'], 'ipm avg ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'ipm its', d['config']['mean_ipm_iterations_per_trajectory'])"
