#!/bin/bash
# Same-box A/B of one kernel switch on the training bench: alternates the two settings (boxes of the pool differ by 2-3 %,
# so only runs taken in ONE call are comparable).  usage (on the GPU box): tools/ab_bench.sh SNWS_PRIO 0 1 [extra bench flags]
#   -> gpurun_out/ab_<switch>/<value>_<k>.json and a summary line per run (images/s, ms/step, TFLOP/s of the 3x3 kernels)
set -e
sw=$1; a=$2; b=$3; shift 3
out=gpurun_out/ab_$sw
mkdir -p $out
for k in 1 2; do
  for v in $a $b; do
    env $sw=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 "$@" > $out/${v}_$k.json 2> $out/${v}_$k.err
    python - $out/${v}_$k.json $sw $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d.get("kernels", {})
conv = {n.replace("conv_mfma_", ""): round(v.get("TFLOP/s", 0)) for n, v in k.items() if n.startswith("conv_mfma") and "k3" in n}
print("%s=%s  %.1f images/s  %.2f ms/step  %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], conv), flush=True)
PY
  done
done
