#!/bin/bash
# each thread-group class alone through the first pipeline iteration: stage times (diagnostic)
for k in 0 1 2; do
  SOAP_ENGINE_PIPE_ONE_ITER=1 SOAP_ENGINE_PIPE_ONLY_CLASS=$k SOAP_ENGINE_PIPELINE=1 timeout -k 10 400 python bench.py --workload ${WL:-c3} --steps 1 --warmup 1 --passes-per-step 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/pc_$k.json
  python - $k <<'PY'
import json,sys
k=sys.argv[1]
d=json.load(open(f"/tmp/pc_{k}.json"))
m=d["kernel_ms"]
print("class", k, "build %.2f a %.2f props %.2f" % (m["pipe_build"], m["pipe_stage_a"], m["pipe_stage_b"]), d["halos_per_class"], d["halos_by_path"])
PY
done
