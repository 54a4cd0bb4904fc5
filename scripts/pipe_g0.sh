#!/bin/bash
# groups of the wavefront class next to the large halos' kernels (first pipeline iteration): diagnostic sweep
for g0 in 2048 1536 1280 1024; do
  SOAP_ENGINE_G0=$g0 SOAP_ENGINE_PIPE_ONE_ITER=1 SOAP_ENGINE_PIPELINE=1 timeout -k 10 400 python bench.py --workload ${WL:-c3} --steps 1 --warmup 1 --passes-per-step 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ps.json
  python - "$g0" <<'PY'
import json,sys
d=json.load(open("/tmp/ps.json"))
m=d["kernel_ms"]
print(sys.argv[1], "halo %.2f build %.2f a %.2f props %.2f" % (m["halo"], m["pipe_build"], m["pipe_stage_a"], m["pipe_stage_b"]), [round(x,1) for x in m["pipe_build_by_class"]], [round(x,1) for x in m["pipe_props_by_class"]], d["halos_per_class"])
PY
done
