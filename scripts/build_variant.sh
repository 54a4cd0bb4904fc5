#!/bin/bash
# Another build of the engine library for an experiment: scripts/build_variant.sh NAME UNIT "FLAGS" [UNIT "FLAGS" ...]
# recompiles the named translation units (e.g. halo_256n) with extra -D flags and links them with the objects of the
# default build into soap_amd/lib/variants/libsoap_engine_NAME.so; run with SOAP_ENGINE_LIB=<that path>.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
OUT=soap_amd/lib/variants; mkdir -p $OUT/obj_$NAME
python3 - "$NAME" "$@" <<'PY'
import os, subprocess, sys
sys.path.insert(0, ".")
import __graft_entry__ as g
name, rest = sys.argv[1], sys.argv[2:]
over = {rest[i]: rest[i + 1].split() for i in range(0, len(rest), 2)}
g.build()
objs = []
for unit, src, defs, extra in g.UNITS:
    obj = os.path.join("soap_amd/lib/obj", unit + ".o")
    if unit in over:
        obj = os.path.join("soap_amd/lib/variants", "obj_" + name, unit + ".o")
        cmd = [g.HIPCC] + g.HIP_FLAGS + defs + over[unit] + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(g.CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-4000:])
        keep = [l.split("remark: ")[1].split(" [-R")[0] for l in r.stderr.splitlines() if any(k in l for k in ("Function Name", " VGPRs:", "Spill", "ScratchSize", "Occupancy", "LDS Size"))]
        print(unit, over[unit], "|", "; ".join(k.strip() for k in keep if "k_halo" in k or "Function" not in k)[:600])
    objs.append(obj)
out = f"soap_amd/lib/variants/libsoap_engine_{name}.so"
subprocess.run([g.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print("built", out)
PY
