# scripts/ab.py under several builds of the library (scripts/build_variant.sh): scripts/lib_ab.sh "NAME:CFG NAME:CFG ..."  (NAME "default" = the in-tree build)
for item in $1; do
  name=${item%%:*}; cfg=${item#*:}
  lib=soap_amd/lib/variants/libsoap_engine_$name.so; [ "$name" = default ] && lib=soap_amd/lib/libsoap_engine.so
  echo -n "$name "; SOAP_ENGINE_LIB=$PWD/$lib REPS=${REPS:-3} timeout -k 10 200 python scripts/ab.py "$cfg" 2>&1 | grep -v amdgpu.ids | cut -c1-${CUT:-230}
done
