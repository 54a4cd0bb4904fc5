# bench.py (short) under alternating settings of one environment variable: scripts/bench_ab.sh VAR "v1 v2 v1 v2" [bench args]   (STEPS / WARMUP: default 4 / 2)
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v timeout -k 10 600 python bench.py --steps ${STEPS:-4} --warmup ${WARMUP:-2} --no-cpu-baseline --no-full-set "$@" 2>/dev/null | V="$VAR=$v" python -c "
import json, os, sys
d = json.loads(sys.stdin.read())
print(os.environ['V'], 'ms_per_pass %.2f' % d['ms_per_pass'], {k: round(v, 2) for k, v in d['kernel_ms'].items() if k in ('halo', 'mesh', 'reorder')}, 'value %.3g' % d['value'], 'guesses', d.get('halos_with_a_radius_guess'))
"
done
