# usage: bash tools/ab_env.sh VAR v1 v2 [v1 v2 ...]  -- same-box A/B of one environment variable inside the full training step
# (e.g. PSWIN_DISABLE "" grouped_wgrad "" grouped_wgrad: ops.py feature names)
VAR=$1; shift
for v in "$@"; do env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', d['value'], 'pano/s', d['ms_per_step'], 'ms')
"; done
