# usage: bash profiles/ab_one.sh "<bench args>" [ENV=VAL ...]  -- one short bench run, per-kernel launch times
ARGS=$1; shift
env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$ARGS $*', round(d['value']), round(d['ms_per_step'],3), [(k.get('kernel','?')[:22], round(k.get('launch_ms') or 0,3)) for k in d['roofline']['per_kernel']])"
