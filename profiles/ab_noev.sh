# usage: bash profiles/ab_noev.sh "<bench args>" -- QPS without the per-stage events in the timed region
ARGS=$1; shift
env BENCH_NO_EVENTS=1 "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('noev $ARGS $*', round(d['value']), round(d['ms_per_step'],3))"
