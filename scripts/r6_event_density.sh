cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
for ev in "--event-every 7" "--event-every 13" "--event-every 19" "--no-events"; do
  h=$(python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --min-seconds 1.0 $ev 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['roofline'].get('launches_timed'))")
  echo "round $r [$ev]: $h"
done; done
