for r in 1 2; do
for a in "--inflight 4 --queue-depth 2" "--inflight 5 --queue-depth 2" "--inflight 6 --queue-depth 2" "--inflight 4 --queue-depth 3" "--inflight 3 --queue-depth 2" "--inflight 8 --queue-depth 1"; do
python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-extra $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', round(d['value']), round(d['ms_per_step'],3))"
done; done
