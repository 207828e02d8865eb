for i in 1 2; do
for f in "" "--no-events"; do
 timeout -k 10 120 python bench.py --steps 100 --warmup 5 --cpu-frames 0 --no-secondary $f 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"
done; done
