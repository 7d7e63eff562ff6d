set -e
for cfg in "0 0" "17000000 2" "8500000 2" "8500000 4" "4200000 4"; do
  set -- $cfg
  echo "== chunk $1 lanes $2"
  timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --wf-chunk $1 --wf-lanes $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(d['ms_per_step'],3), 'c4', round(d['also'][0]['ms_per_step'],3))"
done
