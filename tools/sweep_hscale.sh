for h in 1.2 1.5 2.5 3.0 4.0; do
  echo "hscale $h"; KSS_GRID_HSCALE=$h timeout -k 10 100 python bench.py --no-cpu-baseline --brute-steps 0 --steps 10 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['avg_launch_ms'], r['distance_evaluations_per_launch'], d['setup'])" || exit 1
done
