# does an untimed settle period remove the stalled step from a 20-step timed region? (bench.py --settle-steps)
R=${GRAFT_REPO_ROOT:-/root/repo}
for S in 0 80 0 80 0 80 0 80; do
  timeout 200 python $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 --settle-steps $S 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('settle', $S, j['value'], j['ms_per_step'], j['config']['host_step_ms'], j['config']['settle_steps_untimed'])"
done
