# smoke() + five consecutive fresh processes of the driver's bench command (quadrant masks on): one line each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3 4 5; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; k=j.get('kernels_avg_ms') or c.get('kernels_avg_ms')
print(json.dumps({'lease':'f','sequence':'quadrant masks on (commit of the round\'s last kernel change)','variant':'default','ms_per_step':j['ms_per_step'],'mpix_s':j['value'],'host_step_ms':c['host_step_ms'],'device_step_ms':c['device_step_ms'],'gpu_idle_ms_per_step':c['gpu_idle_ms_per_step'],'render_fwd':k['render_fwd'],'render_bwd':k['render_bwd'],'emit_keys':k['emit_keys']}))"
done
