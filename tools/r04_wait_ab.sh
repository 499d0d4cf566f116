cd $GRAFT_REPO_ROOT
for rep in 1 2; do for L in "" tools/bin/libts2d_spinwait.so; do
TS2D_LIBRARY_PATH=${L:+$GRAFT_REPO_ROOT/$L} python - <<PY
import json, os, resource, subprocess, sys, time
t0 = time.time()
p = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "300", "--warmup", "5"], capture_output=True, text=True)
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
j = json.loads(p.stdout.strip().splitlines()[-1])
print("${L:-blocking(product)}".split("/")[-1], j["ms_per_step"], j["config"]["host_step_ms"], j["config"]["device_step_ms"], "cpu user+sys %.2f s of %.2f s wall" % (ru.ru_utime + ru.ru_stime, time.time() - t0))
PY
done; done
