cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_create_lut_gpu.py tests/test_optimize_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -4
python tools/k267_probe.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k:v['ms'] for k,v in d.items()})"
for e in 0 1; do
  if [ $e = 1 ]; then export ECCKD_LBFGS_NO_STEP_MEMORY=1; fi
  python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench
from ecckd_amd import api
ctx = api.Context(0)
for sw in (False, True):
    for its in (40, 300):
        r = bench.lut_opt_bench(ctx, its, sw=sw)
        print("no_step_memory" if os.environ.get("ECCKD_LBFGS_NO_STEP_MEMORY") else "step_memory", "sw" if sw else "lw", its, "it/s %.0f" % r["iters_per_s"], "J0 %.4f J_final %.6f" % (r["J0"], r["J_final"]), "evals/s?", r.get("cost_grad_ms"))
PY
done
unset ECCKD_LBFGS_NO_STEP_MEMORY
python tools/fsck_tools_bench.py > gpurun_out/fsck_tools_bench.json 2> gpurun_out/fsck_tools_bench.err; tail -c 600 gpurun_out/fsck_tools_bench.err; python -c "
import json; d=json.load(open('gpurun_out/fsck_tools_bench.json')); print({k:d[k] for k in d if k!='tool_processes'})"
out=gpurun_out/r04_profiles; mkdir -p $out
rocprofv3 --kernel-trace -d $out/kt -o kt --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-lut-opt --no-sw --no-e2e --no-gas-after-gas --no-single-gas > $out/kt_bench.json 2> $out/kt.err
python3 - > $out/sweep_union.json <<PYEOF
import json, subprocess, glob
line = json.loads(open("$out/kt_bench.json").read().strip().splitlines()[-1])
rl = line["roofline"]
total = rl["points_swept_per_step"] * rl["algorithmic_bytes_per_point"] * (line["steps"] + line["warmup"])
trace = glob.glob("$out/kt/*kernel_trace.csv")[0]
u = json.loads(subprocess.check_output(["python3", "tools/kernel_union.py", trace, "--kernel", "k_rt_lw_bb_mirror", "--total-bytes", str(total)]))
u["bench_line_roofline"] = {k: rl[k] for k in ("achieved", "frac", "avg_launch_ms", "launches", "search_window_ms_per_step", "points_swept_per_step")}
print(json.dumps(u, indent=1))
PYEOF
rm -rf $out/kt
head -40 $out/sweep_union.json
