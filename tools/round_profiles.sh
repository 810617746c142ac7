#!/bin/bash
# The measurements a round commits under profiles/: run on the GPU box from the repository root,
#   bash tools/round_profiles.sh r03
# writes gpurun_out/<tag>_profiles/; copy what is to be judged into profiles/ (see profiles/README.md).
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_profiles; mkdir -p $out
quiet="--no-cpu --no-lut-opt --no-sw --no-e2e"
# per-kernel totals of the headline step
rocprofv3 --kernel-trace --stats -d $out/ks -o ks --output-format csv -- python3 bench.py --steps 2 --warmup 1 $quiet > $out/ks_bench.json 2> $out/ks.err
# the same with the side-by-side steps only: the UNION of the time in which a sweep launch was executing (the launches of the six
# streams overlap: launches x average duration is not the device time of the sweep) and its bytes over that time
rocprofv3 --kernel-trace -d $out/kt -o kt --output-format csv -- python3 bench.py --steps 2 --warmup 1 $quiet --no-gas-after-gas --no-single-gas > $out/kt_bench.json 2> $out/kt.err
python3 - > $out/sweep_union.json 2>> $out/kt.err <<PYEOF
import json, subprocess, glob
line = json.loads(open("$out/kt_bench.json").read().strip().splitlines()[-1])
rl = line["roofline"]
# (3 steps ran: warm-up + 2 timed; points swept per step x bytes per point x 3)
total = rl["points_swept_per_step"] * rl["algorithmic_bytes_per_point"] * (line["steps"] + line["warmup"])
trace = glob.glob("$out/kt/*kernel_trace.csv")[0]
u = json.loads(subprocess.check_output(["python3", "tools/kernel_union.py", trace, "--kernel", "k_rt_lw_bb_mirror", "--total-bytes", str(total)]))
u["bench_line_roofline"] = {k: rl[k] for k in ("achieved", "frac", "avg_launch_ms", "launches", "search_window_ms_per_step", "points_swept_per_step")}
print(json.dumps(u, indent=1))
PYEOF
# the same job gas after gas ONLY (no overlap of launches): the per-kernel averages a reader can price with bytes per launch
rocprofv3 --kernel-trace --stats -d $out/ksq -o ksq --output-format csv -- python3 bench.py --steps 2 --warmup 1 $quiet --gases-side-by-side 1 --no-gas-after-gas --no-single-gas > $out/ksq_bench.json 2> $out/ksq.err
cp $out/ksq/ksq_kernel_stats.csv $out/kernel_stats_gas_after_gas.csv 2>/dev/null
# HBM bytes of the dominant kernel: separate --pmc passes, kernel trace only beside them
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 $quiet --no-gas-after-gas --no-single-gas > $out/pmc_$c.json 2> $out/pmc_$c.err
done
python3 tools/traffic_json.py k_rt_lw_bb_mirror $out/pmc_FETCH_SIZE/p_counter_collection.csv $out/pmc_WRITE_SIZE/p_counter_collection.csv $out/pmc_FETCH_SIZE.json > $out/traffic_k_rt_lw_bb.json
# the bench line as the driver runs it; its `traffic` comes from profiles/<tag>_traffic_k_rt_lw_bb.json: this build's, just measured
cp $out/traffic_k_rt_lw_bb.json profiles/${tag}_traffic_k_rt_lw_bb.json
python3 bench.py > $out/bench.json 2> $out/bench.err
# K6 (g-point averaging) and K2 / K7: timings, then the SQ counters and HBM bytes of K6
python3 tools/k267_probe.py > $out/k267_probe.json 2> $out/k267.err
# ... and the kernels' own durations (K2's probe figure includes its call's constants upload and flag read-back)
rocprofv3 --kernel-trace --stats -d $out/k267ks -o k --output-format csv -- python3 tools/k267_probe.py > /dev/null 2> $out/k267ks.err
cp $out/k267ks/k_kernel_stats.csv $out/k267_kernel_stats.csv 2>/dev/null
# the LUT optimisation: HBM bytes per iteration of its kernels (separate counter passes), kernel averages
python3 tools/lut_opt_probe.py --iterations 100 > $out/lut_opt_probe.json 2> $out/lop.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/opt_$c -o p --output-format csv -- python3 tools/lut_opt_probe.py --iterations 100 > $out/opt_$c.json 2> $out/opt_$c.err
done
python3 tools/opt_traffic.py $out/opt_FETCH_SIZE/p_counter_collection.csv $out/opt_WRITE_SIZE/p_counter_collection.csv $out/opt_FETCH_SIZE.json > $out/traffic_lut_opt.json 2>> $out/lop.err
cp $out/traffic_lut_opt.json profiles/${tag}_traffic_lut_opt.json
rocprofv3 --kernel-trace --stats -d $out/optks -o k --output-format csv -- python3 tools/lut_opt_probe.py --iterations 300 > /dev/null 2> $out/optks.err
cp $out/optks/k_kernel_stats.csv $out/opt_kernel_stats.csv 2>/dev/null
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $set | tr " " "_")
  rocprofv3 --kernel-trace --pmc $set -d $out/k6_$n -o p --output-format csv -- python3 tools/k267_probe.py > /dev/null 2> $out/k6_$n.err
  python3 tools/pmc_summary.py k_gavg_partial $out/k6_$n/p_counter_collection.csv >> $out/pmc_k6.txt
done
# the sweep kernel alone on fixed batches (FLOAT pairs / DOUBLE rows), and where a search batch's time goes (rocpd database of
# one step's kernel trace)
python3 tools/sweep_probe.py > $out/sweep_probe.json 2> $out/sp.err
ECCKD_BG64=1 python3 tools/sweep_probe.py > $out/sweep_probe_double_rows.json 2>> $out/sp.err
rocprofv3 --kernel-trace -d $out/bb -o bb --output-format rocpd -- python3 bench.py --steps 1 --warmup 0 $quiet > /dev/null 2> $out/bb.err
python3 tools/batch_breakdown.py $(ls $out/bb/*.db | head -1) > $out/batch_breakdown.txt 2>> $out/bb.err
# the searches of the six prepared gases: gas after gas / side by side; the job by the tools on files
python3 tools/gases_probe.py --widths 1,6,3,1 --out $out/gases_probe.json > $out/gases_probe.log 2>&1
python3 tools/fsck_tools_bench.py > $out/fsck_tools_bench.json 2> $out/ftb.err
python3 tools/fsck_tools_bench.py --netcdf4 > $out/fsck_tools_bench_netcdf4.json 2>> $out/ftb.err
python3 tools/sort_probe.py > $out/sort_probe.json 2> $out/sortp.err
bash tools/probes/sort_pmc.sh > $out/pmc_sort.txt 2>> $out/sortp.err
rocprofv3 --kernel-trace --stats -d $out/sortks -o s --output-format csv -- python3 tools/sort_probe.py --reps 5 > /dev/null 2>> $out/sortp.err
grep -E "k_sort|k_iota|Name" $out/sortks/s_kernel_stats.csv > $out/sort_kernel_stats.csv 2>/dev/null; rm -rf $out/sortks gpurun_out/sortpmc
# the other configurations
python3 bench.py --config 2 --steps 3 > $out/bench_config2.json 2> $out/c2.err
python3 bench.py --config 3 --steps 2 > $out/bench_config3.json 2> $out/c3.err
python3 bench.py --config 4 > $out/bench_config4.json 2> $out/c4.err
# gpurun returns at most 64 MiB: keep the summaries, drop the raw traces they were computed from
cp $out/ks/ks_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
rm -rf $out/ks $out/ksq $out/k267ks $out/optks $out/opt_FETCH_SIZE $out/opt_WRITE_SIZE $out/kt $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/k6_*/ $out/bb
du -sh $out; ls -la $out | tail -30
