#!/bin/bash
# ablation of K8a's phases under rocprofv3 (ECCKD_K8A_DEBUG bit mask: 1 no gather, 2 no recurrences, 4 no band sums, 8 no cost, 16 no owner adjoint)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
for m in 0 1 2 4 8 16 31; do
  ECCKD_K8A_DEBUG=$m rocprofv3 --kernel-trace --stats -d $out/m$m -o t --output-format csv -- python3 bench.py --config 4 --lut-opt-iterations 15 > $out/m$m.json 2> $out/m$m.err
  grep -h "k_opt_forward_adjoint\|k_opt_gradient" $out/m$m/t_kernel_stats.csv | cut -c1-60,200- | sed "s/^/mask $m: /" | awk -F, '{print $1, $(NF-7), $(NF-5)}' 
  python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$out/m$m/t_kernel_trace.csv")))
agg=collections.defaultdict(list)
for r in rows:
    if 'k_opt_' in r['Kernel_Name']:
        agg[(r['Kernel_Name'].split('(')[1][-30:] if False else r['Kernel_Name'][22:60], r['Grid_Size_X'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(agg.items()): print("mask $m", k, len(v), round(sum(v)/len(v)/1e3,1))
PY
done
