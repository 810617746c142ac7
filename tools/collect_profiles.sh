#!/bin/bash
# Copy the summaries of `bash tools/round_profiles.sh <tag>` (merged back into gpurun_out/<tag>_profiles/) into profiles/.
tag=${1:-rXX}
O=gpurun_out/${tag}_profiles
cp $O/bench.json profiles/${tag}_bench.json
cp $O/kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv
cp $O/kernel_stats_gas_after_gas.csv profiles/${tag}_bench_kernel_stats_gas_after_gas.csv
for f in sweep_union.json traffic_k_rt_lw_bb.json traffic_lut_opt.json k267_probe.json k267_kernel_stats.csv pmc_k6.txt opt_kernel_stats.csv \
         lut_opt_probe.json gases_probe.json fsck_tools_bench.json fsck_tools_bench_netcdf4.json sort_probe.json pmc_sort.txt sort_kernel_stats.csv \
         sweep_probe.json sweep_probe_double_rows.json batch_breakdown.txt bench_config2.json bench_config3.json bench_config4.json; do
  [ -s $O/$f ] && cp $O/$f profiles/${tag}_$f
done
git status --short profiles | head -40
