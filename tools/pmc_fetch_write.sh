cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$set
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$set -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-lut-opt --no-sw > gpurun_out/pmc_bench_$set.json 2>/dev/null
  f=$(ls /tmp/pmc_$set/*counter_collection.csv | head -1)
  for k in k_rt_lw_bb_mirror k_gas_prep_lw_mirror k_scatter_column_halves k_reorder_key_lw_fast; do python3 tools/pmc_summary.py $k $f | sed "s#^.*csv: #$k $set: #"; done
  cp $f gpurun_out/pmc_$set.csv
done
