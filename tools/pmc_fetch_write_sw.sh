cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$set
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcs_$set -o p -- python3 bench.py --config 2 --steps 1 --warmup 0 > /dev/null 2>&1
  f=$(ls /tmp/pmcs_$set/*counter_collection.csv | head -1)
  for k in k_gas_prep_sw_staged k_scatter_column_halves k_rt_sw_bb_fast; do python3 tools/pmc_summary.py $k $f | sed "s#^.*csv: #$k #"; done
done
