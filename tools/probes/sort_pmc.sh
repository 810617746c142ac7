#!/bin/bash
# counters of the radix sort's scatter and histogram kernels (separate passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sortpmc; mkdir -p $out; : > $out/pmc_sort.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $set | tr " " "_")
  rocprofv3 --kernel-trace --pmc $set -d $out/$n -o p --output-format csv -- python3 tools/sort_probe.py --reps 2 > /dev/null 2> $out/$n.err
  for k in k_sort_scatter_lds k_sort_hist k_sort_scan_rows; do
    echo "== $k" >> $out/pmc_sort.txt
    python3 tools/pmc_summary.py $k $out/$n/p_counter_collection.csv >> $out/pmc_sort.txt
  done
  rm -rf $out/$n
done
cat $out/pmc_sort.txt | sed 's#.*collection.csv: ##'
