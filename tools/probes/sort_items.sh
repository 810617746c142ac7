#!/bin/bash
# K3 A/B: keys per thread of the radix-sort pass kernels (tile = 256 x items), same ranks required
for it in 16 8 12 15 20; do
  ECCKD_SORT_ITEMS=$it python3 tools/sort_probe.py --reps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('items', $it, 'one_band_ms', round(d['one_band_ms'],3), 'thirteen', round(d['thirteen_bands_ms'],3), d['one_band_rank_checksum'], d['thirteen_bands_rank_checksum'])"
done
