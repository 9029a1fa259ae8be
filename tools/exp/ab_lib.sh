#!/bin/bash
# A/B of two builds of the library on one box: bash tools/exp/ab_lib.sh <old.so> [bench args]; alternates new, old, new, old
set -e
old=$1; shift
new=surfh_amd/libsurfh_amd.so
cp $new gpurun_out/_new.so
for i in 1 2; do
  for w in new old; do
    if [ $w = new ]; then cp gpurun_out/_new.so $new; else cp $old $new; fi
    python3 bench.py "$@" --cpu-seconds 0 --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms; surfh_cg', d.get('surfh_cg_it_s') and round(d['surfh_cg_it_s'],2))"
  done
done
cp gpurun_out/_new.so $new
rm -f gpurun_out/_new.so
