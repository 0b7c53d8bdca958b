#!/bin/bash
# A/B of library BUILDS on one GPU box (run through gpurun from the repo root):
#     hipcc ... -DSMR_SORT_PREFETCH=1 -o sortmerna_amd/lib/pf.so ...          (in the container; the .so files travel with the snapshot)
#     /usr/local/graft/bin/gpurun --timeout 900 -- 'MB_STEPS=8 bash tools/ab_libs.sh <tag> keep pf keep'
# Every named build (`keep` = the shipped libsmr_hip.so) runs tools/hw_minibench_r5.py in a process of its own; the lines land in gpurun_out/<tag>/minibench_ab.log.
OUT=gpurun_out/$1; shift; mkdir -p $OUT
cp sortmerna_amd/lib/libsmr_hip.so /tmp/keep.so
for L in "$@"; do
  if [ $L = keep ]; then cp /tmp/keep.so sortmerna_amd/lib/libsmr_hip.so; else cp sortmerna_amd/lib/$L.so sortmerna_amd/lib/libsmr_hip.so; fi
  MB_STEPS=${MB_STEPS:-8} timeout 300 python tools/hw_minibench_r5.py $L >> $OUT/minibench_ab.log 2>&1
done
cp /tmp/keep.so sortmerna_amd/lib/libsmr_hip.so
grep -E "==|per step|FAILED|Error" $OUT/minibench_ab.log | sed -E 's/; sw_fwd.*//'
