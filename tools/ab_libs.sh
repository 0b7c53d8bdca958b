OUT=gpurun_out/$1; shift; mkdir -p $OUT
cp sortmerna_amd/lib/libsmr_hip.so /tmp/keep.so
for L in "$@"; do
  if [ $L = keep ]; then cp /tmp/keep.so sortmerna_amd/lib/libsmr_hip.so; else cp sortmerna_amd/lib/$L.so sortmerna_amd/lib/libsmr_hip.so; fi
  MB_STEPS=${MB_STEPS:-8} MB_NOCHECK=1 timeout 300 python tools/hw_minibench_r5.py $L >> $OUT/minibench_ab.log 2>&1
done
cp /tmp/keep.so sortmerna_amd/lib/libsmr_hip.so
grep -E "==|per step|FAILED|Error" $OUT/minibench_ab.log | sed -E 's/; sw_fwd.*//'
