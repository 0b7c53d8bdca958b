#!/bin/bash
# Round-4 GPU sessions (run through gpurun from the repo root):
#     /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_session_r4.sh <tag> <stage> ...'
# Everything lands in gpurun_out/<tag>/ ; what should be judged is copied into profiles/ afterwards.
TAG=${1:-r4x}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
mini() {   # mini <name> [ENV=val ...] : the 2 M-read mini bench under the given environment, last lines into the log
  local NAME=$1; shift
  env "$@" timeout 300 python tools/hw_minibench.py > $OUT/minibench_$NAME.log 2>&1
  echo "== $NAME $*"; grep -E "SW kernel|kernels:" $OUT/minibench_$NAME.log | tail -2 | cut -c1-1100
}
for W in "$@"; do case $W in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q -rs --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -15 $OUT/pytest_gpu.log ;;
seedtests)
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q -rs > $OUT/pytest_seed.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_seed.log
  tail -8 $OUT/pytest_seed.log ;;
mini) mini base ;;
pgab)
  mini grid0 SMR_PG_GRID=0
  mini grid64k_swz SMR_PG_SWZ=1
  mini grid16k SMR_PG_GRID=16384
  mini grid16k_swz SMR_PG_GRID=16384 SMR_PG_SWZ=1 ;;
mini8m)
  MB_BATCH=8000000 timeout 400 python tools/hw_minibench.py > $OUT/minibench_8m.log 2>&1; grep -E "SW kernel|kernels:" $OUT/minibench_8m.log | tail -2 | cut -c1-1100 ;;
alt)
  # alternative builds of the library (sortmerna_amd/lib/libsmr_hip_alt*.so, made in the container) on the same mini bench
  for A in sortmerna_amd/lib/libsmr_hip_alt*.so; do [ -f $A ] || continue
    cp sortmerna_amd/lib/libsmr_hip.so /tmp/libsmr_hip.keep && cp $A sortmerna_amd/lib/libsmr_hip.so
    timeout 300 python tools/hw_minibench.py > $OUT/minibench_$(basename $A .so).log 2>&1; echo "== $A"; grep -E "SW kernel|kernels:" $OUT/minibench_$(basename $A .so).log | tail -2 | cut -c1-1100
    cp /tmp/libsmr_hip.keep sortmerna_amd/lib/libsmr_hip.so
  done ;;
bench20)
  ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_steps20_warmup5.json 2> $OUT/bench_steps20_warmup5.err; tail -c 3000 $OUT/bench_steps20_warmup5.json; tail -4 $OUT/bench_steps20_warmup5.err ;;
prof)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --profile-run > $ROOT/$OUT/bench_prof.json 2> $ROOT/$OUT/bench_prof.err )
  find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
  head -16 $OUT/kernel_stats.csv | cut -c1-60,150-260
  rm -rf $OUT/prof ;;
pmc)
  for CTR in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 500 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$CTR -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > $ROOT/$OUT/pmc_$CTR.json 2> $ROOT/$OUT/pmc_$CTR.err )
  done
  F=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $F $W 8000000 150 140000000 $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt 2>&1; cat $OUT/hbm_traffic.txt | head -40
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE ;;
sq)
  # the issue-side counters (rocprofv3 derived metrics, at most three per pass) -> sq_counters.json
  FILES=""
  for SET in "VALUBusy SALUBusy LDSBankConflict" "MemUnitStalled VALUUtilization"; do N=$(echo $SET | cut -c1-8)
    ( cd /tmp && timeout 500 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_sq_$N -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > /dev/null 2> $ROOT/$OUT/pmc_sq_$N.err )
    FILES="$FILES $(find $OUT/pmc_sq_$N -name "*counter_collection.csv" | head -1)"
  done
  ( cd tools && python pmc_sq.py $ROOT/$OUT/sq_counters.json 8000000 150 140000000 $(for F in $FILES; do echo $ROOT/$F; done) ) > $OUT/sq_counters.txt 2>&1; cat $OUT/sq_counters.txt
  rm -rf $OUT/pmc_sq_* ;;
sqi)
  ( cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_sqi -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > $ROOT/$OUT/pmc_sqi.json 2> $ROOT/$OUT/pmc_sqi.err )
  find $OUT/pmc_sqi -name "*counter_collection.csv" -exec python tools/pmc_summary.py {} \; > $OUT/pmc_sqi.txt 2>&1; head -18 $OUT/pmc_sqi.txt
  rm -rf $OUT/pmc_sqi ;;
refs8)
  ( time timeout 900 python bench.py --workload refs8 --steps 5 --warmup 1 --resident-batches 2 ) > $OUT/bench_refs8.json 2> $OUT/bench_refs8.err; tail -c 2500 $OUT/bench_refs8.json; tail -4 $OUT/bench_refs8.err ;;
pacbio)
  ( time timeout 1200 python bench.py --workload pacbio5k --steps 3 --warmup 1 --resident-batches 2 ) > $OUT/bench_pacbio5k.json 2> $OUT/bench_pacbio5k.err; tail -c 2500 $OUT/bench_pacbio5k.json; tail -6 $OUT/bench_pacbio5k.err ;;
phases)
  # where a wave spends its cycles on the headline workload (-DSMR_CHAIN_PHASES / -DSMR_SEED_PHASES build of the library, 2 M-read batches)
  if [ -f sortmerna_amd/lib/libsmr_hip_phases.so ]; then
    cp sortmerna_amd/lib/libsmr_hip.so /tmp/libsmr_hip.keep && cp sortmerna_amd/lib/libsmr_hip_phases.so sortmerna_amd/lib/libsmr_hip.so
    SMR_DEBUG_PHASES=1 timeout 300 python tools/hw_minibench.py > $OUT/minibench_phases.log 2>&1; grep -E "phase cycles|SW kernel" $OUT/minibench_phases.log | tail -8 | cut -c1-420
    cp /tmp/libsmr_hip.keep sortmerna_amd/lib/libsmr_hip.so
  fi ;;
dropin)
  timeout 900 python -m pytest tests/test_dropin.py tests/test_cpp_driver.py -m gpu -x -q -rs > $OUT/pytest_dropin_mgpu.log 2>&1; tail -6 $OUT/pytest_dropin_mgpu.log ;;
e2e)
  timeout 280 python tools/e2e_quick.py > $OUT/e2e_quick.log 2>&1; tail -12 $OUT/e2e_quick.log | cut -c1-700 ;;
esac; done
ls $OUT
