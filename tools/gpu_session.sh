#!/bin/bash
# One GPU session on the MI355X box (run through gpurun from the repo root):
#     /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_session.sh r02b [tests|bench|prof|pmc|ab ...]'
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/ afterwards.
TAG=${1:-rXX}; shift
WHAT=${*:-tests bench prof pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
for W in $WHAT; do case $W in
tests)
  timeout 1200 python -m pytest tests -m gpu -x -q -rs --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -15 $OUT/pytest_gpu.log ;;
bench)
  timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 900 $OUT/bench_default.json ;;
bench20)
  # the driver's own command line
  ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_steps20_warmup5.json 2> $OUT/bench_steps20_warmup5.err; tail -c 1200 $OUT/bench_steps20_warmup5.json; tail -4 $OUT/bench_steps20_warmup5.err ;;
prof)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --profile-run > $ROOT/$OUT/bench_prof.json 2> $ROOT/$OUT/bench_prof.err )
  find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
  head -14 $OUT/kernel_stats.csv | cut -c1-60,150-260
  rm -rf $OUT/prof ;;
pmc)
  # HBM traffic of the seed stage: one counter per pass (FETCH_SIZE, WRITE_SIZE), at the bench's own batch size
  for CTR in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 500 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$CTR -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > $ROOT/$OUT/pmc_$CTR.json 2> $ROOT/$OUT/pmc_$CTR.err )
  done
  F=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $F $W 2000000 150 140000000 $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt 2>&1; cat $OUT/hbm_traffic.txt | head -30
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE ;;
sq)
  # issue / wait / LDS counters of the two biggest kernels (SQ block, one pass)
  ( cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_sq -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > /dev/null 2> $ROOT/$OUT/pmc_sq.err )
  find $OUT/pmc_sq -name "*counter_collection.csv" -exec python tools/pmc_summary.py {} \; > $OUT/pmc_sq.txt 2>&1; head -12 $OUT/pmc_sq.txt
  rm -rf $OUT/pmc_sq ;;
sqi)
  # instruction counts per kernel (SQ block, one pass)
  ( cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_sqi -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > $ROOT/$OUT/pmc_sqi.json 2> $ROOT/$OUT/pmc_sqi.err )
  find $OUT/pmc_sqi -name "*counter_collection.csv" -exec python tools/pmc_summary.py {} \; > $OUT/pmc_sqi.txt 2>&1; head -12 $OUT/pmc_sqi.txt
  rm -rf $OUT/pmc_sqi ;;
phases)
  if [ -f sortmerna_amd/lib/libsmr_hip_phases.so ]; then
    cp sortmerna_amd/lib/libsmr_hip.so /tmp/libsmr_hip.keep && cp sortmerna_amd/lib/libsmr_hip_phases.so sortmerna_amd/lib/libsmr_hip.so
    SMR_DEBUG_PHASES=1 timeout 300 python tools/hw_minibench.py > $OUT/minibench_phases.log 2>&1; grep -E "phase cycles|SW kernel" $OUT/minibench_phases.log | tail -12
    cp /tmp/libsmr_hip.keep sortmerna_amd/lib/libsmr_hip.so
  fi ;;
ab)
  # the wave_ror variant of the packed SW kernel against the default (same workload, torch-free)
  SMR_SW_PACKED=2 timeout 300 python tools/hw_minibench.py > $OUT/minibench_ror.log 2>&1; grep "SW kernel" $OUT/minibench_ror.log ;;
alt)
  # an alternative build of the library (sortmerna_amd/lib/libsmr_hip_alt.so, made in the container with other -D flags) on the same mini bench
  if [ -f sortmerna_amd/lib/libsmr_hip_alt.so ]; then
    cp sortmerna_amd/lib/libsmr_hip.so /tmp/libsmr_hip.keep && cp sortmerna_amd/lib/libsmr_hip_alt.so sortmerna_amd/lib/libsmr_hip.so
    timeout 300 python tools/hw_minibench.py > $OUT/minibench_alt.log 2>&1; grep -E "SW kernel" $OUT/minibench_alt.log | tail -4
    cp /tmp/libsmr_hip.keep sortmerna_amd/lib/libsmr_hip.so
  fi ;;
e2e)
  timeout 900 python tools/e2e_cpp.py > $OUT/e2e_cpp.log 2>&1; tail -12 $OUT/e2e_cpp.log ;;
mini)
  timeout 300 python tools/hw_minibench.py > $OUT/minibench.log 2>&1; tail -6 $OUT/minibench.log ;;
esac; done
ls $OUT
