#!/bin/bash
# Round-3 GPU sessions (run through gpurun from the repo root):
#     /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu_session_r3.sh <tag> <stage> ...'
# Everything lands in gpurun_out/<tag>/ ; what should be judged is copied into profiles/ afterwards.
TAG=${1:-r03x}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
MB=$ROOT/tools/microbench/build/mb
pmcrun() {   # pmcrun <name> <counters...> -- <command...> : one counter pass, summary into $OUT/<name>.txt
  local NAME=$1; shift; local CTRS=(); while [ "$1" != "--" ]; do CTRS+=("$1"); shift; done; shift
  ( cd /tmp && timeout 600 rocprofv3 --pmc "${CTRS[@]}" --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$NAME -o pmc -- "$@" > $ROOT/$OUT/pmc_$NAME.stdout 2> $ROOT/$OUT/pmc_$NAME.err )
  local F=$(find $OUT/pmc_$NAME -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then python tools/pmc_calib.py $F > $OUT/$NAME.txt 2>&1; else echo "no counter file (see pmc_$NAME.err)" > $OUT/$NAME.txt; tail -5 $OUT/pmc_$NAME.err >> $OUT/$NAME.txt; fi
  rm -rf $OUT/pmc_$NAME
}
for W in "$@"; do case $W in
counters)
  ( cd /tmp && timeout 120 rocprofv3 -L 2>&1 | grep -iE "TCC_EA|FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS|MALL|TCC_REQ|TCC_READ|TCC_WRITE" | cut -c1-200 | head -120 ) > $OUT/counters_list.txt 2>&1; wc -l $OUT/counters_list.txt ;;
calib)
  timeout 300 $MB calib 12 > $OUT/calib_timing.txt 2>&1; cat $OUT/calib_timing.txt
  pmcrun calib_fetch FETCH_SIZE -- $MB calib 12
  pmcrun calib_write WRITE_SIZE -- $MB calib 12
  pmcrun calib_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- $MB calib 12
  head -40 $OUT/calib_fetch.txt ;;
sort)
  timeout 300 $MB sort 60 > $OUT/sort_timing.txt 2>&1; cat $OUT/sort_timing.txt
  pmcrun sort_write WRITE_SIZE -- $MB sort 60
  pmcrun sort_fetch FETCH_SIZE -- $MB sort 60 ;;
tests)
  timeout 1200 python -m pytest tests -m gpu -x -q -rs --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -15 $OUT/pytest_gpu.log ;;
bench20)
  ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_steps20_warmup5.json 2> $OUT/bench_steps20_warmup5.err; tail -c 3000 $OUT/bench_steps20_warmup5.json; tail -4 $OUT/bench_steps20_warmup5.err ;;
bench2)
  # the unwrapped multi-rank command on ONE GPU: both ranks on device 0, collectives on gloo (tests the self-launch on the GPU box)
  ( time SMR_BENCH_BACKEND=gloo SMR_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --resident-batches 2 ) > $OUT/bench_2ranks_one_gpu.json 2> $OUT/bench_2ranks_one_gpu.err; tail -c 1500 $OUT/bench_2ranks_one_gpu.json; tail -4 $OUT/bench_2ranks_one_gpu.err ;;
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
  ( cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_sq -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > /dev/null 2> $ROOT/$OUT/pmc_sq.err )
  find $OUT/pmc_sq -name "*counter_collection.csv" -exec python tools/pmc_summary.py {} \; > $OUT/pmc_sq.txt 2>&1; head -14 $OUT/pmc_sq.txt
  rm -rf $OUT/pmc_sq ;;
batchsize)
  # does a larger resident batch amortise the index streaming of the seed searches?
  for B in 4000000 8000000; do
    timeout 600 python bench.py --steps 4 --warmup 1 --resident-batches 2 --batch-reads $B --no-cpu-baseline > $OUT/bench_batch_$B.json 2> $OUT/bench_batch_$B.err
    python - $OUT/bench_batch_$B.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("batch", d["config"]["batch_reads"], "reads/s %.2f M" % (d["value"] / 1e6), "ms/step %.1f" % d["ms_per_step"], "pcie-incl %.2f M" % ((d["pcie_inclusive_reads_per_s_per_gpu"] or 0) / 1e6))
for k, v in r["kernels"].items(): print("   %-16s %.3f ms x %d" % (k, v["avg_launch_ms"], v["launches"]))
print("   k_chain %.1f ms/step  k_trace %.2f" % (d["kernels"]["k_chain"]["ms"] / d["steps"], d["kernels"]["k_trace"]["ms"] / d["steps"]))
PY
  done ;;
refs8)
  ( time timeout 900 python bench.py --workload refs8 --steps 5 --warmup 1 --resident-batches 2 ) > $OUT/bench_refs8.json 2> $OUT/bench_refs8.err; tail -c 2500 $OUT/bench_refs8.json; tail -4 $OUT/bench_refs8.err ;;
pacbio)
  ( time timeout 1200 python bench.py --workload pacbio5k --steps 3 --warmup 1 --resident-batches 2 ) > $OUT/bench_pacbio5k.json 2> $OUT/bench_pacbio5k.err; tail -c 2500 $OUT/bench_pacbio5k.json; tail -6 $OUT/bench_pacbio5k.err ;;
sqi)
  ( cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_sqi -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > $ROOT/$OUT/pmc_sqi.json 2> $ROOT/$OUT/pmc_sqi.err )
  find $OUT/pmc_sqi -name "*counter_collection.csv" -exec python tools/pmc_summary.py {} \; > $OUT/pmc_sqi.txt 2>&1; head -16 $OUT/pmc_sqi.txt
  rm -rf $OUT/pmc_sqi ;;
pacbio_phases)
  # where a wave of k_chain spends its cycles on 5 kb reads (-DSMR_CHAIN_PHASES build of the library)
  if [ -f sortmerna_amd/lib/libsmr_hip_phases.so ]; then
    cp sortmerna_amd/lib/libsmr_hip.so /tmp/libsmr_hip.keep && cp sortmerna_amd/lib/libsmr_hip_phases.so sortmerna_amd/lib/libsmr_hip.so
    SMR_DEBUG_PHASES=1 timeout 600 python bench.py --workload pacbio5k --steps 1 --warmup 1 --resident-batches 2 --no-cpu-baseline --profile-run > $OUT/pacbio_phases.json 2> $OUT/pacbio_phases.err; grep "phase cycles" $OUT/pacbio_phases.err | tail -3 | cut -c1-400
    cp /tmp/libsmr_hip.keep sortmerna_amd/lib/libsmr_hip.so
  fi ;;
phases)
  # where a wave of k_chain spends its cycles on the headline workload (-DSMR_CHAIN_PHASES build of the library, 2 M-read batches)
  if [ -f sortmerna_amd/lib/libsmr_hip_phases.so ]; then
    cp sortmerna_amd/lib/libsmr_hip.so /tmp/libsmr_hip.keep && cp sortmerna_amd/lib/libsmr_hip_phases.so sortmerna_amd/lib/libsmr_hip.so
    SMR_DEBUG_PHASES=1 timeout 300 python tools/hw_minibench.py > $OUT/minibench_phases.log 2>&1; grep -E "phase cycles|SW kernel" $OUT/minibench_phases.log | tail -8 | cut -c1-420
    cp /tmp/libsmr_hip.keep sortmerna_amd/lib/libsmr_hip.so
  fi ;;
quadab)
  # the 16-lane walk of the small marked reads (smr_quad.hpp) on and off, same mini bench
  for Q in 0 1; do SMR_QUAD=$Q timeout 300 python tools/hw_minibench.py > $OUT/minibench_quad$Q.log 2>&1; echo "== SMR_QUAD=$Q"; grep -E "SW kernel|kernels:" $OUT/minibench_quad$Q.log | tail -2 | cut -c1-900; done ;;
bloomab)
  # k_cand's Bloom bitmap per read: 512 words (3 blocks of k_cand per CU), 256 (6), 128 (9) -- fewer words mark more reads for k_chain
  for B in 512 256 128; do SMR_CAND_BLOOM=$B timeout 300 python tools/hw_minibench.py > $OUT/minibench_bloom$B.log 2>&1; echo "== SMR_CAND_BLOOM=$B"; grep -E "SW kernel|kernels:" $OUT/minibench_bloom$B.log | tail -2 | cut -c1-900; done ;;
twoctx)
  # two batches in flight on one GPU (two contexts, two streams) against one
  for N in 1 2; do MB_CTX=$N timeout 300 python tools/hw_minibench2.py > $OUT/minibench_ctx$N.log 2>&1; grep -E "context" $OUT/minibench_ctx$N.log | tail -3 | cut -c1-300; done ;;
twoctx8m)
  MB_CTX=3 timeout 300 python tools/hw_minibench2.py > $OUT/minibench_ctx3.log 2>&1; grep -E "context" $OUT/minibench_ctx3.log | tail -2 | cut -c1-300
  for N in 1 2; do MB_BATCH=8000000 MB_STEPS=2 MB_CTX=$N timeout 400 python tools/hw_minibench2.py > $OUT/minibench_8m_ctx$N.log 2>&1; grep -E "context" $OUT/minibench_8m_ctx$N.log | tail -2 | cut -c1-300; done ;;
handab)
  # k_cand's records of the marked reads' positions handed to k_chain (SMR_HANDOVER=1) against k_chain gathering them itself
  for H in 0 1; do SMR_HANDOVER=$H timeout 300 python tools/hw_minibench.py > $OUT/minibench_handover$H.log 2>&1; echo "== SMR_HANDOVER=$H"; grep -E "SW kernel|kernels:" $OUT/minibench_handover$H.log | tail -2 | cut -c1-900; done ;;
dropin)
  timeout 900 python -m pytest tests/test_dropin.py tests/test_cpp_driver.py -m gpu -x -q -rs > $OUT/pytest_dropin_mgpu.log 2>&1; tail -6 $OUT/pytest_dropin_mgpu.log ;;
mini)
  timeout 300 python tools/hw_minibench.py > $OUT/minibench.log 2>&1; tail -8 $OUT/minibench.log ;;
alt)
  # alternative builds of the library (sortmerna_amd/lib/libsmr_hip_alt*.so, made in the container) on the same mini bench
  for A in sortmerna_amd/lib/libsmr_hip_alt*.so; do [ -f $A ] || continue
    cp sortmerna_amd/lib/libsmr_hip.so /tmp/libsmr_hip.keep && cp $A sortmerna_amd/lib/libsmr_hip.so
    timeout 300 python tools/hw_minibench.py > $OUT/minibench_$(basename $A .so).log 2>&1; echo "== $A"; grep -E "SW kernel|kernels:" $OUT/minibench_$(basename $A .so).log | tail -4
    cp /tmp/libsmr_hip.keep sortmerna_amd/lib/libsmr_hip.so
  done ;;
e2e)
  SMR_IB_TIMING=1 timeout 900 python tools/e2e_cpp.py > $OUT/e2e_cpp.log 2>&1; grep -E "timing|smr index build|wall|aligned.fq" $OUT/e2e_cpp.log | tail -60 | cut -c1-420 ;;
esac; done
ls $OUT
