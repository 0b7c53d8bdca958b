#!/bin/bash
# Round-6 GPU sessions (run through gpurun from the repo root):
#     /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_session_r6.sh <tag> <stage> ...'
# Everything lands in gpurun_out/<tag>/ ; what should be judged is copied into profiles/ afterwards.
TAG=${1:-r6x}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
for W in "$@"; do case $W in
tests)
  timeout 1800 python -m pytest tests -m gpu -x -q -rs --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -15 $OUT/pytest_gpu.log ;;
one=*)
  # one=<pytest -k expression>   selected GPU tests
  timeout 900 python -m pytest tests -m gpu -x -q -rs -k "${W#one=}" > $OUT/pytest_one.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_one.log
  tail -8 $OUT/pytest_one.log ;;
paritytests)
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q -rs > $OUT/pytest_parity.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_parity.log
  tail -8 $OUT/pytest_parity.log ;;
mb=*)
  # mb=NAME:ENV=v,ENV=v+NAME2:...   several library configurations in one process (tools/hw_minibench_r5.py)
  CFGS=$(echo "${W#mb=}" | tr '+' ' ')
  timeout 600 python tools/hw_minibench_r5.py $CFGS > $OUT/minibench.log 2>&1; grep -E "==|per step|hit lists|FAILED|walk rounds" $OUT/minibench.log | cut -c1-1200 ;;
mb2m=*)
  CFGS=$(echo "${W#mb2m=}" | tr '+' ' ')
  MB_BATCH=2000000 timeout 600 python tools/hw_minibench_r5.py $CFGS > $OUT/minibench2m.log 2>&1; grep -E "==|per step|FAILED|walk rounds" $OUT/minibench2m.log | cut -c1-1200 ;;
pmcmb)
  # issue-side counters of every kernel on the mini bench (one 8 M-read step per pass): instruction counts, then wave cycles / waits
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do N=$(echo $SET | cut -d' ' -f2)
    ( cd /tmp && MB_STEPS=1 timeout 500 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$N -o pmc -- python $ROOT/tools/hw_minibench_r5.py base > $ROOT/$OUT/pmcmb_$N.log 2> $ROOT/$OUT/pmcmb_$N.err )
    find $OUT/pmc_$N -name "*counter_collection.csv" -exec python tools/pmc_summary.py {} \; > $OUT/pmcmb_$N.txt 2>&1
    grep -E "k_walk|k_sw16|k_wnext|k_wlist|k_cand|k_chain|k_begins|k_seed" $OUT/pmcmb_$N.txt | cut -c1-420
    find $OUT/pmc_$N -name "*counter_collection.csv" -exec python tools/pmc_dispatches.py {} k_walk \; > $OUT/pmcmb_${N}_k_walk_dispatches.txt 2>&1
    rm -rf $OUT/pmc_$N
  done ;;
pmcalt=*)
  # pmcalt=LIB   the instruction / cycle counters of the seed kernels on the mini bench with the alternative build sortmerna_amd/lib/LIB.so
  LIBN=${W#pmcalt=}
  cp sortmerna_amd/lib/libsmr_hip.so /tmp/libsmr_hip.keep && cp sortmerna_amd/lib/$LIBN.so sortmerna_amd/lib/libsmr_hip.so
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do N=$(echo $SET | cut -d' ' -f2)
    ( cd /tmp && MB_STEPS=1 timeout 500 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$N -o pmc -- python $ROOT/tools/hw_minibench_r5.py base > $ROOT/$OUT/pmcalt_${LIBN}_$N.log 2> $ROOT/$OUT/pmcalt_${LIBN}_$N.err )
    find $OUT/pmc_$N -name "*counter_collection.csv" -exec python tools/pmc_summary.py {} \; > $OUT/pmcalt_${LIBN}_$N.txt 2>&1
    grep -E "k_seed" $OUT/pmcalt_${LIBN}_$N.txt | cut -c1-420
    rm -rf $OUT/pmc_$N
  done
  cp /tmp/libsmr_hip.keep sortmerna_amd/lib/libsmr_hip.so ;;
alt=*)
  # alt=LIB[:ENV=v,...]  the mini bench on an alternative build sortmerna_amd/lib/LIB.so (made in the container), e.g. the -DSMR_WALK_PHASES one
  A=${W#alt=}; LIBN=${A%%:*}; ENVS=""; [ "$A" != "$LIBN" ] && ENVS=${A#*:}
  cp sortmerna_amd/lib/libsmr_hip.so /tmp/libsmr_hip.keep && cp sortmerna_amd/lib/$LIBN.so sortmerna_amd/lib/libsmr_hip.so
  SMR_DEBUG_PHASES=1 MB_STEPS=2 timeout 400 python tools/hw_minibench_r5.py $LIBN:$ENVS > $OUT/minibench_$LIBN.log 2>&1; grep -E "==|per step|FAILED|phase cycles" $OUT/minibench_$LIBN.log | cut -c1-700 | tail -8
  cp /tmp/libsmr_hip.keep sortmerna_amd/lib/libsmr_hip.so ;;
bench20)
  ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_steps20_warmup5.json 2> $OUT/bench_steps20_warmup5.err; tail -c 3000 $OUT/bench_steps20_warmup5.json; tail -4 $OUT/bench_steps20_warmup5.err ;;
prof)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --profile-run > $ROOT/$OUT/bench_prof.json 2> $ROOT/$OUT/bench_prof.err )
  find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
  head -24 $OUT/kernel_stats.csv | cut -c1-60,150-260
  rm -rf $OUT/prof ;;
prof=*)
  # prof=WORKLOAD   rocprofv3 kernel stats of one of the secondary workloads (config2, refs8, pacbio5k)
  WL=${W#prof=}
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_$WL -o bench -- python $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --resident-batches 2 --no-cpu-baseline > $ROOT/$OUT/bench_prof_$WL.json 2> $ROOT/$OUT/bench_prof_$WL.err )
  find $OUT/prof_$WL -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$WL.csv \;
  python - <<EOF
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats_$WL.csv")))
for r in rows[:32]:
    print("%-70s calls %6s total %10.3f ms avg %9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
EOF
  rm -rf $OUT/prof_$WL ;;
pmc)
  for CTR in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 500 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$CTR -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > $ROOT/$OUT/pmc_$CTR.json 2> $ROOT/$OUT/pmc_$CTR.err )
  done
  F=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W2=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $F $W2 8000000 150 140000000 $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt 2>&1; cat $OUT/hbm_traffic.txt | head -40
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE ;;
sq)
  # derived issue-side metrics per kernel family, every dispatch weighted by its duration (tools/pmc_sq.py), + SQ_INSTS_VALU of k_sw16 per cell pair
  FILES=""
  for SET in "VALUBusy SALUBusy LDSBankConflict" "MemUnitStalled VALUUtilization" "SQ_INSTS_VALU SQ_WAVES"; do N=$(echo $SET | cut -c1-8)
    ( cd /tmp && timeout 500 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_sq_$N -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > $ROOT/$OUT/pmc_sq_$N.json 2> $ROOT/$OUT/pmc_sq_$N.err )
    FILES="$FILES $(find $OUT/pmc_sq_$N -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_sq_$N -name "*kernel_trace.csv" | head -1)"
  done
  CELLS=$(python -c "
import json
o=json.loads([l for l in open('$OUT/pmc_sq_SQ_INSTS.json') if l.startswith('{')][-1]); print(2 * o['kernels']['k_chain']['sw_cells'])")      # (the pass counts the warm-up step and the timed step: two steps' cells)
  ( cd tools && python pmc_sq.py $ROOT/$OUT/sq_counters.json 8000000 150 140000000 $(for F in $FILES; do echo $ROOT/$F; done) --sw-cells $CELLS ) > $OUT/sq_counters.txt 2>&1; cat $OUT/sq_counters.txt
  rm -rf $OUT/pmc_sq_* ;;
sqi)
  ( cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_sqi -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run > $ROOT/$OUT/pmc_sqi.json 2> $ROOT/$OUT/pmc_sqi.err )
  find $OUT/pmc_sqi -name "*counter_collection.csv" -exec python tools/pmc_summary.py {} \; > $OUT/pmc_sqi.txt 2>&1; head -24 $OUT/pmc_sqi.txt
  rm -rf $OUT/pmc_sqi ;;
refs8)
  ( time timeout 900 python bench.py --workload refs8 --steps 5 --warmup 1 --resident-batches 2 ) > $OUT/bench_refs8.json 2> $OUT/bench_refs8.err; tail -c 2500 $OUT/bench_refs8.json; tail -4 $OUT/bench_refs8.err ;;
pacbio)
  ( time timeout 1200 python bench.py --workload pacbio5k --steps 3 --warmup 1 --resident-batches 2 ) > $OUT/bench_pacbio5k.json 2> $OUT/bench_pacbio5k.err; tail -c 2500 $OUT/bench_pacbio5k.json; tail -6 $OUT/bench_pacbio5k.err ;;
config2)
  ( time timeout 900 python bench.py --workload config2 --steps 5 --warmup 1 ) > $OUT/bench_config2.json 2> $OUT/bench_config2.err; tail -c 2500 $OUT/bench_config2.json; tail -6 $OUT/bench_config2.err ;;
ranks8)
  # `python bench.py --gpus 8`, its 8 ranks on the one GPU (2 M-read batches: 8 ranks with 8 M-read batches need 8 x 36 GB of the one GPU's HBM) of this box (the two tiny collectives on gloo): what an 8-GPU node will run, set-up time per rank included
  ( time SMR_BENCH_BACKEND=gloo SMR_BENCH_DEVICE=0 timeout 1200 python bench.py --gpus 8 --steps 3 --warmup 1 --batch-reads ${RANKS8_BATCH:-2000000} ) > $OUT/bench_8ranks_on_one_gpu.json 2> $OUT/bench_8ranks_on_one_gpu.err
  grep -E "set-up|real" $OUT/bench_8ranks_on_one_gpu.err | tail -4; python -c "
import json,sys
o=json.loads([l for l in open('$OUT/bench_8ranks_on_one_gpu.json') if l.startswith('{')][-1])
print('value %.3g reads/s, nranks %s, n_gpus %d, setup %s' % (o['value'], o['config'].get('nranks'), o['n_gpus'], o['config']['setup_s']))" ;;
counters)
  ( cd /tmp && timeout 120 rocprofv3 -L > $ROOT/$OUT/counters_list.txt 2>&1 ); grep -iE "^\s*(Name|Counter)|TCC_EA.*(WR|STALL)|TCP_.*STALL|SQ_WAIT_INST|SQ_INSTS_LDS|SQ_INST_CYCLES|LDS_IDX|BARRIER|TCC_.*BUBBLE|WRREQ" $OUT/counters_list.txt | head -80 | cut -c1-200 ;;
pmcx=*)
  # pmcx=COUNTER+COUNTER+...  one PMC pass of the given counters on the mini bench, per kernel
  SET=$(echo "${W#pmcx=}" | tr '+' ' ')
  ( cd /tmp && MB_STEPS=1 timeout 500 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $ROOT/$OUT/pmcx -o pmc -- python $ROOT/tools/hw_minibench_r5.py base > $ROOT/$OUT/pmcx.log 2> $ROOT/$OUT/pmcx.err )
  find $OUT/pmcx -name "*counter_collection.csv" -exec python tools/pmc_summary.py {} \; > $OUT/pmcx_$(echo $SET | cut -d' ' -f1).txt 2>&1
  grep -E "k_walk|k_sw16|k_cand|k_seed" $OUT/pmcx_$(echo $SET | cut -d' ' -f1).txt | cut -c1-420
  rm -rf $OUT/pmcx ;;
r8dbg)
  SMR_WALK_DEBUG=1 timeout 400 python bench.py --workload refs8 --steps 1 --warmup 0 --no-cpu-baseline --resident-batches 1 > $OUT/bench_refs8_dbg.json 2> $OUT/bench_refs8_dbg.err; grep "walk rounds" $OUT/bench_refs8_dbg.err | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12,$13,$14,$15,$16,$17,$18,$19,$20,$21,$22,$23,$24}' | head -60 ;;
c2dbg)
  SMR_WALK_DEBUG=1 timeout 300 python bench.py --workload config2 --steps 1 --warmup 0 --no-cpu-baseline --resident-batches 1 > $OUT/bench_config2_dbg.json 2> $OUT/bench_config2_dbg.err; grep "walk rounds" $OUT/bench_config2_dbg.err | head -8 | cut -c1-300 ;;
dropin)
  timeout 900 python -m pytest tests/test_dropin.py tests/test_cpp_driver.py -m gpu -x -q -rs > $OUT/pytest_dropin_mgpu.log 2>&1; tail -6 $OUT/pytest_dropin_mgpu.log ;;
e2e)
  timeout 280 python tools/e2e_quick.py > $OUT/e2e_quick.log 2>&1; tail -12 $OUT/e2e_quick.log | cut -c1-700 ;;
esac; done
ls $OUT
