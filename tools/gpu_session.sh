#!/bin/bash
# One GPU session on the MI355X box (run through gpurun from the repo root):
#     /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_session.sh r02'
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/ afterwards.
# Order = most important first, so that a cut-off session still leaves the essentials:
#   1. GPU parity tests   2. bench (default)   3. rocprofv3 kernel stats of the bench   4. A/B of the round's switches
#   5. device index build timing   6. PMC passes (HBM traffic of the seed stage; separate passes, no trace domains)
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/bench_prof.json 2> $OLDPWD/$OUT/bench_prof.err )
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# A/B of the Smith-Waterman kernels on the bench workload, torch-free (20 s): packed, 32-bit, packed again; then the wave_ror variant
timeout 200 python tools/hw_minibench.py > $OUT/minibench_modes_1_0_1.log 2>&1; tail -5 $OUT/minibench_modes_1_0_1.log
SMR_SW_PACKED=2 timeout 200 python tools/hw_minibench.py > $OUT/minibench_ror.log 2>&1; grep "SW kernel" $OUT/minibench_ror.log
# phase cycles of k_chain (library built in the container with -DSMR_CHAIN_PHASES: sortmerna_amd/lib/libsmr_hip_phases.so)
if [ -f sortmerna_amd/lib/libsmr_hip_phases.so ]; then
  cp sortmerna_amd/lib/libsmr_hip.so /tmp/libsmr_hip.keep && cp sortmerna_amd/lib/libsmr_hip_phases.so sortmerna_amd/lib/libsmr_hip.so
  SMR_DEBUG_PHASES=1 timeout 200 python tools/hw_minibench.py > $OUT/minibench_phases.log 2>&1; grep -E "phase cycles|SW kernel" $OUT/minibench_phases.log | tail -8
  cp /tmp/libsmr_hip.keep sortmerna_amd/lib/libsmr_hip.so
fi
# device vs host index build (14 Mnt and the bench DB size)
timeout 600 python - > $OUT/index_build.log 2>&1 <<'PY'
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import sortmerna_amd as smr
from sortmerna_amd import synth
e = smr.Engine(0)
for nt in (14_000_000, 140_000_000):
    d = tempfile.mkdtemp(prefix="smr_ib_")
    db = os.path.join(d, "db.fasta")
    synth.make_db(db, nt, seed=42, family_size=40, mean_len=1500)
    t = time.time(); h = smr.Index.build(db, 18, 3072.0, 10000, 0); th = time.time() - t
    t = time.time(); g = smr.Index.build_gpu(e, db, 18, 3072.0, 10000); tg = time.time() - t
    ih, ig = h[0].info(), g[0].info()
    print("db %d nt: host %.2f s, device %.2f s, same counts %s" % (nt, th, tg, (ih.n_ids, ih.n_pos, ih.n_nodes, ih.n_buckets) == (ig.n_ids, ig.n_pos, ig.n_nodes, ig.n_buckets)), flush=True)
    for ix in h + g:
        ix.free()
PY
cat $OUT/index_build.log
# HBM traffic of the seed stage: one counter per pass (FETCH_SIZE, WRITE_SIZE), see MI355X_MICROARCH.md for the unit / gfx950 corrections
for CTR in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $CTR -d $OLDPWD/$OUT/pmc_$CTR -o pmc -- python $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batch-reads 1000000 > /dev/null 2> $OLDPWD/$OUT/pmc_$CTR.err )
  find $OUT/pmc_$CTR -name "*counter_collection.csv" -exec python tools/pmc_summary.py {} \; > $OUT/pmc_$CTR.txt 2>&1
done
ls -la $OUT
