OUT=gpurun_out/r6s38; mkdir -p $OUT; ROOT=$PWD; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/tr -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --profile-run > $ROOT/$OUT/bench_tr.json 2> $ROOT/$OUT/bench_tr.err )
F=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python - "$F" > $OUT/dispatches.txt <<'P'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
ks=[r for r in rows if any(k in r['Kernel_Name'] for k in ('k_walk','k_sw16','k_cand','k_wnext','k_seed_pg','k_seed_split','k_seed_bins'))]
ks.sort(key=lambda r:int(r['Start_Timestamp']))
# last step only: take the last third
n=len(ks); ks=ks[2*n//3:]
t0=int(ks[0]['Start_Timestamp'])
for r in ks:
    nm=r['Kernel_Name'].split('(')[0].replace('void smr::','').replace('smr::','')
    print(nm[:16].ljust(16), 'start %9.3f ms  dur %8.3f ms  grid %s wg %s' % ((int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r.get('Grid_Size_X', r.get('Grid_Size','?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size','?'))))
P
rm -rf $OUT/tr; head -5 $OUT/dispatches.txt
