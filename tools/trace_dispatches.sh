#!/bin/bash
# rocprofv3 --kernel-trace of two bench steps -> gpurun_out/<tag>/dispatches.txt: one line per dispatch of the round and sort kernels of the last step in launch
# order (start, duration, grid), then every kernel and copy of the steady state (after the first k_cand) summed by name.
#     /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/trace_dispatches.sh <tag>'
OUT=gpurun_out/${1:-r6trace}; mkdir -p $OUT; ROOT=$PWD; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/tr -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --profile-run > $ROOT/$OUT/bench_tr.json 2> $ROOT/$OUT/bench_tr.err )
F=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python - "$F" > $OUT/dispatches.txt <<'P'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
short=lambda r: r['Kernel_Name'].split('(')[0].replace('void smr::','').replace('smr::','')
ks=[r for r in rows if any(k in r['Kernel_Name'] for k in ('k_walk','k_sw16','k_cand','k_wnext','k_seed_pg','k_seed_split','k_seed_bins'))]
n=len(ks); ks=ks[2*n//3:]
t0=int(ks[0]['Start_Timestamp'])
for r in ks:
    print(short(r)[:16].ljust(16), 'start %9.3f ms  dur %8.3f ms  grid %s wg %s' % ((int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r.get('Grid_Size_X', r.get('Grid_Size','?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size','?'))))
first=next(i for i,r in enumerate(rows) if 'k_cand' in r['Kernel_Name'])
ss=rows[first:]
span=(int(ss[-1]['End_Timestamp'])-int(ss[0]['Start_Timestamp']))/1e6
tot=collections.Counter(); cnt=collections.Counter()
for r in ss: tot[short(r)[:40]]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6; cnt[short(r)[:40]]+=1
print('\nsteady state: %.1f ms from the first k_cand to the last kernel, %.1f ms inside kernels, %d dispatches' % (span, sum(tot.values()), len(ss)))
for k,v in tot.most_common(40): print('%-40s %6d dispatches %9.3f ms' % (k, cnt[k], v))
P
rm -rf $OUT/tr; tail -45 $OUT/dispatches.txt
