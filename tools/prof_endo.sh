#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02h
(cd /tmp && NCG_TIMING=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02h/st_endo -- python $GRAFT_REPO_ROOT/tools/endo_timing.py > $GRAFT_REPO_ROOT/gpurun_out/r02h/st_endo.log 2>&1)
grep "^curve\|host finish" gpurun_out/r02h/st_endo.log | tail -6
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02h/st_endo/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
seq=[(r['Kernel_Name'][:58],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,int(r['Start_Timestamp'])) for r in rows]
for tag in ('CurveG1','CurveG2P'):
    idx=[i for i,s in enumerate(seq) if 'k_msm_accum<ncg::%s'%tag in s[0]]
    i0=idx[-1]; j=i0
    while 'k_msm_digits' not in seq[j][0]: j-=1
    t0=seq[j][2]; agg={}
    k=j
    while True:
        s=seq[k]; nm=s[0].split('(')[0].replace('void ncg::','').replace('ncg::','')
        agg.setdefault(nm,[0,0.0]); agg[nm][0]+=1; agg[nm][1]+=s[1]
        if 'group_pending' in s[0]: break
        k+=1
    print(tag,'span %.1f us'%((seq[k][2]-t0)/1e3+seq[k][1]))
    for nm,(c,t) in agg.items(): print('   %-50s x%-3d %8.1f us'%(nm[:50],c,t))
PY
find gpurun_out/r02h/st_endo -name "*.db" -delete
