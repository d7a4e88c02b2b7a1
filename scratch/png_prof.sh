python scratch/png_time.py 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/png_prof -o png -- python3 $GRAFT_REPO_ROOT/scratch/png_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv,glob
for f in glob.glob('gpurun_out/png_prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
for f in glob.glob('gpurun_out/png_prof/**/*kernel_trace.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'png_rows' in r['Kernel_Name']]
    import collections
    by=collections.defaultdict(list)
    for r in rows: by[(r['Grid_Size_X'],r['Grid_Size_Y'],r.get('LDS_Block_Size',''))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    for k,v in by.items(): v.sort(); print(k, len(v), 'median ns', v[len(v)//2])
P
