# Diagnostic: where the PNG encoder's time is — knock-out builds (wrong files), rocprof durations of png_rows_kernel for the three callers of scratch/png_time.py
for k in 0 1 2 4 8 16 31; do
  export BSR_EXTRA_FLAGS="-DBSR_PNG_KNOCK=$k"
  python -c "from blindshadowremoval_amd.build import build_library; build_library(force=True)" || continue
  cd /tmp && export TMPDIR=/tmp
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/png_prof
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/png_prof -o png -- python3 $GRAFT_REPO_ROOT/scratch/png_time.py > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python - "$k" <<'P'
import csv,glob,sys
for f in glob.glob('gpurun_out/png_prof/**/*kernel_trace.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'png_rows' in r['Kernel_Name']]
    # three callers in order: encode_figs (33 + 50 calls), u8 768, u8 1792
    d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
    n=len(d)//3
    med=lambda v: sorted(v)[len(v)//2]
    print("knock %2s: figs %.1f us   u8 768 %.1f us   u8 1792 %.1f us   (%d launches)" % (sys.argv[1], med(d[:n])/1e3, med(d[n:2*n])/1e3, med(d[2*n:])/1e3, len(d)))
P
done
unset BSR_EXTRA_FLAGS
python -c "from blindshadowremoval_amd.build import build_library; build_library(force=True)"
