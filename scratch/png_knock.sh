for k in 0 1 2 4 8 15; do
  export BSR_EXTRA_FLAGS="-DBSR_PNG_KNOCK=$k"
  python -c "from blindshadowremoval_amd.build import build_library; build_library(force=True)" && echo "knock $k: $(python scratch/png_time.py 2>&1 | tail -1)"
done
