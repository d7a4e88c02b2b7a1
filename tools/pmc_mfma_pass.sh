#!/bin/bash
# Matrix-pipe / clock counter passes for tools/pmc_mfma.py, run ON THE GPU BOX (gpurun) from the repo root:
#     bash tools/pmc_mfma_pass.sh <tag> [dtype ...]
# Counters are collected in their own rocprofv3 runs with --kernel-trace only, the program directly after `--`; a counter name the
# box's `rocprofv3 -L` does not list is dropped from its pass (an unknown name fails the whole pass).
tag=${1:-r3}; shift
dtypes=${@:-f32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/${tag}_counters_avail.txt 2>&1
pick() { local out=""; for c in "$@"; do grep -qw "$c" gpurun_out/${tag}_counters_avail.txt && out="$out $c"; done; echo $out; }
PASS_A=$(pick SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE)
PASS_B=$(pick SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE)
PASS_C=$(pick SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE)
echo "pass a: $PASS_A"; echo "pass b: $PASS_B"; echo "pass c: $PASS_C"
for d in $dtypes; do
  sfx=""; [ $d != f32 ] && sfx="_$d"
  for p in a b c; do
    eval ctr=\$PASS_$(echo $p | tr a-c A-C)
    [ -z "$ctr" ] && continue
    rm -rf gpurun_out/${tag}_pmc_mfma_${p}${sfx}
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/${tag}_pmc_mfma_${p}${sfx} -- python3 scratch/run_fwd.py 32 4 $d > gpurun_out/${tag}_pmc_mfma_${p}${sfx}.log 2>&1
    echo "pass $p $d rc=$?"
  done
done
