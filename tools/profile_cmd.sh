#!/bin/bash
# rocprofv3 passes over ANY python command (run on the GPU box via gpurun):
#   tools/profile_cmd.sh <outdir> <kernel-name substring> <script.py> [args...]
# One kernel-trace/stats pass, then PMC passes (counters only, never mixed with tracing), then a JSON summary of the
# kernels whose name contains the substring (tools/profile_cmd_summary.py). The program after `--` is python3 itself.
set -uo pipefail
OUT=$1; KSUB=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$@" > "$OUT/trace.log" 2>&1
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc$i" -- python3 "$@" > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 tools/profile_cmd_summary.py "$OUT" "$KSUB"
