#!/bin/bash
# Hardware-counter passes over the config-C learn launch (scripts/ab_probe.py, one variant, one round).
# usage (GPU box, repo root): bash scripts/pmc_probe.sh TAG "name:window=2" [FWGPU_LIBRARY]
# PMC_CMD="bench.py --k 16 ..." profiles another python command (relative to the repo root) instead of ab_probe.py; PMC_MATCH selects the kernel lines (default fw_example_kernel).
# One rocprofv3 run per counter group (PMC only with --kernel-trace, as the pool requires); summaries in gpurun_out/TAG_pmc.txt
set -u
TAG=$1; SPEC=$2; LIB=${3:-}
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
[ -n "$LIB" ] && export FWGPU_LIBRARY=$LIB
export ROUNDS=1
cd /tmp && export TMPDIR=/tmp
CGROUPS=(
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_STALL_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
: > $OUT/${TAG}_pmc.txt
i=0
for g in "${CGROUPS[@]}"; do
  d=$OUT/${TAG}_pmc_$i
  rm -rf $d
  if [ -n "${PMC_CMD:-}" ]; then
    (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $g -d $d -o p -- python3 $PMC_CMD > $OUT/${TAG}_pmc_$i.log 2>&1)
  else
    timeout 300 rocprofv3 --kernel-trace --pmc $g -d $d -o p -- python3 $R/scripts/ab_probe.py "$SPEC" > $OUT/${TAG}_pmc_$i.log 2>&1
  fi
  python3 $R/scripts/rocprof_summary.py $(find $d -name "*.db") 2>&1 | grep -E "${PMC_MATCH:-fw_example_kernel}" | grep -v "<100" >> $OUT/${TAG}_pmc.txt
  rm -rf $d
  i=$((i+1))
done
# one line per counter: mean over the learn dispatches
python3 - "$OUT/${TAG}_pmc.txt" <<'PY'
import sys, collections
acc = collections.defaultdict(list); dur = []
for line in open(sys.argv[1]):
    parts = [x.strip() for x in line.rsplit(",", 8)]
    if len(parts) < 9: continue
    try:
        acc[parts[1]].append(float(parts[2])); dur.append(float(parts[3]))
    except ValueError:
        pass
print("mean over dispatches (kernel avg duration %.1f us):" % (sum(dur) / max(1, len(dur))))
for k, v in acc.items():
    print(f"  {k:40s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
