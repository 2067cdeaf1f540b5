#!/bin/bash
# SQ counters (where the waves' cycles go) of ONE kernel run through tools/run_kernel.py.  usage (GPU box): tools/pmc_sq.sh <kernel> [batch] [name filter]
set -u
k=$1; b=${2:-1}; flt=${3:-}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf $out/pmcs_$i
  rocprofv3 --pmc $P --output-format csv -d $out/pmcs_$i -o run -- python3 $root/tools/run_kernel.py $k $b 3 > $out/pmcs_$i.log 2>&1
done
cd $root
python3 - "$k" "$flt" <<'PY' > $out/pmc_sq_$k.txt
import csv, glob, sys, collections
flt = sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcs_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "at::native" in n or "rocclr" in n or (flt and flt not in n):
            continue
        agg[n[:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print(open("gpurun_out/pmcs_1.log").read().strip().splitlines()[-1])
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m.get("SQ_WAVE_CYCLES", 0) < 1e5:
        continue
    print(k)
    for c in sorted(m):
        print(f"  {c:32s} {m[c]:16.0f}")
    wc = m["SQ_WAVE_CYCLES"]
    print("  -- of wave cycles: wait_any %.1f %%, wait_inst_any %.1f %%, active_inst_any %.1f %% (wait_inst_lds %.1f %%, active valu %.1f %%, active lds %.1f %%)" % tuple(
        100 * m.get(c, 0) / wc for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS")))
    print("  -- cycles per XCD (GRBM_GUI_ACTIVE / 8): %.0f; matrix pipe busy %.1f %%" % (m["GRBM_GUI_ACTIVE"] / 8, 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * m["GRBM_GUI_ACTIVE"] / 8)))
PY
rm -rf $out/pmcs_*
cat $out/pmc_sq_$k.txt
