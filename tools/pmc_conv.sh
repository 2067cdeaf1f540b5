#!/bin/bash
# SQ counters of the dominant conv (tools/run_conv.py): where the waves' cycles go.  Two passes (8 SQ slots each).
# usage (GPU box): tools/pmc_conv.sh <tag> [env assignments for run_conv.py ...]   -> gpurun_out/pmc_conv_<tag>.txt
set -u
tag=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
for v in "$@"; do export "$v"; done
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf $out/pmcc_$i
  rocprofv3 --pmc $P --output-format csv -d $out/pmcc_$i -o run -- python3 $root/tools/run_conv.py f16x3 3 > $out/pmcc_$i.log 2>&1
done
cd $root
python3 - "$tag" <<'PY' > $out/pmc_conv_$tag.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "conv3d_bf16s" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print(open("gpurun_out/pmcc_1.log").read().strip().splitlines()[-1])
for k, d in agg.items():
    print(k)
    m = {c: sum(v) / len(v) for c, v in d.items()}
    for c in sorted(m):
        print(f"  {c:32s} {m[c]:16.0f}")
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        print("  -- of wave cycles: wait_any %.1f %%, wait_inst_any %.1f %%, active_inst_any %.1f %% (wait_inst_lds %.1f %%, active valu %.1f %%, active lds %.1f %%)" % tuple(
            100 * m.get(c, 0) / wc for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS")))
        print("  -- matrix pipe busy: %.1f %% of GRBM_GUI_ACTIVE/8 x 1024 SIMDs" % (100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8)))
PY
rm -rf $out/pmcc_*
cat $out/pmc_conv_$tag.txt
