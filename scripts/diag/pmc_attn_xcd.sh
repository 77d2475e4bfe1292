#!/bin/bash
# L2 / fabric / TLB counters of the 64-row attention kernels under the two (batch, head) -> XCD mappings (scratch_libs/lib_noremap.so, lib_remap.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in noremap remap; do
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    MICRODIT_LIB=$R/scratch_libs/lib_$lib.so timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_axcd_${lib}_$i -- python $R/scripts/pmc_attn.py 256 small > $R/gpurun_out/pmc_axcd_${lib}_$i.log 2>&1
  done
done
cd $R
python - <<'PY' | tee gpurun_out/x6_pmc_attn_xcd.txt
import collections, csv, glob
for lib in ("noremap", "remap"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_axcd_{lib}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "attn_" in n:
                k = n.replace("void (anonymous namespace)::", "").split("(")[0]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("##", lib)
    for k, c in sorted(acc.items()):
        print(" ", k)
        for cn, v in sorted(c.items()):
            print(f"      {cn:40s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
find gpurun_out -path "*pmc_axcd_*" -type f -size +1M -delete 2>/dev/null
tail -3 gpurun_out/pmc_axcd_remap_1.log
