cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn_a -- python $R/scripts/pmc_attn.py > /dev/null 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn_b -- python $R/scripts/pmc_attn.py > /dev/null 2>&1
cd $R
python - <<'PY'
import collections, csv, glob
for d in ("gpurun_out/pmc_attn_a", "gpurun_out/pmc_attn_b"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "attn_" in n:
                k = n.split("(")[0].replace("void (anonymous namespace)::", "")
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("##", d)
        for k, c in sorted(acc.items()):
            print(" ", k)
            for cn, v in sorted(c.items()):
                print(f"      {cn:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
                m = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024
                g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8
                print(f"      -> MFMA pipe busy {100*m/g:.1f} % of the kernel; waves waiting {100*sum(c['SQ_WAIT_ANY'])/sum(c['SQ_WAVE_CYCLES']):.1f} %; waiting on LDS {100*sum(c['SQ_WAIT_INST_LDS'])/sum(c['SQ_WAVE_CYCLES']):.1f} %")
            if "SQ_LDS_BANK_CONFLICT" in c and "SQ_ACTIVE_INST_LDS" in c:
                print(f"      -> LDS bank-conflict cycles / LDS active cycles {100*sum(c['SQ_LDS_BANK_CONFLICT'])/max(1,sum(c['SQ_ACTIVE_INST_LDS'])):.1f} %")
PY
find gpurun_out/pmc_attn_a gpurun_out/pmc_attn_b -type f -size +1M -delete 2>/dev/null
