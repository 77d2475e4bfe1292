#!/bin/bash
# SQ counter pass (MFMA pipe busy, effective clock, wait shares) over ONE shape for a list of (library, variant) pairs:
#   scripts/pmc_w4.sh "M N K" lib.so:variant [lib.so:variant ...]      ('-' as library = the in-tree one)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
shape=$1; shift
for lv in "$@"; do
  lib=${lv%%:*}; var=${lv#*:}
  d=gpurun_out/pmc_w4_$(basename "$lib" .so)_$var
  rm -rf "$d"
  ( cd /tmp && if [ "$lib" != "-" ]; then export MICRODIT_LIB="$GRAFT_REPO_ROOT/$lib"; fi
    timeout -k 10 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
      SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$d" -- \
      python "$GRAFT_REPO_ROOT/scripts/pmc_gemm.py" $shape 1 1 $var b 6 > /dev/null 2>&1 )
  python - "$d" "$lv" <<'PY'
import collections, csv, glob, sys
d, tag = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_bf16' in r['Kernel_Name']:
            dur.setdefault('ns', []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gemm_bf16' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    a = {k: sum(v) / len(v) for k, v in acc.items()}
    ns = sum(dur['ns'][1:]) / max(1, len(dur['ns'][1:])) if dur else 0
    g = a.get('GRBM_GUI_ACTIVE', 0) / 8          # summed over the 8 XCDs
    print(f"{tag:40s} {ns / 1e3:8.1f} us  clock {g / ns if ns else 0:5.2f} GHz  MFMA busy {100 * a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / g if g else 0:5.1f} %  "
          f"wait_any {100 * a.get('SQ_WAIT_ANY', 0) / a.get('SQ_WAVE_CYCLES', 1):5.1f} %  wait_inst {100 * a.get('SQ_WAIT_INST_ANY', 0) / a.get('SQ_WAVE_CYCLES', 1):5.1f} %  "
          f"wait_lds {100 * a.get('SQ_WAIT_INST_LDS', 0) / a.get('SQ_WAVE_CYCLES', 1):5.1f} %  active {100 * a.get('SQ_ACTIVE_INST_ANY', 0) / a.get('SQ_WAVE_CYCLES', 1):5.1f} %")
PY
  find "$d" -type f -size +1M -delete 2>/dev/null
done
