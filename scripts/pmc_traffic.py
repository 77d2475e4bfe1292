"""Turn the two rocprofv3 --pmc passes of scripts/pmc_workload.py into profiles/r4_gemm_traffic.json (bench.py reads it for
roofline.traffic and compares the `library_source_hash` stamped here with the build it runs: a stale file yields traffic null).  FETCH_SIZE / WRITE_SIZE are calibrated on the known-size copy launches of the same run (the guide's
gfx950 note: FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads; WRITE_SIZE is uncalibrated): the factor
that maps the counter to 2^30 bytes on the copy kernel is applied to the GEMM launches.
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    per = defaultdict(list)            # kernel name -> [value per dispatch]
    for f in files:
        by_dispatch = defaultdict(float)
        names = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                key = (row.get("Dispatch_Id"), row.get("Kernel_Name"))
                by_dispatch[key] += float(row["Counter_Value"])
                names[key] = row.get("Kernel_Name")
        for key, v in by_dispatch.items():
            per[names[key]].append(v)
    return per


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r4_gemm_traffic.json")
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")

    def copy_kernel(per):
        # the calibration launches: the elementwise copy kernel with the largest per-launch counter value
        cands = [(max(v), k) for k, v in per.items() if "copy" in k.lower() or "elementwise" in k.lower()]
        return max(cands)[1]
    fk, wk = copy_kernel(fetch), copy_kernel(write)
    f_cal = (1 << 30) / (sum(sorted(fetch[fk])[-3:]) / 3)
    w_cal = (1 << 30) / (sum(sorted(write[wk])[-3:]) / 3)
    gem_f = {k: v for k, v in fetch.items() if "gemm_bf16" in k}
    gem_w = {k: v for k, v in write.items() if "gemm_bf16" in k}
    n = sum(len(v) for v in gem_f.values())
    fb = sum(sum(v) for v in gem_f.values()) * f_cal
    wb = sum(sum(v) for v in gem_w.values()) * w_cal
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from micro_diffusion_amd import hip
    res = {"library_source_hash": hip._source_hash(), "gemm_source_hash": hip._gemm_source_hash(), "bytes_per_launch": (fb + wb) / n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n, "launches": n,
           "calibration": {"fetch_counter_to_bytes": f_cal, "write_counter_to_bytes": w_cal, "on": fk[:60],
                           "note": "factor that maps the counter to 2^30 bytes on a torch copy of 1 GiB in the same run"},
           "per_kernel": {k[:70]: {"launches": len(v), "fetch_bytes_per_launch": sum(v) * f_cal / len(v),
                                   "write_bytes_per_launch": sum(gem_w.get(k, [0])) * w_cal / max(1, len(gem_w.get(k, [0])))}
                          for k, v in gem_f.items()},
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over scripts/pmc_workload.py: one fwd+bwd of a 1024-image "
                     "XL/2 microbatch"}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
