"""The REAL data path feeding the XL/2 step, timed once on the GPU (the latents reader "must deliver
~0.7 GB/s/GPU"): `python train.py --config-name res_256_pretrain ...` as a subprocess, reading synthetic MDS
shards (written here with the CPU restatement's writer) through StreamingLatentsDataset -> LatentsLoader (pinned staging slots,
background gather thread, H2D copies on a side stream) -> Trainer.train_step.  Reports the step rate train.py itself logs
(samples_per_sec, window 1) over the steps after the warm-up, next to the batch-resident rate bench.py measures.

    python tests/bench_train_py_loader.py [--steps 10] [--samples 4096] [--microbatch 1024] [--out gpurun_out/train_py_loader.json]

Lives under tests/ because it uses oracle/mds_ref.py (test infrastructure) to write the shards."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mds_ref   # noqa: E402  (shard writer only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--skip", type=int, default=3, help="leading steps left out of the mean (arena sizing, first-touch)")
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--microbatch", type=int, default=1024)
    ap.add_argument("--bench-value", type=float, default=None, help="bench.py's images/sec of the same build, for the ratio")
    ap.add_argument("--override", action="append", default=[], help="extra train.py overrides (key=value), e.g. scheduler.t_warmup=10ba")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "train_py_loader.json"))
    a = ap.parse_args()
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        t0 = time.time()
        mds_ref.write_synthetic_latents(d, a.samples, seed=3, size_limit=1 << 28, with_512=False)
        shard_bytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
        t_write = time.time() - t0
        cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--config-path", os.path.join(ROOT, "configs"), "--config-name", "res_256_pretrain",
               f"dataset.train.datadir=[{d}]", f"trainer.max_duration={a.steps}ba", f"trainer.device_train_microbatch_size={a.microbatch}",
               "trainer.eval_interval=0ba", "trainer.save_interval=0ba", "trainer.save_folder=null", "+misc.log_interval=1"] + list(a.override)
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=1500)
        wall = time.time() - t0
    logs = []
    for line in r.stdout.splitlines():
        if line.startswith("{") and "samples_per_sec" in line:
            logs.append(json.loads(line))
    if r.returncode != 0 or len(logs) < a.skip + 2:
        print(r.stdout[-2000:])
        print(r.stderr[-4000:])
        raise SystemExit(f"train.py failed (rc {r.returncode}, {len(logs)} step lines)")
    rates = [l["samples_per_sec"] for l in logs[a.skip:]]
    mean = len(rates) / sum(1.0 / x for x in rates)            # harmonic: total samples / total time
    per_sample = shard_bytes / a.samples
    out = {"command": " ".join(cmd[1:]).replace(ROOT + "/", ""), "steps": len(logs), "skipped": a.skip,
           "images_per_sec_train_py_loader": mean, "per_step": [l["samples_per_sec"] for l in logs], "loss_last": logs[-1]["loss"],
           "loss_per_step": [l["loss"] for l in logs], "lr_per_step": [l["lr"] for l in logs],
           "shard_bytes_per_sample": per_sample, "host_to_device_gb_per_sec": mean * per_sample / 1e9,
           "samples_in_shards": a.samples, "shard_write_s": t_write, "wall_s": wall,
           "bench_py_images_per_sec": a.bench_value, "ratio_to_bench_py": (mean / a.bench_value) if a.bench_value else None,
           "note": "train.py end to end: MDS shards (page cache) -> native reader -> pinned slots -> side-stream H2D -> train_step; "
                   "noise warm-up schedule (lr ramps from 0), microbatch as given"}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
