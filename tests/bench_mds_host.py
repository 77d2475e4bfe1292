"""Host-side throughput of the latents data path: gather shuffled batches from MDS shards into a staging buffer with the
native reader, next to the per-sample Python decode the reference does in its DataLoader workers (restated in
oracle/mds_ref.py).  CPU only; shards are synthetic and page-cache resident (steady state of a multi-epoch run).

    python tests/bench_mds_host.py [--samples 3000] [--batch 256] [--threads 8]

Lives under tests/ because it uses the CPU restatement (oracle/mds_ref.py) to write the synthetic shards and as the
per-sample baseline; nothing outside tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/.
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_diffusion_amd import data as mdata   # noqa: E402
from oracle import mds_ref                       # noqa: E402  (checker / baseline only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=3000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--image-size", type=int, default=256)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        mds_ref.write_synthetic_latents(d, a.samples, seed=1, size_limit=1 << 28, with_512=a.image_size == 512)
        ds = mdata.StreamingLatentsDataset(streams=[d], shuffle=True, image_size=a.image_size, cap_seq_size=77,
                                           cap_emb_dim=1024, batch_size=a.batch)
        per_sample = ds.cap_bytes + ds.lat_bytes
        cap = torch.empty(a.batch, 1, 77, 1024, dtype=torch.float16)
        lat = torch.empty(a.batch, ds.in_channels, ds.latent_res, ds.latent_res, dtype=torch.float16)
        order = np.random.default_rng(0).permutation(a.samples)
        nb = a.samples // a.batch
        for threads in sorted({1, a.threads}):
            for rep in range(2):                       # first pass warms the page cache
                t0 = time.perf_counter()
                for b in range(nb):
                    ds.read_batch(order[b * a.batch:(b + 1) * a.batch], cap, lat, threads)
                dt = time.perf_counter() - t0
            print(f"native gather  threads={threads:2d}: {nb * a.batch / dt:9.0f} samples/s  {nb * a.batch * per_sample / dt / 1e9:6.2f} GB/s")
        ref = mds_ref.RefMDSReader(d)
        n = min(a.samples, 512)
        t0 = time.perf_counter()
        for i in order[:n]:
            s = mds_ref.latents_getitem(ref[int(i)], a.image_size, 77, 1024)
            torch.from_numpy(s["caption_latents"]), torch.from_numpy(s["image_latents"])
        dt = time.perf_counter() - t0
        print(f"per-sample python decode (1 process): {n / dt:9.0f} samples/s  {n * per_sample / dt / 1e9:6.2f} GB/s")


if __name__ == "__main__":
    main()
