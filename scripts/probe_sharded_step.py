"""What one rank of an 8-GPU run executes, timed on ONE GPU: per-rank batch 256 (one microbatch), the whole exchange path on a
one-rank RCCL communicator (every collective is the identity, but the staging casts, the side-stream norms, the asynchronous
launches and the stream waits are the production ones), and -- for the sharded optimiser -- the AdamW pass over 1 / 8 of every
bucket (FusedAdamW.step_sharded(chunk_of=8)).  Prints one JSON line per mode: step time, images/s of this rank's share, the
compute-stream time of the optimiser section (norm finish + AdamW) and of the wait for the exchange.

    python scripts/probe_sharded_step.py [--per-rank 256] [--steps 4]"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-rank", type=int, default=256)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--pretend-world", type=int, default=8)
    ap.add_argument("--cu-limit", type=int, default=0, help="limit every persistent-GEMM grid to this many CUs (248 = what 8 RCCL channels leave)")
    ap.add_argument("--ab-store", action="store_true", help="sharded mode only: the one-microbatch bf16 store path off / on, alternating")
    a = ap.parse_args()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from micro_diffusion_amd.trainer import Trainer
    st = bench.Stage("res_256_pretrain", "MicroDiT_XL_2", a.per_rank, a.per_rank, 1, 0)      # global batch = the rank's share
    runs = [("sharded", 0), ("sharded", 1), ("sharded", 1), ("sharded", 0)] if a.ab_store else [("allreduce", 1), ("sharded", 1)]
    for mode, store in runs:
        tr = Trainer(st.model, st.trainer.opt, st.trainer.schedule, clip_norm=st.trainer.clip_norm, microbatch_size=a.per_rank,
                     exchange="bf16", single_rank_exchange=True, dp_mode=mode)
        tr.batches_seen = 100
        tr.measure_comm = True
        if mode == "sharded":
            tr.shard_chunk_of = a.pretend_world
        tr.sync.store_bf16 = bool(store)
        if a.cu_limit:
            st.model.dit.engine.cu_limit_fn = lambda lim=a.cu_limit: lim
        st.trainer = tr
        e, loss = st.timed(a.steps, 2, 1)
        out = {"mode": mode, "store_bf16": bool(store), "cu_limit": a.cu_limit, "gradient_launches_stored": tr.sync.last_stored, "per_rank_batch": a.per_rank, "ms_per_step": e / a.steps * 1e3, "rank_images_per_s": a.per_rank * a.steps / e,
               "optimizer_ms": tr.optimizer_ms(last=a.steps), "exchange_wait_ms": tr.exposed_comm_ms(last=a.steps),
               "buckets": tr.sync.last_buckets, "exchange": tr.sync.describe(),
               "note": ("AdamW over 1/%d of every bucket + the whole small region; collectives are identities on one rank" % a.pretend_world)
               if mode == "sharded" else "every rank runs the whole AdamW pass"}
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
