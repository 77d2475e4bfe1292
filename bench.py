"""Headline benchmark: training images/sec of MicroDiT-XL/2 (res_256_pretrain: 32x32x4 latents, mask 0.75, bf16,
global batch 2048) on N MI355X GPUs — one full optimisation step per "step": all microbatches fwd+bwd, gradient
exchange (N > 1), clip, AdamW.  Prints ONE JSON line (contract in the task description).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: spawns its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Extra keys of the line (the headline fields are unchanged by them):
  other_stages   the same full step on the reference's other stage configs (BASELINE.json configs[3], [4]):
                 res_256_finetune (mask 0: 256 backbone tokens) and res_512_pretrain (64x64 latents, pos_interp_scale 2),
                 1 warm-up + 5 timed steps each at microbatch 256 (their YAMLs say 64 / 32: stated per stage), N = 1 only
                 (--no-other-stages skips them);
  value_mb256    the headline step with the YAML microbatch (256 = the per-rank shape of an 8-GPU run), N = 1 only;
  value_mb256_cu248   the same with every persistent-GEMM grid limited to the 248 CUs an 8-channel RCCL kernel leaves;
  rank_of_8_step, n8_ceiling   ONE 256-image microbatch per step through the production sharded exchange on a one-rank RCCL communicator, 248 CUs
                 (the rank-of-8 step emulated on one GPU) and 8 x that / the headline;
  roofline       dominant kernel (the MFMA GEMM family): flop / per-launch HIP-event time, plus HBM traffic per launch from
                 the committed rocprofv3 PMC passes (profiles/r6_gemm_traffic.json: counters need rocprofv3 around the process,
                 so they are NOT measured by this run -- `traffic_measured_in_run` false; the file carries the source hash of
                 the library it was measured on and `traffic` is null when that differs from the running build);
  roofline_hbm   the bandwidth-bound kernel classes (attention, LayerNorm, QK-LayerNorm, SwiGLU, gate backward, split-K reduce):
                 algorithmic bytes / per-launch HIP-event time against the 8 TB/s HBM3E peak AND against the streaming envelope
                 measured on this chip for the class's read : write mix (scripts/hbm_envelope.hip), from the same profiling step;
  dp             (N > 1) ranks verified by an all-reduce of ones over RCCL, exchange format, buckets, and the time the
                 compute stream waited for the gradient exchange after the last backward kernel (`exposed_comm_ms`);
  cpu_baseline   the CPU restatement of the reference step on the host cores (kind "port": the reference is pure Python and
                 /root/reference does not travel to the GPU box), >= 3 timed steps, threads used and host cores stated.
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FWD_BWD_GFLOP_PER_IMG = {("res256", 0.75): 282.3, ("res256", 0.0): 714.4, ("res512", 0.75): 1069.4, ("res512", 0.0): 3002.8}
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md (dense; AMD's 5 PF figure is 2:1 sparse)
# The reference's training stages (configs/*.yaml; BASELINE.json configs[1], [3], [4]).  `microbatch` is this repo's
# HBM-sized default per stage (the YAML values 256 / 64 / 32 are 80 GB-H100 settings; any split accumulates the same gradient).
STAGES = {
    "res_256_pretrain": dict(latent_res=32, pos_interp_scale=1.0, mask=0.75, p_mean=-0.6, p_std=1.2, lr=2.4e-4, clip=0.25,
                             microbatch=1024, yaml_microbatch=256, key=("res256", 0.75),
                             sched=("cosine_with_warmup", dict(t_warmup="2500ba", t_max="250000ba", alpha_f=0.33))),
    "res_256_finetune": dict(latent_res=32, pos_interp_scale=1.0, mask=0.0, p_mean=-0.6, p_std=1.2, lr=8e-5, clip=0.25,
                             microbatch=256, yaml_microbatch=64, key=("res256", 0.0), sched=("constant", dict(alpha=1.0))),
    "res_512_pretrain": dict(latent_res=64, pos_interp_scale=2.0, mask=0.75, p_mean=0.0, p_std=0.6, lr=8e-5, clip=0.5,
                             microbatch=256, yaml_microbatch=32, key=("res512", 0.75), sched=("constant_with_warmup", dict(t_warmup="500ba", alpha=1.0))),
}


def dezero_(dit, seed=1234):
    """The reference init zeroes every adaLN / final-layer / caption-block output weight, which makes most of the
    backward numerically trivial.  The benchmark perturbs those tensors so every kernel sees
    full-range data (zero operands let the chip clock higher and would flatter the number)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for name, p in dit.named_parameters():
            if float(p.abs().max()) == 0.0:
                p.add_(torch.randn(p.shape, device=p.device, generator=g) * 0.02)


HBM_PEAK_TBS = 8.0                  # /opt/skills/guides/MI355X_MICROARCH.md (HBM3E)
# What the same HBM delivers to STREAMING kernels, measured on this chip with no arithmetic at all (scripts/hbm_envelope.hip,
# profiles/r4_hbm_envelope.txt: 16-byte accesses, 1 GiB per stream, several loads in flight per lane): the practical ceiling of a
# bandwidth-bound kernel depends on its read : write mix, so every class below is also reported as a fraction of ITS envelope.
HBM_ENVELOPE_TBS = {"read_only": 6.26, "write_only": 4.59, "copy_1r_1w": 4.95, "rows_1r_1w": 5.53, "2r_1w": 5.64, "3r_2w": 5.46}
KERNEL_ENVELOPE = {"layernorm": "rows_1r_1w", "qk_layernorm": "rows_1r_1w", "swiglu": "3r_2w", "gate_bwd": "2r_1w", "splitk_reduce": "read_only",
                   "attention": "3r_2w"}
OTHER_STAGE_STEPS = 5              # timed steps of the res_256_finetune / res_512_pretrain legs (after one warm-up)
CPU_BASELINE_THREADS_CAP = 32      # torch CPU ops with hundreds of threads on small tensors oversubscribe badly
CPU_BASELINE_TIMEOUT_S = 300
CPU_BASELINE_STEPS = 3             # timed steps after one warm-up


def _cpu_baseline_worker():
    """Runs in a child process: the oracle (CPU restatement of the reference step: fwd + bwd + clip + AdamW) on the host
    cores, MicroDiT-XL/2, batch 4, 1 warm-up + CPU_BASELINE_STEPS timed steps.  Prints one JSON object."""
    from oracle import microdit_ref as orc
    threads = max(1, min(os.cpu_count() or 1, CPU_BASELINE_THREADS_CAP))
    torch.set_num_threads(threads)
    cfg = orc.xl2_config()
    sd = orc.synth_state_dict(cfg, 3)
    names = [k for k in sd if k not in ("pos_embed", "mask_token")]
    for k in names:
        sd[k].requires_grad_(True)
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    B = 4
    batch, rnd, epsn, mnoise = orc.synth_batch(cfg, B, 4)
    times = []
    for step in range(1, CPU_BASELINE_STEPS + 2):
        ts = time.time()
        loss = orc.latent_diffusion_forward(sd, cfg, batch, rnd, epsn, mnoise, 0.75, -0.6, 1.2)
        loss.backward()
        with torch.no_grad():
            grads = [sd[k].grad for k in names]
            orc.clip_grad_norm(grads, 0.25)
            for k in names:
                orc.adamw_step(sd[k], sd[k].grad, m[k], v[k], step, 2.4e-4)
                sd[k].grad = None
        times.append(time.time() - ts)
    per = sum(times[1:]) / len(times[1:])          # first step = warm-up
    print(json.dumps({"value": B / per, "unit": "images/sec", "cores": threads, "host_cores": os.cpu_count(),
                      "cores_note": f"{threads} threads (cap {CPU_BASELINE_THREADS_CAP}) of {os.cpu_count()} host cores", "kind": "port",
                      "sample": f"oracle/microdit_ref.py (CPU fp32 restatement of the reference step: LatentDiffusion.forward + backward + "
                                f"clip_grad_norm_ + AdamW, pinned to the reference by tests/golden/xl2_mask75.npz) MicroDiT-XL/2 mask=0.75, "
                                f"batch {B}, {len(times) - 1} timed steps after 1 warm-up, {per:.2f} s/step, torch.set_num_threads({threads}) "
                                f"on a {os.cpu_count()}-core host; the reference itself is Python under /root/reference, which does not "
                                f"exist on the GPU box"}), flush=True)


def cpu_baseline():
    """Bounded CPU leg: a child process with a hard timeout, so the benchmark can never hang on the host side."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True, text=True,
                           timeout=CPU_BASELINE_TIMEOUT_S, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        raise RuntimeError((r.stderr or r.stdout)[-300:])
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "images/sec", "cores": min(os.cpu_count() or 1, CPU_BASELINE_THREADS_CAP),
                "host_cores": os.cpu_count(), "kind": "port",
                "sample": f"oracle XL/2 batch 4 did not finish {CPU_BASELINE_STEPS + 1} steps within {CPU_BASELINE_TIMEOUT_S} s on this host"}


class Stage:
    """Model + optimiser + synthetic rank batch of one training stage; `step()` = one full optimisation step."""

    def __init__(self, name, arch, global_batch, microbatch, world, rank):
        from micro_diffusion_amd.model import create_latent_diffusion
        from micro_diffusion_amd.trainer import FusedAdamW, LRSchedule, Trainer
        st = STAGES[name]
        self.name, self.st = name, st
        torch.manual_seed(18)                       # configs/*.yaml: seed 18 (same init on every rank)
        self.model = create_latent_diffusion(dit_arch=arch, latent_res=st["latent_res"], in_channels=4, pos_interp_scale=st["pos_interp_scale"],
                                             dtype="bfloat16", precomputed_latents=True, p_mean=st["p_mean"], p_std=st["p_std"],
                                             train_mask_ratio=st["mask"])
        self.model.dit.to("cuda")
        dezero_(self.model.dit)
        self.model.train()
        opt = FusedAdamW(self.model.dit, lr=st["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
        kind, kw = st["sched"]
        self.trainer = Trainer(self.model, opt, LRSchedule(kind, **kw), clip_norm=st["clip"], microbatch_size=microbatch)
        self.trainer.batches_seen = 100          # a non-zero LR (the warm-up schedules' first batch runs at lr = 0)
        per_rank = global_batch // world
        self.per_rank, self.microbatch = per_rank, min(microbatch, per_rank)
        torch.manual_seed(2024 + rank)           # data / noise stream differs per rank (Composer seeds rank-wise)
        g = torch.Generator(device="cuda").manual_seed(2024 + rank)
        r = st["latent_res"]
        self.batch = {
            "image_latents": (torch.randn(per_rank, 4, r, r, device="cuda", generator=g) * 0.8).half(),
            "caption_latents": torch.randn(per_rank, 1, 77, 1024, device="cuda", generator=g).half(),
            "drop_caption_mask": (torch.rand(per_rank, device="cuda", generator=g) >= 0.1).float(),
        }

    def step(self):
        return self.trainer.train_step(self.batch)       # the caption-drop mask is applied inside the first kernel, not in place

    def timed(self, steps, warmup, world):
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = None
        for _ in range(steps):
            loss = self.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, float(loss.item())

    def close(self):
        eng = self.model.dit.engine
        eng._tape_arena.buf = eng._scratch_arena.buf = None      # the activation arenas (156 GB at microbatch 1024) go first,
        eng._tape_arena.peaks.clear()                            # whoever else still holds the engine
        eng._scratch_arena.peaks.clear()
        eng.ws = None
        eng = None
        self.model = self.trainer = self.batch = None
        gc.collect()
        torch.cuda.empty_cache()


def rank_of_8_leg(head, steps=6, pretend_world=8, cu_limit=248):
    """One rank of the 8-GPU run on ONE GPU (see the caller): returns {images_per_s (this rank's share), ms_per_step, ...}.  Uses the
    headline stage's model / optimiser with a second Trainer (sharded bf16 exchange over a one-rank RCCL communicator)."""
    from micro_diffusion_amd.trainer import Trainer
    made_pg = False
    if not dist.is_initialized():
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
        made_pg = True
    old_tr, old_batch, old_per = head.trainer, head.batch, head.per_rank
    eng = head.model.dit.engine
    saved_fn = eng.cu_limit_fn
    try:
        tr = Trainer(head.model, old_tr.opt, old_tr.schedule, clip_norm=old_tr.clip_norm, microbatch_size=256, exchange="bf16",
                     single_rank_exchange=True, dp_mode="sharded")
        tr.batches_seen = 100
        tr.shard_chunk_of = pretend_world
        tr.measure_comm = True
        eng.cu_limit_fn = lambda: cu_limit
        for ar in (eng._tape_arena, eng._scratch_arena):     # another launch sequence: the arenas are re-measured on its first pass
            ar.peaks.clear()
            ar.buf = None
        head.trainer, head.batch, head.per_rank = tr, {k: v[:256] for k, v in old_batch.items()}, 256
        e, loss = head.timed(steps, 2, 1)
        res = {"images_per_s": 256 * steps / e, "ms_per_step": e / steps * 1e3, "steps": steps, "per_rank_batch": 256, "microbatches_per_step": 1,
               "cu_limit": cu_limit, "adamw_share": f"1/{pretend_world} of every bucket", "gradient_launches_stored_as_bf16": tr.sync.last_stored,
               "optimizer_ms": tr.optimizer_ms(last=steps), "exchange_wait_ms": tr.exposed_comm_ms(last=steps), "loss": loss,
               "exchange": tr.sync.describe(), "collectives": "identities on a one-rank RCCL communicator (no byte crosses xGMI)"}
        tr.sync.wait_gather()
        tr.consolidate()
        return res
    finally:
        eng.cu_limit_fn = saved_fn
        head.trainer, head.batch, head.per_rank = old_tr, old_batch, old_per
        head.model.dit._on_segment = old_tr.sync.on_segment
        old_tr.sync.norm_partials = old_tr.opt.partials          # (ensure_norm_slots may have re-allocated them)
        for ar in (eng._tape_arena, eng._scratch_arena):
            ar.peaks.clear()
            ar.buf = None
        eng.before_segment = old_tr.sync.wait_gather if old_tr.sharded else None
        head.model.dit.shadow_is_authoritative = False
        if made_pg:
            torch.cuda.synchronize()
            dist.destroy_process_group()


TRAFFIC_FILE = "r6_gemm_traffic.json"


def gemm_traffic():
    """HBM bytes per GEMM launch of the headline step from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate --pmc runs, calibrated on a 1 GiB copy of the same pass; scripts/pmc_workload.py + scripts/pmc_traffic.py write the
    file).  The counters need rocprofv3 around the process, so they are NOT measured by this run; the number is returned only
    when the file was measured on THIS build of the GEMM kernels (hash of the GEMM sources stamped into it), else null + `stale`."""
    from micro_diffusion_amd import hip
    path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if not os.path.exists(path):
        return None, {"file": None, "traffic_measured_in_run": False}
    with open(path) as fh:
        t = json.load(fh)
    info = {k: v for k, v in t.items() if k != "per_kernel"}
    # keyed to the GEMM sources (gemm*.hip + headers + flags); files written before that key existed carry the whole-library hash
    if "gemm_source_hash" in t:
        stale = t["gemm_source_hash"] != hip._gemm_source_hash()
    else:
        stale = t.get("library_source_hash") != hip._source_hash()
    info.update(file="profiles/" + TRAFFIC_FILE, traffic_measured_in_run=False, running_build=hip._source_hash(),
                running_gemm_sources=hip._gemm_source_hash(), stale=stale)
    return (None if info["stale"] else t.get("bytes_per_launch")), info


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch_command(n, argv):
    """argv of the launcher `python bench.py --gpus N` (no WORLD_SIZE in the environment) re-executes itself under: N ranks of
    this script on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", port, os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--global-batch", type=int, default=2048)
    ap.add_argument("--microbatch", type=int, default=1024,
                    help="gradient-accumulation microbatch per rank.  res_256_pretrain.yaml says 256, which is an 80 GB-H100 "
                         "memory setting (Composer also accepts 'auto'); the accumulated gradient of the rank batch is the same "
                         "for any split, and 288 GB of HBM3E holds 1024 (205 GB peak at N=1)")
    ap.add_argument("--arch", default="MicroDiT_XL_2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-other-stages", action="store_true", help="skip the res_256_finetune / res_512_pretrain / microbatch-256 legs")
    ap.add_argument("--attn-bwd", default="auto", choices=["auto", "pair", "fused1", "fused2", "fused2s", "stream"],
                    help="A/B runs: force one attention-backward kernel wherever it covers the shape (default: the library's rule)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rank-probe", action="store_true", help=argparse.SUPPRESS)   # tests: every rank reports itself and exits (no GPU needed)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        _cpu_baseline_worker()
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` is ONE command per node, like the reference's `composer train.py ...`
        # (/root/reference/train_e2e.sh:10): it spawns its own N ranks (one process per GPU) and the ranks take the
        # torch.distributed.run branch below.  exec, not a child: rank 0's JSON line is this command's stdout.
        os.execv(sys.executable, self_launch_command(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rank_probe:
        print(json.dumps({"rank_probe": rank, "world": world, "local_rank": local_rank, "master": os.environ.get("MASTER_ADDR")}), flush=True)
        return
    # The JSON line must be the ONLY thing on this command's stdout.  RCCL printf()s a version banner when a communicator is made
    # (the rank_of_8_step leg makes one at N = 1 too) and stdio holds it until exit, i.e. AFTER the line: file descriptor 1 is
    # pointed at stderr for the run and the line is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus N` (it spawns its ranks) or "
                         f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    ndev = torch.cuda.device_count()
    # MD_DIST_BACKEND=gloo lets several ranks share one GPU (functional test of the distributed path on a 1-GPU box)
    backend = os.environ.get("MD_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank % ndev)
    dp = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from micro_diffusion_amd.trainer import cap_rccl_channels
        channels = cap_rccl_channels()     # NCCL_MAX_NCHANNELS, before RCCL starts (an exported value wins)
        if backend == "nccl":
            if ndev < world:
                raise SystemExit(f"--gpus {world} over RCCL needs {world} visible GPUs, found {ndev} "
                                 "(MD_DIST_BACKEND=gloo runs a functional test of several ranks on one GPU)")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % ndev))
        else:
            dist.init_process_group(backend)
        if dist.get_backend() != backend:
            raise SystemExit(f"process group came up on '{dist.get_backend()}', not '{backend}'")
        # every rank really is on the wire: an all-reduce of ones must count them
        ones = torch.ones(1, device="cuda")
        if backend == "nccl":
            dist.all_reduce(ones)
        else:
            h = ones.cpu()
            dist.all_reduce(h)
            ones = h
        counted = int(ones.item())
        if counted != world:
            raise SystemExit(f"all-reduce of ones returned {counted}, expected {world} ranks")
        dp = {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "rccl_ranks": counted if backend == "nccl" else 0,
              "ranks": counted, "devices_visible": ndev, "rccl_max_channels": channels}

    head = Stage("res_256_pretrain", args.arch, args.global_batch, args.microbatch, world, rank)
    head.trainer.measure_comm = world > 1
    if args.attn_bwd != "auto":
        head.model.dit.engine.attn_bwd_prefer = {"pair": 1, "fused1": 2, "fused2": 3, "fused2s": 4, "stream": 5}[args.attn_bwd]
    elapsed, loss = head.timed(args.steps, args.warmup, world)
    ms_per_step = elapsed / args.steps * 1e3
    value = args.global_batch * args.steps / elapsed
    gf = FWD_BWD_GFLOP_PER_IMG[("res256", 0.75)]
    out = {
        "metric": "training images/sec (global batch 2048) MicroDiT-XL/2 256-res mask=0.75",
        "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic (N(0,1)*0.8 fp16 latents 32x32x4, N(0,1) fp16 captions 77x1024, 10% caption drop; "
                "reference init seed 18 with zero-init tensors perturbed)",
        "config": {"workload": f"{args.arch} res_256_pretrain.yaml mask=0.75, full step = "
                               f"{head.per_rank // head.microbatch} microbatches x {head.microbatch} fwd+bwd"
                               " + grad exchange + clip 0.25 + AdamW",
                   "global_batch": args.global_batch, "microbatch": head.microbatch, "parallelism": f"dp{world}",
                   "grad_exchange": head.trainer.sync.describe() if hasattr(head.trainer.sync, "describe") else None},
        "loss": loss,
        "step_mfma_frac": value / world * gf / 1e3 / MFMA_BF16_DENSE_PEAK_TFLOPS,
    }
    if dp is not None:
        sync = head.trainer.sync
        ex = head.trainer.exposed_comm_ms(last=args.steps)
        dp.update(mode=sync.mode, transport=("md_comm (libmicrodit_comm.so over RCCL)" if sync.comm is not None else "torch.distributed"),
                  exchange_dtype=sync.exchange, buckets=sync.last_buckets, bytes_per_step=sync.last_bytes,
                  optimizer_ms=head.trainer.optimizer_ms(last=args.steps),
                  gemm_cu_limit_while_comm_in_flight=(256 - head.trainer.rccl_channels) if head.trainer.rccl_channels else None,
                  exposed_comm_ms=ex, exposed_comm_share=(ex / ms_per_step if ex is not None else None),
                  note="exposed_comm_ms = time the compute stream waited for the gradient exchange (and the side-stream bucket "
                       "norms) after the last backward kernel, mean over the timed steps of rank 0; criterion for moving the "
                       "exchange into an md_comm_* C ABI with reduce-scatter + all-gather over all 7 xGMI links: > 10 ms at N = 8")
        out["dp"] = dp

    if args.attn_bwd != "auto":
        out["config"]["attn_bwd_forced"] = args.attn_bwd
    if rank == 0 and not args.no_profile:
        # ---- roofline leg: per-launch HIP events around every launch of the dominant kernel (the MFMA GEMM) in one
        # extra, untimed step (events are recorded on the stream the kernels are launched on).
        eng = head.model.dit.engine
        eng.gemm_profile = []
        eng.kernel_profile = {}
        if world == 1:
            head.step()
        else:   # profile a single microbatch locally, without collectives
            part = {k: v[:head.microbatch] for k, v in head.batch.items()}
            head.trainer.sync.active = False
            head.model.train_microbatch(part)
        torch.cuda.synchronize()
        prof, eng.gemm_profile = eng.gemm_profile, None
        tot_ms = sum(r[0].elapsed_time(r[1]) for r in prof)
        tot_fl = sum(r[2] for r in prof)
        tot_by = sum(r[4] for r in prof)
        n = len(prof)
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        traffic, tinfo = gemm_traffic()
        out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
                           "kernel": "md_gemm_bf16 family (gemm_bf16_w4_kernel + gemm_bf16_pp_kernel + gemm_bf16_kernel + gemm_bf16_dma_kernel)",
                           "launches": n, "avg_launch_us": tot_ms * 1e3 / n, "gflop_per_launch": tot_fl / n / 1e9,
                           "gemm_time_share_of_step": (tot_ms / ms_per_step) if world == 1 else None,
                           "algorithmic_bytes_per_launch": tot_by / n,
                           "traffic_over_algorithmic": (traffic / (tot_by / n)) if traffic else None,
                           "traffic_measured_in_run": False, "traffic_source": tinfo}
        # ---- HBM side: the largest bandwidth-bound kernel classes of the same profiling step
        hb = {}
        for name, recs in (eng.kernel_profile or {}).items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            byt = sum(r[2] for r in recs)
            if ms > 0:
                tbs = byt / (ms * 1e-3) / 1e12
                env = KERNEL_ENVELOPE.get(name)
                hb[name] = {"launches": len(recs), "total_ms": ms, "algorithmic_gb": byt / 1e9, "achieved_tbs": tbs,
                            "frac_of_hbm_peak": tbs / HBM_PEAK_TBS,
                            "streaming_envelope": env, "frac_of_streaming_envelope": (tbs / HBM_ENVELOPE_TBS[env]) if env else None,
                            "share_of_step": (ms / ms_per_step) if world == 1 else None}
        eng.kernel_profile = None
        eng = None
        out["roofline_hbm"] = {"bound": "hbm", "peak": HBM_PEAK_TBS, "unit": "TB/s",
                               "streaming_envelope_tbs": HBM_ENVELOPE_TBS,
                               "streaming_envelope_source": "scripts/hbm_envelope.hip on MI355X, profiles/r4_hbm_envelope.txt (no-arithmetic kernels of the same access shapes)",
                               "kernels": dict(sorted(hb.items(), key=lambda kv: -kv[1]["total_ms"]))}
    if world == 1 and not args.no_other_stages:
        # ---- the YAML microbatch (the per-rank shape of an 8-GPU run) on the same model
        head.trainer.microbatch_size = 256
        e256, _ = head.timed(2, 1, 1)
        out["value_mb256"] = args.global_batch * 2 / e256
        # the same headline step at the YAML's device_train_microbatch_size (configs/res_256_pretrain.yaml: 256 = the per-rank
        # shape of an 8-GPU run): the N = 1 figure an N = 8 scaling ratio has to be read against
        out["config"]["yaml_microbatch"] = 256
        out["config"]["value_at_yaml_microbatch"] = out["value_mb256"]
        # ---- the rank-of-8 step emulated on ONE GPU.  (i) value_mb256_cu248 (round-5 definition, kept for continuity): the same
        # 256-image microbatches, 8 per step, with every persistent-GEMM grid limited to the 248 CUs an 8-channel RCCL kernel leaves.
        # (ii) rank_of_8_step (round 6): what a rank of the 8-GPU run really executes -- ONE 256-image microbatch per optimiser step
        # (configs/res_256_pretrain.yaml:24,111: 2048 / 8 = device_train_microbatch_size), the production sharded exchange on a
        # one-rank RCCL communicator (every collective an identity; staging, side-stream norms, stream waits as shipped), weight
        # gradients stored straight into the bf16 exchange buffer (DiTEngine.wgrad_bf16), AdamW over 1 / 8 of every bucket, all of
        # it on 248 CUs.  n8_ceiling = 8 x rank_of_8_step / value: the scaling an 8-GPU run can reach before a single exposed byte.
        eng = head.model.dit.engine
        saved_fn = eng.cu_limit_fn
        eng.cu_limit_fn = lambda: 248
        e248, _ = head.timed(2, 1, 1)
        out["value_mb256_cu248"] = args.global_batch * 2 / e248
        eng.cu_limit_fn = saved_fn
        out["n8_ceiling_r5_definition"] = 8.0 * out["value_mb256_cu248"] / value
        try:
            out["rank_of_8_step"] = rank_of_8_leg(head)
            out["n8_ceiling"] = 8.0 * out["rank_of_8_step"]["images_per_s"] / value
            out["n8_ceiling_note"] = ("8 x rank_of_8_step.images_per_s / value: 8 ranks each running their one-microbatch step on 248 CUs "
                                      "(sharded exchange path on a one-rank communicator), against the 1-GPU headline (microbatch %d); the "
                                      "round-5 definition (8 x value_mb256_cu248 / value) is n8_ceiling_r5_definition" % args.microbatch)
        except Exception as e:          # no RCCL communicator on this box: keep the round-5 number, say so
            out["rank_of_8_step"] = {"images_per_s": None, "error": f"{type(e).__name__}: {e}"}
            out["n8_ceiling"] = out["n8_ceiling_r5_definition"]
            out["n8_ceiling_note"] = "rank_of_8_step failed: 8 x value_mb256_cu248 / value (round-5 definition)"
        eng = None
        head.trainer.microbatch_size = args.microbatch
    head.close()
    if world == 1 and not args.no_other_stages:
        other = {}
        for name in ("res_256_finetune", "res_512_pretrain"):
            st = Stage(name, args.arch, args.global_batch, STAGES[name]["microbatch"], 1, 0)
            e, l = st.timed(OTHER_STAGE_STEPS, 1, 1)
            v = args.global_batch * OTHER_STAGE_STEPS / e
            other[name] = {"value": v, "unit": "images/sec", "ms_per_step": e / OTHER_STAGE_STEPS * 1e3, "steps": OTHER_STAGE_STEPS, "warmup": 1,
                           "microbatch": st.microbatch, "yaml_microbatch": STAGES[name]["yaml_microbatch"],
                           "microbatch_note": "the YAML's device_train_microbatch_size is an 80 GB setting; gradient accumulation is split-invariant",
                           "loss": l, "gflop_per_image": FWD_BWD_GFLOP_PER_IMG[STAGES[name]["key"]],
                           "step_mfma_frac": v * FWD_BWD_GFLOP_PER_IMG[STAGES[name]["key"]] / 1e3 / MFMA_BF16_DENSE_PEAK_TFLOPS}
            st.close()
        out["other_stages"] = other
    if rank == 0 and world > 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": None, "kind": "port",
                               "sample": "measured at N=1 only (the other ranks would idle behind it)"}
    elif rank == 0 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # host too small for the XL/2 oracle: report, do not fail the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
