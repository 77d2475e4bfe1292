"""Data parallelism on the real engine: two ranks (two processes sharing cuda:0, gloo rendezvous on
127.0.0.1 — the same code path as one process per GPU over RCCL, minus the transport) each train one step on their half of a
global batch through Trainer / GradSync / DiTEngine.backward's on_segment hand-off, and must end with the parameters and the
gradient norm of a single rank that trains the whole batch with two microbatches on the same recorded noise.

Reference behaviour being matched: train.py:50 (global batch split over ranks), configs/res_256_pretrain.yaml:111-118
(microbatching + FSDP gradient averaging) — averaged-gradient data parallelism is numerically the same update.
Tolerance: the two runs differ only in fp32 summation order (and, for the bf16 exchange, in one bf16 rounding of the
reduced gradient): gradient norm within 1e-4 (2e-3 for bf16), updated parameters equal to 1e-5 absolute on >= 99.9 % of the
elements (the first AdamW update is lr * g / (|g| + eps): elements with |g| ~ 1e-8 are the rest)."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SEED, BATCH = 61, 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(world_batch_slice, exchange, microbatch, dp_mode="allreduce"):
    from oracle import microdit_ref as orc
    from micro_diffusion_amd import dit as mdit
    from micro_diffusion_amd.model import LatentDiffusion, _FrozenStub
    from micro_diffusion_amd.trainer import FusedAdamW, LRSchedule, Trainer
    cfg = orc.tiny_config()
    sd = orc.synth_state_dict(cfg, SEED)
    batch, rnd, epsn, mnoise = orc.synth_batch(cfg, BATCH, SEED + 1)
    d = mdit.DiT(**cfg.__dict__)
    d.load_state_dict(sd, strict=True)
    model = LatentDiffusion(d.to("cuda"), _FrozenStub("vae"), _FrozenStub("te"), _FrozenStub("tok"), train_mask_ratio=0.75)
    model.train()
    lo, hi = world_batch_slice
    calls = {"n": 0}

    def noise_fn(B):                      # recorded draws, consumed microbatch by microbatch (a later step starts over)
        a = lo + (calls["n"] * B) % (hi - lo)
        calls["n"] += 1
        return rnd[a:a + B].cuda(), epsn[a:a + B].cuda(), mnoise[a:a + B].cuda()
    model._noise_fn = noise_fn
    opt = FusedAdamW(model.dit, lr=2.4e-4)
    tr = Trainer(model, opt, LRSchedule("constant", alpha=1.0), clip_norm=0.25, microbatch_size=microbatch, exchange=exchange,
                 dp_mode=dp_mode)
    part = {k: v[lo:hi].cuda() for k, v in batch.items()}
    return model, opt, tr, part


def _rank_main(rank, world, port, exchange, out_path, dp_mode="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        per = BATCH // world
        model, opt, tr, part = _build((rank * per, (rank + 1) * per), exchange, per, dp_mode)
        assert tr.world == world and tr.sync.exchange == exchange and tr.sync.mode == dp_mode
        loss = tr.train_step(part)
        torch.cuda.synchronize()
        flat = model.dit.flat_buffers()
        shadow_ok = True
        if dp_mode == "sharded":
            # the bf16 weights every rank computes with are whole right after the step (all-gathered); the fp32 masters of the
            # other rank's chunks are stale until consolidate()
            tr.sync.wait_gather()
            mine_s = flat["s"].float().cpu()
            gs = [torch.empty_like(mine_s) for _ in range(world)]
            dist.all_gather(gs, mine_s)
            shadow_ok = all(torch.equal(gs[0], g) for g in gs)
            assert tr.stale_foreign_chunks
            tr.consolidate()
            shadow_ok = shadow_ok and torch.equal(flat["s"], flat["p"].to(torch.bfloat16)) and float(flat["g"].abs().max()) == 0.0
        # the exact replica checksum (md_checksum_u16 over the bf16 shadow per bucket, int64 MIN / MAX all-reduces): identical replicas
        # pass; ONE bf16 ulp in one weight of one rank must be seen (the fp32 sum of squares it replaces resolved ~4e-3)
        in_sync = tr.replicas_in_sync()
        if rank == 1:
            w = flat["s"].view(torch.int16)
            w[12345] += 1                                  # one ulp of one bf16 weight
        flipped_seen = not tr.replicas_in_sync()
        if rank == 1:
            flat["s"].view(torch.int16)[12345] -= 1
        shadow_ok = shadow_ok and in_sync and flipped_seen and tr.replicas_in_sync()
        # every rank must hold identical weights after the step (no broadcast ever happens)
        mine = flat["p"].detach().cpu()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        same = all(torch.equal(gathered[0], g) for g in gathered)
        if rank == 0:
            torch.save({"p": mine, "gnorm": float(opt.grad_norm().item()), "loss": float(loss), "ranks_identical": same and shadow_ok,
                        "buckets": len(tr.sync.bucket_list)}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,dp_mode", [("fp32", "allreduce"), ("bf16", "allreduce"), ("bf16", "sharded")])
def test_two_ranks_match_one_rank(hip, exchange, dp_mode):
    # ---- one rank, whole batch, two microbatches
    model, opt, tr, part = _build((0, BATCH), "fp32", BATCH // 2)
    tr.train_step(part)
    torch.cuda.synchronize()
    p1 = model.dit.flat_buffers()["p"].detach().cpu()
    g1 = float(opt.grad_norm().item())
    del model, opt, tr
    torch.cuda.empty_cache()
    # ---- two ranks, half the batch each
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rank0.pt")
        ctx = mp.get_context("spawn")
        port = _free_port()
        procs = [ctx.Process(target=_rank_main, args=(r, 2, port, exchange, out, dp_mode)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
        codes = [p.exitcode for p in procs]
        assert codes == [0, 0], f"rank processes failed: {codes}"
        r = torch.load(out)
    assert r["ranks_identical"], "replicas diverged within one step"
    assert r["buckets"] >= 4
    tol_n, tol_frac = (1e-4, 1e-3) if exchange == "fp32" else (2e-3, 1e-2)
    assert abs(r["gnorm"] - g1) <= tol_n * g1, (r["gnorm"], g1)
    bad = ((r["p"] - p1).abs() > 1e-5).float().mean().item()
    assert bad <= tol_frac, f"{bad:.2e} of the parameters differ by more than 1e-5 after one step"


def _rccl_single_rank_main(port, exchange, out_path, dp_mode="allreduce", transport="torch"):
    """One rank, backend "nccl" (= RCCL): every collective is the identity, but the calls are the production ones —
    asynchronous all-reduce on RCCL's stream behind an event on the compute stream, `work.wait()` as a stream dependency of
    the side stream that takes the bucket norms, the bf16 staging buffer feeding the optimiser kernel."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from micro_diffusion_amd.trainer import Trainer
        model, opt, tr, part = _build((0, BATCH), exchange, BATCH // 2)
        tr = Trainer(model, opt, tr.schedule, clip_norm=0.25, microbatch_size=BATCH // 2, exchange=exchange, single_rank_exchange=True,
                     dp_mode=dp_mode, transport=transport)
        assert tr.sync.enabled and not tr.sync.host_bounce and tr.sync.exchange == exchange and tr.sync.mode == dp_mode
        assert (tr.sync.comm is not None) == (transport == "native")
        seen = []
        inner = tr.sync._exchange
        tr.sync._exchange = lambda lo, hi: (seen.append((lo, hi)), inner(lo, hi))[1]
        loss = tr.train_step(part)
        dist.barrier()
        torch.cuda.synchronize()
        total = model.dit.flat_buffers()["total"]
        covered = sum(hi - lo for lo, hi in seen)
        res = {"p": model.dit.flat_buffers()["p"].detach().cpu(), "gnorm": float(opt.grad_norm().item()), "loss": float(loss),
               "covered": covered, "total": total, "buckets": len(seen),
               "g_zeroed": bool((model.dit.flat_buffers()["g"] == 0).all().item())}
        if dp_mode == "sharded":
            assert tr.sync.gather_work, "the bf16 weights are all-gathered asynchronously; the next forward waits per bucket"
            f = model.dit.flat_buffers()
            tr.sync.wait_gather()
            torch.cuda.synchronize()
            assert torch.equal(f["s"], f["p"].to(torch.bfloat16)), "gathered bf16 weights != round(fp32 masters)"
            loss2 = tr.train_step(part)          # a second step: the forward waits on the gathers bucket by bucket
            torch.cuda.synchronize()
            assert torch.isfinite(loss2).item()
        torch.save(res, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,dp_mode,transport", [("bf16", "allreduce", "torch"), ("fp32", "allreduce", "torch"), ("bf16", "sharded", "torch"),
                                                        ("bf16", "sharded", "native"), ("bf16", "allreduce", "native")])
def test_rccl_exchange_path_on_one_rank(hip, exchange, dp_mode, transport):
    """The RCCL transport itself needs N GPUs, which a 1-GPU box does not have; the code AROUND it (everything GradSync does
    under backend "nccl" that the gloo test above replaces by a host bounce) runs here on a one-rank communicator and must
    reproduce the step without any exchange."""
    model, opt, tr, part = _build((0, BATCH), "fp32", BATCH // 2)
    tr.train_step(part)
    torch.cuda.synchronize()
    p1 = model.dit.flat_buffers()["p"].detach().cpu()
    g1 = float(opt.grad_norm().item())
    del model, opt, tr
    torch.cuda.empty_cache()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "r.pt")
        ctx = mp.get_context("spawn")
        proc = ctx.Process(target=_rccl_single_rank_main, args=(_free_port(), exchange, out, dp_mode, transport))
        proc.start()
        proc.join(600)
        assert proc.exitcode == 0, f"RCCL single-rank process failed: {proc.exitcode}"
        r = torch.load(out)
    assert r["covered"] == r["total"] and r["buckets"] >= 4, "every gradient element goes through exactly one bucket"
    assert r["g_zeroed"], "the optimiser pass zeroes the fp32 accumulators also when it reads the bf16 exchange buffer"
    tol_n, tol_frac = (1e-4, 1e-3) if exchange == "fp32" else (2e-3, 1e-2)
    assert abs(r["gnorm"] - g1) <= tol_n * g1, (r["gnorm"], g1)
    bad = ((r["p"] - p1).abs() > 1e-5).float().mean().item()
    assert bad <= tol_frac, f"{bad:.2e} of the parameters differ by more than 1e-5 after one step"


def _one_microbatch_main(port, out_path):
    """One rank over RCCL, sharded exchange, the rank's whole batch as ONE microbatch (a rank of the 8-GPU run: res_256_pretrain.yaml:24,111):
    the step with weight gradients stored straight into the bf16 exchange buffer against the same step through the fp32 accumulators."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from micro_diffusion_amd.trainer import Trainer
        res = {}
        for store in (False, True):
            model, opt, tr, part = _build((0, BATCH), "bf16", BATCH)
            tr = Trainer(model, opt, tr.schedule, clip_norm=0.25, microbatch_size=BATCH, exchange="bf16", single_rank_exchange=True,
                         dp_mode="sharded")
            tr.sync.store_bf16 = store
            loss = tr.train_step(part)
            torch.cuda.synchronize()
            f = model.dit.flat_buffers()
            res[store] = {"gbf": tr.sync.gbf.detach().float().cpu(), "stored": tr.sync.last_stored, "loss": float(loss),
                          "gnorm": float(opt.grad_norm().item()), "g_zero": bool((f["g"] == 0).all().item())}
            tr.sync.wait_gather()
            torch.cuda.synchronize()
            res[store]["p"] = f["p"].detach().cpu()
            res[store]["n_matrix"] = sum(1 for n, v in f["P"].items() if v.dim() >= 2)
            loss2 = tr.train_step(part)            # a second step on the same buffers (nothing stale in gbf, accumulators still clean)
            torch.cuda.synchronize()
            res[store]["loss2"] = float(loss2)
            res[store]["g_zero2"] = bool((f["g"] == 0).all().item())
            del model, opt, tr
            torch.cuda.empty_cache()
        torch.save(res, out_path)
    finally:
        dist.destroy_process_group()


def test_one_microbatch_step_stores_bf16_gradients(hip):
    """DiTEngine.wgrad_bf16 / GradSync.begin_backward: with one microbatch per step the split-K reduction (or the GEMM epilogue) writes
    bf16 weight gradients into the exchange buffer; the fp32 accumulators stay untouched (zero), the cast + clear pass runs only for
    what was not stored.  The staged gradient must equal the accumulate-then-cast one up to the bf16 rounding of a differently ordered
    fp32 sum, and the updated weights must agree."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "r.pt")
        ctx = mp.get_context("spawn")
        proc = ctx.Process(target=_one_microbatch_main, args=(_free_port(), out))
        proc.start()
        proc.join(600)
        assert proc.exitcode == 0, f"process failed: {proc.exitcode}"
        r = torch.load(out)
    a, b = r[False], r[True]
    assert a["stored"] == 0 and b["stored"] >= 0.8 * b["n_matrix"], (a["stored"], b["stored"], b["n_matrix"])
    assert a["g_zero"] and b["g_zero"] and a["g_zero2"] and b["g_zero2"]
    assert abs(a["loss"] - b["loss"]) <= 1e-6 * abs(a["loss"])               # the forward is the same
    d = (a["gbf"] - b["gbf"]).abs()
    scale = a["gbf"].abs().clamp_min(1e-12)
    assert float((d > 0.0079 * scale).float().mean()) <= 1e-4, "more than 1e-4 of the staged gradient differs by more than one bf16 ulp"
    assert float((d.double().pow(2).sum() / a["gbf"].double().pow(2).sum()).sqrt()) <= 2e-3
    assert abs(a["gnorm"] - b["gnorm"]) <= 1e-3 * a["gnorm"]
    bad = ((a["p"] - b["p"]).abs() > 1e-5).float().mean().item()
    assert bad <= 1e-2, bad
    assert abs(a["loss2"] - b["loss2"]) <= 2e-3 * abs(a["loss2"])


def _comm_direct_main(port, out_path):
    """libmicrodit_comm.so by itself on a one-rank communicator: every collective is the identity, what is checked is the stream
    contract -- a collective runs behind the kernels already enqueued on the caller's stream, the caller's later kernels run
    behind wait(ticket), tickets are refused once recycled."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    from micro_diffusion_amd import comm
    c = comm.Comm(comm.Comm.unique_id(), 0, 1, 0)
    n = 1 << 24
    a = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    ok = True
    for it in range(4):
        a.fill_(float(it + 1))                                   # compute stream: produce the bucket
        t = c.all_reduce(a)                                      # comm stream, behind the fill
        t.wait()                                                 # compute stream behind the collective
        b = a.float().sum()                                      # consumer on the compute stream
        ok = ok and float(b) == float(it + 1) * n
    src = torch.arange(1024, device="cuda", dtype=torch.float32)
    dst = torch.zeros(1024, device="cuda")
    c.reduce_scatter(dst, src).wait()
    g = torch.zeros(1024, device="cuda")
    c.all_gather(g, dst).wait()
    torch.cuda.synchronize()
    ok = ok and torch.equal(dst, src) and torch.equal(g, src)
    first = comm.Ticket(c, 1)
    for _ in range(1100):                                        # more collectives than the ticket ring holds
        last = c.all_reduce(dst)
    last.wait()
    stale = comm.lib().md_comm_wait(c.handle, first.ticket, torch.cuda.current_stream().cuda_stream)
    c.synchronize()
    c.destroy()
    torch.save({"ok": bool(ok), "stale_rc": int(stale)}, out_path)


def test_md_comm_stream_contract_on_one_rank(hip):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "r.pt")
        ctx = mp.get_context("spawn")
        proc = ctx.Process(target=_comm_direct_main, args=(_free_port(), out))
        proc.start()
        proc.join(300)
        assert proc.exitcode == 0, f"md_comm process failed: {proc.exitcode}"
        r = torch.load(out)
    assert r["ok"], "a collective did not see the bucket its stream order promises, or the consumer ran ahead of it"
    assert r["stale_rc"] == -1, "a ticket whose event was recycled must be refused"
