"""CPU-side checks: (1) the C-ABI library builds for gfx950, loads, and exports every symbol include/microdit_hip.h
declares (no compute without a GPU); (2) the data-parallel gradient bucketing reduces every element of the flat
gradient buffer exactly once — world_size 2, gloo backend, 127.0.0.1."""
import ctypes
import os
import re
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from micro_diffusion_amd import hip
    path = hip.build()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "microdit_hip.h")).read()
    declared = set(re.findall(r"^int\s+(md_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(hip.exported_symbols()), declared ^ set(hip.exported_symbols())
    assert lib.md_abi_version() == 6 == hip.ABI_VERSION
    assert re.search(r"#define MD_ABI_VERSION 6\b", header)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under micro_diffusion_amd/ (nor train.py) may import it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|import_module\([\"']oracle", re.M)
    files = [os.path.join(ROOT, "train.py")]
    for base, _, fs in os.walk(os.path.join(ROOT, "micro_diffusion_amd")):
        files += [os.path.join(base, f) for f in fs if f.endswith(".py")]
    for base, _, fs in os.walk(os.path.join(ROOT, "micro_diffusion")):
        files += [os.path.join(base, f) for f in fs if f.endswith(".py")]
    for f in files:
        assert not pat.search(open(f).read()), f"{f} imports the oracle"


class _FakeDiT:
    """Flat CPU buffers with the real layout of a small model (GradSync only needs the table and the buffers)."""

    def __init__(self, rank):
        from micro_diffusion_amd.arch import DiTConfig, param_table
        from micro_diffusion_amd.dit import flat_layout
        from oracle import microdit_ref as orc
        c = orc.micro_config()
        self._table = param_table(DiTConfig(**c.__dict__))
        offs, total = flat_layout(self._table)
        self._flat = {"offs": offs, "total": total, "g": torch.full((total,), float(rank + 1))}

    def flat_buffers(self):
        return self._flat


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from micro_diffusion_amd.arch import plan_blocks, DiTConfig
        from micro_diffusion_amd.trainer import GradSync
        from oracle import microdit_ref as orc
        fake = _FakeDiT(rank)
        sync = GradSync(fake)
        mixer, backbone = plan_blocks(DiTConfig(**orc.micro_config().__dict__))
        # the engine's segment order (engine.backward): final layer, backbone reversed, mixer reversed, rest
        order = ["final_layer"] + [b.name for b in reversed(backbone)] + [b.name for b in reversed(mixer)] + ["rest"]
        sync.active = False
        for name in order:                       # inactive (not the last microbatch): nothing may be reduced
            sync.on_segment(name)
        assert not sync.pending and torch.all(fake._flat["g"] == rank + 1)
        sync.active = True
        for name in order:
            sync.on_segment(name)
        sync.finish()
        g = fake._flat["g"]
        ok = bool(torch.all(g == sum(range(1, world + 1))))
        q.put((rank, ok, float(g.min()), float(g.max())))
    finally:
        dist.destroy_process_group()


def test_grad_buckets_cover_flat_buffer_once_gloo_ws2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, ok, lo, hi in res:
        assert ok, f"rank {rank}: every element must be reduced exactly once (min {lo}, max {hi}, expected 3)"


def test_splitk_factor_fills_whole_rounds():
    """DiTEngine._ksplit (host logic, no GPU): weight-gradient GEMMs get a split-K factor the persistent 256 x 256 kernel
    accepts (contraction per split a multiple of 128) that deals at least 192 work items (the AUTO rule's pp256 threshold) to the 256 CUs and
    fits the workspace; the two heaviest XL/2 weight-gradient shapes keep the factors measured best on MI355X
    (profiles/r2_gemm_variants_mb1024_call1.txt); ragged contractions fall back to the 128 x 128 rule (>= 512 k per split)."""
    import types
    from micro_diffusion_amd.engine import DiTEngine

    class _WS:
        def numel(self):
            return 128 << 20
    eng = types.SimpleNamespace(ws=_WS(), wgrad_target_blocks=768)
    pick = lambda *a: DiTEngine._ksplit(eng, *a)   # noqa: E731
    assert pick(1024, 1024, 65536, 1) == 16        # 16 tiles x 16 = 256 work items, one per CU
    assert pick(2048, 1024, 78848, 1) == 8         # 32 tiles x 8 = 256
    assert pick(1024, 1024, 16384, 1) == 16        # microbatch 256 (the per-rank shape of an 8-GPU run)
    assert pick(78848, 1024, 2048, 1) == -1        # 1232 tiles of its own: no split, one slice through the workspace
    assert pick(1024, 3840, 4096, 8) == -1         # the 8-expert weight gradient of a MoE block at microbatch 256: 480 tiles
    for shape in [(1024, 1024, 65536, 1), (768, 3072, 65536, 8), (16, 1024, 65536, 1), (256, 256, 1024, 1), (1024, 8, 512, 1),
                  (768, 768, 262144, 1), (5376, 1024, 65536, 1), (3072, 1024, 16384, 1), (6144, 1024, 1024, 1), (1024, 1024, 1000, 1)]:
        ks = pick(*shape)
        rows, cols, K, batch = shape
        if ks == -1:                         # enough tiles without splitting: ONE slice through the workspace on pp256
            assert ((rows + 255) // 256) * ((cols + 255) // 256) * batch >= 192 and K % 128 == 0
            ks = 1
        assert 1 <= ks <= 64 and ks * rows * cols * batch <= (128 << 20)
        t256 = ((rows + 255) // 256) * ((cols + 255) // 256) * batch
        pp256_pick = ks > 1 and K % (128 * ks) == 0 and t256 * ks >= 192     # md_gemm_bf16's AUTO rule: >= 192 tiles run on pp256
        assert pp256_pick or K // ks >= 512 or ks == 1


def _bf16_sum_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    g = torch.Generator().manual_seed(1234 + rank)
    n = 1 << 16
    # rank gradients of a realistic spread: a shared direction (the batch-mean gradient) plus per-rank noise of the same size
    torch.manual_seed(77)
    common = torch.randn(n) * torch.logspace(-4, 0, n)
    grad = common + torch.randn(n, generator=g) * common.abs()
    exact = grad.double().clone()
    dist.all_reduce(exact)                                   # fp64 sum of the fp32 rank gradients = the reference value
    f32 = grad.clone()
    dist.all_reduce(f32)                                     # fp32 exchange
    # bf16 exchange as GradSync("bf16") does it: each rank rounds its fp32 gradient to bf16, the collective sums in bf16.
    # gloo has no bf16 arithmetic of its own choosing, so model the two extremes of a ring: (a) every partial sum rounded to
    # bf16 (what a ring reduce-scatter that accumulates in the wire dtype does), (b) bf16 inputs, fp32 accumulation.
    mine = grad.to(torch.bfloat16)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    acc_bf16 = gathered[0].clone()
    for t in gathered[1:]:
        acc_bf16 = (acc_bf16 + t)                            # bf16 + bf16 -> rounded to bf16 at every hop
    acc_f32 = torch.stack(gathered).float().sum(0)
    if rank == 0:
        ex = exact
        rel = lambda a: float((a.double() - ex).norm() / ex.norm())          # noqa: E731
        nrm = lambda a: float(abs(a.double().norm() - ex.norm()) / ex.norm())  # noqa: E731
        q.put({"fp32": rel(f32), "bf16_hop_rounded": rel(acc_bf16.float()), "bf16_in_f32_acc": rel(acc_f32),
               "norm_fp32": nrm(f32), "norm_bf16_hop_rounded": nrm(acc_bf16.float()), "norm_bf16_in_f32_acc": nrm(acc_f32)})
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_exchange_error_at_8_ranks():
    """the data-parallel exchange sums bf16-rounded rank gradients (trainer.GradSync 'bf16'; FSDP's default
    mixed precision reduces in the low precision too, SURVEY.md C.7 [memory]).  Measured here at world size 8 (gloo, CPU) against
    the fp64 sum of the fp32 gradients: the element-wise error of the summed gradient is one bf16 rounding (~2^-9 relative,
    0.3 %) whether the ring rounds at every hop or accumulates in fp32, the global norm (what the clip coefficient sees) moves
    by < 0.01 %, and AdamW's update direction (sign / ratio of moments) is insensitive to a zero-mean 0.3 % perturbation.
    The numbers are asserted so a change of the exchange format shows up."""
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 8
    procs = [ctx.Process(target=_bf16_sum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
    print(res)
    assert res["fp32"] < 1e-6
    assert res["bf16_in_f32_acc"] < 3e-3 and res["bf16_hop_rounded"] < 6e-3, res
    assert res["norm_bf16_in_f32_acc"] < 2e-4 and res["norm_bf16_hop_rounded"] < 5e-4, res


def test_shard_plan_partitions_every_bucket():
    """trainer.shard_plan (host logic of the sharded optimiser step, the reference's SHARD_GRAD_OP): for world sizes 1..16 the
    rank chunks of every matrix-shaped bucket of the XL/2 layout tile the bucket exactly once, are 128-byte aligned in bf16,
    the packed per-rank offsets are contiguous, every one-dimensional tensor lies in the "small" tail region and nowhere else,
    and [w1; w2] of a SwiGLU stay adjacent (the fused FFN GEMMs rely on it)."""
    import numpy as np
    from micro_diffusion_amd import dit as mdit
    from micro_diffusion_amd.arch import bucket_key
    from micro_diffusion_amd.trainer import shard_plan
    m = mdit.MicroDiT_XL_2()
    offs, total = mdit.flat_layout(m._table)
    buckets = mdit.bucket_ranges(m._table, offs, total)
    assert buckets[0][1] == 0 and buckets[-1][2] == total and all(a[2] == b[1] for a, b in zip(buckets, buckets[1:]))
    assert [k for k, _, _ in buckets].count("small") == 1 and buckets[-1][0] == "small"
    for spec in m._table:
        if spec.buffer:
            continue
        key = bucket_key(spec.name, len(spec.shape))
        o, n = offs[spec.name], int(np.prod(spec.shape))
        assert o % 64 == 0
        home = [b for b in buckets if b[1] <= o and o + n <= b[2]]
        assert len(home) == 1 and home[0][0] == key, (spec.name, key, home)
        assert (len(spec.shape) <= 1) == (key == "small")
    # the modulation Linear of every block: one contiguous [sum 6 d_l, D] matrix at the head of the flat buffers (bucket "adaln"),
    # mixer blocks first, then backbone blocks, and their biases contiguous in the same order at the head of the "small" region --
    # the engine's batched adaLN GEMM (DiTEngine._adaln_region) reads them as one operand
    from micro_diffusion_amd.arch import plan_blocks
    mixer, backbone = plan_blocks(m.config)
    assert [b[0] for b in buckets[:2]] == ["adaln.m", "adaln.b"] and buckets[0][1] == 0 and buckets[0][2] == buckets[1][1]
    shapes = {s.name: s.shape for s in m._table}
    for suffix, first in ((".adaLN_modulation.1.weight", 0), (".adaLN_modulation.1.bias", buckets[-1][1])):
        nxt = first
        for bp in list(mixer) + list(backbone):
            nm = bp.name + suffix
            assert offs[nm] == nxt, (nm, offs[nm], nxt)
            nxt += int(np.prod(shapes[nm]))
    assert buckets[1][2] >= sum(int(np.prod(shapes[bp.name + ".adaLN_modulation.1.weight"])) for bp in list(mixer) + list(backbone))
    assert buckets[0][2] == sum(int(np.prod(shapes[bp.name + ".adaLN_modulation.1.weight"])) for bp in mixer), "no gap between the two parts"
    w1 = next(s for s in m._table if s.name == "blocks.4.mlp.w1.weight")
    assert offs["blocks.4.mlp.w2.weight"] == offs["blocks.4.mlp.w1.weight"] + int(np.prod(w1.shape))
    for world in (1, 2, 4, 8, 16):
        plan, small, own = shard_plan(buckets, world)
        assert small == (buckets[-1][1], buckets[-1][2])
        covered, nxt = 0, 0
        for key, lo, hi, chunk, olo in plan:
            assert chunk * world == hi - lo and chunk % 64 == 0 and olo == nxt
            nxt += chunk
            covered += hi - lo
        assert nxt == own and covered + (small[1] - small[0]) == total
    with pytest.raises(ValueError):
        shard_plan([("rest", 0, 1024 * 3)], 7)


def _worker_ws3(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings
        from micro_diffusion_amd.trainer import GradSync
        fake = _FakeDiT(rank)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            # what Trainer(dp_mode="auto") builds over RCCL: bf16 exchange, mode picked by GradSync
            sync = GradSync(fake, exchange="bf16", mode="auto")
        explicit = None
        try:
            GradSync(fake, exchange="bf16", mode="sharded")
        except ValueError as e:
            explicit = str(e)
        q.put((rank, sync.mode, sync.mode_note, [str(w.message) for w in rec], explicit))
    finally:
        dist.destroy_process_group()


def test_auto_mode_falls_back_to_allreduce_when_world_does_not_divide_16():
    """buckets are multiples of 1024 elements, so they split into aligned rank chunks only when the world size
    divides 16.  GradSync(mode='auto') must not crash a 3-, 5-, 6- or 7-GPU run: it keeps the all-reduce exchange and says so;
    only an explicit mode='sharded' raises."""
    from micro_diffusion_amd import dit as mdit
    from micro_diffusion_amd.trainer import shard_plan
    m = mdit.MicroDiT_XL_2()
    offs, total = mdit.flat_layout(m._table)
    buckets = mdit.bucket_ranges(m._table, offs, total)
    for world in (3, 5, 6, 7):
        with pytest.raises(ValueError):
            shard_plan(buckets, world)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ws3, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, mode, note, warns, explicit in res:
        assert mode == "allreduce" and note and "world size 3" in note, (rank, mode, note)
        assert any("falling back to the all-reduce exchange" in w for w in warns)
        assert explicit and "does not split into 3" in explicit


def test_comm_library_exports_every_declared_symbol():
    """libmicrodit_comm.so (the gradient exchange on RCCL, SURVEY.md section 8b md_comm_*): builds without a GPU, exports every
    function include/microdit_comm.h declares, and rejects malformed calls before it touches a device or RCCL."""
    from micro_diffusion_amd import comm
    path = comm.build()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "microdit_comm.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(md_comm_\w+)\s*\(", header, flags=re.M))
    assert len(declared) == 13, declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(comm.exported_symbols()), declared ^ set(comm.exported_symbols())
    L = comm.lib()
    assert L.md_comm_abi_version() == 2 == comm.ABI_VERSION
    assert L.md_comm_unique_id(None) == -1
    h = ctypes.c_void_p()
    uid = ctypes.create_string_buffer(128)
    assert L.md_comm_init(ctypes.byref(h), uid, 3, 2, 0) == -1          # rank >= world
    assert L.md_comm_init(ctypes.byref(h), None, 0, 1, 0) == -1
    assert L.md_comm_wait(None, 1, None) == -1 and L.md_comm_destroy(None) == -1
    t = ctypes.c_int64(0)
    assert L.md_comm_allreduce_bucket(None, None, 8, 0, None, ctypes.byref(t)) == -1


def test_grouped_wgrad_host_logic():
    """DiTEngine._wgrad_flush (host logic, no GPU): the weight gradients recorded for one block become ONE grouped launch whose
    problem table mirrors the layout of the gradient tensors (offsets relative to the lowest one, a foreign tensor in between left
    as a gap of the slice), with a split factor that divides the token count into multiples of 128, fits the workspace and reaches
    the persistent kernel's 192 work items when the tiles allow it; one flat reduction per contiguous run."""
    import types
    from micro_diffusion_amd import hip
    from micro_diffusion_amd.engine import DiTEngine

    class T:                                   # a tensor stand-in: address only
        def __init__(self, ptr):
            self.ptr, self.dtype = ptr, None

        def data_ptr(self):
            return self.ptr

    calls = {"gemm": [], "reduce": []}
    L = types.SimpleNamespace(md_splitk_reduce_flat=lambda ws, out, n, stride, ks, acc, st: calls["reduce"].append((ws, out, n, stride, ks, acc)) or 0)

    class WS:
        def numel(self):
            return 128 << 20

        def data_ptr(self):
            return 1 << 40
    eng = DiTEngine.__new__(DiTEngine)
    eng.ws, eng.L, eng.kernel_profile, eng.group_wgrad, eng._wgroup = WS(), L, None, True, None
    eng._st = lambda: 0
    eng._gemm = lambda **kw: calls["gemm"].append(kw)
    base = 1 << 30
    sizes = {"q.weight": (1024, 1024), "kv.weight": (2048, 1024), "xproj.weight": (1024, 1024), "w1.weight": (5632, 1024), "w3.weight": (1024, 2816)}
    eng.G, off = {}, base
    for k, (r, c) in sizes.items():            # laid out back to back in this order, like a block of the flat gradient buffer
        eng.G[k] = T(off)
        off += 4 * r * c
    tokens = 16384
    eng._wgrad_begin()
    # recorded in backward order; kv (different token count in the real block) is NOT part of the group
    eng.lin_wgrad(T(11), T(12), "w3", tokens, 1024, 2816, defer=True)
    eng.lin_wgrad(T(13), T(14), "w1", tokens, 5632, 1024, defer=True)
    eng.lin_wgrad(T(15), T(16), "xproj", tokens, 1024, 1024, defer=True)
    eng.lin_wgrad(T(17), T(18), "q", tokens, 1024, 1024, defer=True)
    assert not calls["gemm"], "deferred weight gradients must not launch before the flush"
    eng._wgrad_flush()
    assert len(calls["gemm"]) == 1
    kw = calls["gemm"][0]
    assert kw["n_problems"] == 4 and kw["K"] == tokens and kw["mode"] == hip.EPI_STORE_F32 and not kw["a_kcontig"] and not kw["b_kcontig"]
    probs = sorted(kw["_problems"], key=lambda g: g["out"])
    assert [g["out"] for g in probs] == [eng.G[k + ".weight"].data_ptr() for k in ("q", "xproj", "w1", "w3")]
    span = (eng.G["w3.weight"].data_ptr() + 4 * 1024 * 2816 - base) // 4
    assert kw["sSplit"] == span
    ks = kw["ksplit"]
    tiles = 16 + 16 + 22 * 4 + 4 * 11
    assert (tokens // 128) % ks == 0 and ks * span <= (128 << 20) and tiles * ks >= 192 and ks <= 8, ks
    # runs: [q] (kv follows it in memory but is not in the group), [xproj, w1, w3]
    assert [(out, n) for _, out, n, _, _, _ in calls["reduce"]] == [(base, 1024 * 1024), (eng.G["xproj.weight"].data_ptr(), 1024 * 1024 + 5632 * 1024 + 1024 * 2816)]
    assert all(stride == span and k_ == ks and acc == 1 for _, _, _, stride, k_, acc in calls["reduce"])
    assert calls["reduce"][1][0] - calls["reduce"][0][0] == eng.G["xproj.weight"].data_ptr() - base     # slice offsets mirror the gradient layout
    assert eng._wgroup == [] and not eng._wgrad_flush()                    # an empty group flushes to nothing


def test_rccl_channel_cap_and_in_flight_host_logic(monkeypatch):
    """trainer.cap_rccl_channels / rccl_channel_cap (NCCL_MAX_NCHANNELS: set once before the communicator exists, an exported value wins)
    and GradSync.in_flight() (what switches md_gemm_args.cu_limit on): host logic only."""
    from micro_diffusion_amd import trainer as tr
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.delenv("NCCL_MIN_NCHANNELS", raising=False)
    assert tr.rccl_channel_cap() == 0                        # unknown: the Trainer then leaves the GEMM grids alone
    assert tr.cap_rccl_channels() == tr.RCCL_CHANNELS_DEFAULT == 8
    assert os.environ["NCCL_MAX_NCHANNELS"] == "8" and os.environ["NCCL_MIN_NCHANNELS"] == "8"
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "4")
    monkeypatch.delenv("NCCL_MIN_NCHANNELS", raising=False)
    assert tr.cap_rccl_channels(16) == 4 and os.environ["NCCL_MIN_NCHANNELS"] == "4"     # the user's cap wins, MIN never exceeds it
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "not-a-number")
    assert tr.rccl_channel_cap() == 0
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "not-a-number")
    monkeypatch.delenv("NCCL_MIN_NCHANNELS", raising=False)
    assert tr.cap_rccl_channels() == 0 and "NCCL_MIN_NCHANNELS" not in os.environ      # no ValueError on a non-numeric export
    sync = tr.GradSync(_FakeDiT(0))                          # no process group: a single rank, nothing in flight
    assert not sync.enabled and not sync.in_flight()

    class Work:                                              # torch.distributed's Work / comm.Ticket as far as in_flight() cares
        def __init__(self):
            self.done = False

        def is_completed(self):
            return self.done

    a, b, c = Work(), Work(), Work()
    sync.pending.append(sync._track(a))
    sync.pending.append(sync._track(b))
    sync.gather_work["blocks.0"] = [sync._track(c)]
    assert sync.in_flight()
    a.done = True
    assert sync.in_flight() and len(sync._wire) == 2          # the finished handle was dropped from the front, b still on the wire
    b.done = c.done = True
    assert not sync.in_flight()                               # the wire is idle: the GEMM grids get their CUs back although finish() /
    assert sync.pending and sync.gather_work                  # wait_gather() have not consumed the handles yet
    sync._track(object())                                     # a handle that cannot be queried counts as busy
    assert sync.in_flight()
