"""Latents data path: the native MDS reader (libmicrodit_io.so, through its C ABI) against the CPU
restatement of the format in oracle/mds_ref.py, and the dataset / loader that stand where the reference's
StreamingLatentsDataset / DataLoader stand (micro_diffusion/datasets/latents_loader.py:8-108).  All CPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from micro_diffusion_amd import data as mdata
from micro_diffusion_amd import mds
from oracle import mds_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shards(tmp_path_factory):
    d = tmp_path_factory.mktemp("mds_a")
    # 4 MiB limit: ~190 KB per sample -> ~21 samples per shard -> several shards, last one ragged
    samples = mds_ref.write_synthetic_latents(str(d), 50, seed=3, size_limit=1 << 22)
    return str(d), samples


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "microdit_io.h")).read()
    declared = sorted(set(re.findall(r"\b(md_(?:io|mds)_\w+)\s*\(", hdr)))
    assert declared == mds.exported_symbols()
    lib = ctypes.CDLL(mds.build())
    for sym in declared:
        getattr(lib, sym)
    assert mds.lib().md_io_abi_version() == 1


def test_oracle_round_trip_and_layout(shards):
    d, samples = shards
    r = mds_ref.RefMDSReader(d)
    assert len(r) == len(samples) and len(r.shards) >= 3
    assert r.shards[0]["column_names"] == sorted(samples[0])          # the writer sorts the column names
    for i in (0, 1, 20, 21, 22, 49):
        assert r[i] == samples[i]
    # header of a shard: count, absolute offsets, last offset = file size
    raw = open(os.path.join(d, r.shards[0]["raw_data"]["basename"]), "rb").read()
    n = int(np.frombuffer(raw[:4], np.uint32)[0])
    off = np.frombuffer(raw[4:4 + 4 * (n + 1)], np.uint32)
    assert n == r.shards[0]["samples"] and off[-1] == len(raw) == r.shards[0]["raw_data"]["bytes"]
    assert off[0] > 4 + 4 * (n + 1)                                    # the column-config JSON sits between table and data


def test_native_reader_matches_oracle_every_value(shards):
    d, samples = shards
    r = mds_ref.RefMDSReader(d)
    m = mds.MDSDir(d)
    assert len(m) == len(r) == 50 and m.num_shards == len(r.shards)
    assert m.column_names == r.shards[0]["column_names"] and m.column_encodings == r.shards[0]["column_encodings"]
    for i in range(len(r)):
        ref = r[i]
        for c, (name, enc) in enumerate(zip(m.column_names, m.column_encodings)):
            got = m.read_value(i, c)
            want = ref[name].encode("utf-8") if enc == "str" else ref[name]
            assert got == want, (i, name)
            assert m.sample_size(i, c) == len(want)


@pytest.mark.parametrize("threads", [1, 4, 64])
def test_read_batch_gather(shards, threads):
    d, samples = shards
    m = mds.MDSDir(d)
    rng = np.random.default_rng(0)
    idx = rng.permutation(50)[:37].astype(np.int64)
    idx[5] = idx[6]                                                    # duplicates are allowed
    col = m.column("latents_256")
    row = 4 * 32 * 32 * 2
    stride = row + 64                                                  # strided destination
    buf = np.full((len(idx), stride), 0xAB, np.uint8)
    m.read_batch(idx, col, buf.ctypes.data, row, stride, threads)
    for j, i in enumerate(idx):
        assert buf[j, :row].tobytes() == samples[i]["latents_256"]
        assert (buf[j, row:] == 0xAB).all()
    m.read_batch(idx[:0], col, buf.ctypes.data, row, stride, threads)  # empty batch is a no-op


def test_fixed_size_columns_and_empty_dataset(tmp_path):
    w = mds_ref.RefMDSWriter(str(tmp_path / "fix"), {"id": "int", "blob": "bytes", "zz": "int"}, size_limit=300)
    vals = [{"id": i * 7 - 3, "blob": bytes([i]) * (i % 5), "zz": -i} for i in range(23)]
    for v in vals:
        w.write(v)
    w.finish()
    m = mds.MDSDir(str(tmp_path / "fix"))
    r = mds_ref.RefMDSReader(str(tmp_path / "fix"))
    assert m.num_shards > 3 and len(m) == 23
    for i, v in enumerate(vals):
        assert r[i] == v
        assert np.frombuffer(m.read_value(i, m.column("id")), np.int64)[0] == v["id"]
        assert np.frombuffer(m.read_value(i, m.column("zz")), np.int64)[0] == v["zz"]
        assert m.read_value(i, m.column("blob")) == v["blob"]           # zero-length values included
    e = mds_ref.RefMDSWriter(str(tmp_path / "empty"), {"caption_latents": "bytes", "latents_256": "bytes"})
    e.finish()
    me = mds.MDSDir(str(tmp_path / "empty"))
    assert len(me) == 0 and me.num_shards == 1
    with pytest.raises(mds.MDSError) as ei:
        me.sample_size(0, 0)
    assert ei.value.code == mds.BAD_ARG


def test_error_behaviour(shards, tmp_path):
    d, samples = shards
    with pytest.raises(mds.MDSError) as ei:
        mds.MDSDir(str(tmp_path / "nowhere"))
    assert ei.value.code == mds.NOT_FOUND
    m = mds.MDSDir(d)
    for bad in (-1, 50):
        with pytest.raises(mds.MDSError) as ei:
            m.sample_size(bad, 0)
        assert ei.value.code == mds.BAD_ARG
    with pytest.raises(mds.MDSError) as ei:
        m.sample_size(0, 99)
    assert ei.value.code == mds.BAD_ARG
    with pytest.raises(KeyError):
        m.column("latents_1024")
    buf = np.zeros((2, 100), np.uint8)
    with pytest.raises(mds.MDSError) as ei:                            # wrong fixed row size
        m.read_batch(np.array([0, 1]), m.column("latents_256"), buf.ctypes.data, 100, 100, 2)
    assert ei.value.code == mds.SIZE_MISMATCH

    def variant(name, edit_index=None, edit_shard=None):
        v = tmp_path / name
        v.mkdir()
        idx = json.load(open(os.path.join(d, "index.json")))
        idx["shards"] = idx["shards"][:1]
        raw = open(os.path.join(d, idx["shards"][0]["raw_data"]["basename"]), "rb").read()
        if edit_shard:
            raw = edit_shard(raw)
        if edit_index:
            edit_index(idx)
        open(v / idx["shards"][0]["raw_data"]["basename"], "wb").write(raw)
        text = idx if isinstance(idx, str) else json.dumps(idx)
        open(v / "index.json", "w").write(text)
        return str(v)

    def expect(path, code, touch=True):
        with pytest.raises(mds.MDSError) as ei:
            mm = mds.MDSDir(path)
            if touch:
                mm.sample_size(0, 0)
        assert ei.value.code == code, str(ei.value)

    expect(variant("zstd", lambda i: i["shards"][0].__setitem__("compression", "zstd")), mds.UNSUPPORTED)
    expect(variant("fmt", lambda i: i["shards"][0].__setitem__("format", "json")), mds.UNSUPPORTED)
    expect(variant("nocols", lambda i: i["shards"][0].pop("column_names")), mds.BAD_FORMAT)
    expect(variant("trunc", edit_shard=lambda r: r[:len(r) // 2]), mds.BAD_FORMAT)
    expect(variant("count", lambda i: i["shards"][0].__setitem__("samples", 3)), mds.BAD_FORMAT)

    def scramble(r):
        b = bytearray(r)
        b[8:12] = (2 ** 31).to_bytes(4, "little")                      # offset 1 beyond the file
        return bytes(b)
    expect(variant("offsets", edit_shard=scramble), mds.BAD_FORMAT)
    bad = tmp_path / "badjson"
    bad.mkdir()
    open(bad / "index.json", "w").write('{"shards": [ {"format": "mds", ')
    expect(str(bad), mds.BAD_FORMAT, touch=False)
    gone = variant("gone")
    os.remove(os.path.join(gone, "shard.00000.mds"))
    expect(gone, mds.NOT_FOUND)


@pytest.mark.parametrize("image_size", [256, 512])
def test_dataset_getitem_matches_reference_decode(shards, tmp_path, image_size):
    d, samples = shards
    d2 = str(tmp_path / "b")
    samples2 = mds_ref.write_synthetic_latents(d2, 9, seed=11, size_limit=1 << 21)
    ds = mdata.StreamingLatentsDataset(streams=[d, d2], shuffle=False, image_size=image_size, cap_seq_size=77,
                                       cap_emb_dim=1024, cap_drop_prob=0.0, batch_size=4)
    assert len(ds) == 59 and ds.in_channels == 4
    allsamples = samples + samples2
    for i in (0, 17, 49, 50, 58):
        got = ds[i]
        want = mds_ref.latents_getitem(allsamples[i], image_size, 77, 1024)
        assert got["drop_caption_mask"] == 1.
        assert got["caption_latents"].dtype == torch.float16 and tuple(got["caption_latents"].shape) == (1, 77, 1024)
        assert tuple(got["image_latents"].shape) == (4, image_size // 8, image_size // 8)
        assert np.array_equal(got["caption_latents"].numpy().view(np.uint16), want["caption_latents"].view(np.uint16))
        assert np.array_equal(got["image_latents"].numpy().view(np.uint16), want["image_latents"].view(np.uint16))
    with pytest.raises(IndexError):
        ds[59]
    # batch gather across the two streams, unordered
    idx = np.array([58, 3, 50, 49, 4, 5, 51], np.int64)
    cap = torch.empty(7, 1, 77, 1024, dtype=torch.float16)
    lat = torch.empty(7, 4, image_size // 8, image_size // 8, dtype=torch.float16)
    ds.read_batch(idx, cap, lat, n_threads=3)
    for j, i in enumerate(idx):
        want = mds_ref.latents_getitem(allsamples[i], image_size, 77, 1024)
        assert np.array_equal(cap[j].numpy().view(np.uint16), want["caption_latents"].view(np.uint16))
        assert np.array_equal(lat[j].numpy().view(np.uint16), want["image_latents"].view(np.uint16))
    # the caption-drop coin of __getitem__ (latents_loader.py:49-51)
    ds.cap_drop_prob = 1.0
    assert ds[0]["drop_caption_mask"] == 0.


def test_loader_epoch_partition_and_determinism(shards):
    d, samples = shards
    ds = mdata.StreamingLatentsDataset(streams=[d], shuffle=True, image_size=256, cap_seq_size=77, cap_emb_dim=1024,
                                       cap_drop_prob=0.3, batch_size=6)
    loaders = [mdata.LatentsLoader(ds, 6, drop_last=True, device="cpu", rank=r, world_size=2, seed=5, depth=2) for r in (0, 1)]
    assert [len(l) for l in loaders] == [4, 4]                          # 25 samples per rank -> 4 full batches of 6
    ids = [l.epoch_indices(0) for l in loaders]
    assert len(np.intersect1d(ids[0], ids[1])) == 0 and len(np.unique(np.concatenate(ids))) == 50
    assert not np.array_equal(loaders[0].epoch_indices(0), loaders[0].epoch_indices(1))
    by_bytes = {s["latents_256"]: i for i, s in enumerate(samples)}
    seen = []
    first_epoch = []
    for r, l in enumerate(loaders):
        batches = list(l)
        assert len(batches) == 4 and l.epoch == 1
        for b, batch in enumerate(batches):
            assert batch["image_latents"].shape == (6, 4, 32, 32) and batch["image_latents"].dtype == torch.float16
            assert batch["caption_latents"].shape == (6, 1, 77, 1024) and batch["drop_caption_mask"].shape == (6,)
            assert set(batch["drop_caption_mask"].tolist()) <= {0.0, 1.0}
            for j in range(6):
                i = by_bytes[batch["image_latents"][j].numpy().tobytes()]
                assert i == ids[r][b * 6 + j]
                assert batch["caption_latents"][j].numpy().tobytes() == samples[i]["caption_latents"]
                seen.append(i)
        first_epoch.append(batches)
    assert len(set(seen)) == 48
    # same seed, same epoch -> identical batches and coins; the loader moved on to epoch 1 -> different order
    again = mdata.LatentsLoader(ds, 6, drop_last=True, device="cpu", rank=0, world_size=2, seed=5, depth=3)
    for a, b in zip(list(again), first_epoch[0]):
        assert all(torch.equal(a[k], b[k]) for k in a)
    nxt = list(loaders[0])
    assert not torch.equal(nxt[0]["image_latents"], first_epoch[0][0]["image_latents"])
    # drop_last=False keeps the ragged tail; loop=True runs across epochs
    tail = mdata.LatentsLoader(ds, 6, drop_last=False, device="cpu", rank=0, world_size=2, seed=5)
    sizes = [b["image_latents"].shape[0] for b in tail]
    assert sizes == [6, 6, 6, 6, 1]
    endless = mdata.LatentsLoader(ds, 6, device="cpu", rank=0, world_size=1, seed=5, loop=True)
    it = iter(endless)
    got = [next(it)["image_latents"].shape[0] for _ in range(20)]       # 8 batches per epoch -> crosses two epoch ends
    assert got == [6] * 20
    it.close()


def test_loader_coin_rate_and_reader_errors_surface(shards, tmp_path):
    d, _ = shards
    ds = mdata.StreamingLatentsDataset(streams=[d], shuffle=True, image_size=256, cap_seq_size=77, cap_emb_dim=1024,
                                       cap_drop_prob=0.1, batch_size=10)
    l = mdata.LatentsLoader(ds, 10, device="cpu", rank=0, world_size=1)
    coins = torch.cat([l._drop_coins(e, b, 10) for e in range(40) for b in range(5)])
    assert abs((1 - coins.mean().item()) - 0.1) < 0.03                  # P(drop) = cap_drop_prob, yaml: 0.1
    # a caption column with the wrong context length is reported, not silently reshaped
    bad = mdata.StreamingLatentsDataset(streams=[d], shuffle=False, image_size=256, cap_seq_size=64, cap_emb_dim=1024,
                                        batch_size=4)
    with pytest.raises(mds.MDSError) as ei:
        list(mdata.LatentsLoader(bad, 4, device="cpu", rank=0, world_size=1))
    assert ei.value.code == mds.SIZE_MISMATCH
    # factory: MDS dirs present -> reader; a missing path raises (never silently random data); random latents only
    # on explicit request (datadir="synthetic"; the generator needs a device, so that branch is exercised on the GPU box)
    ld = mdata.build_streaming_latents_dataloader([d], batch_size=5, image_size=256, cap_drop_prob=0.1, shuffle=True,
                                                  drop_last=True, num_workers=2, prefetch_factor=2, device="cpu")
    assert isinstance(ld, mdata.LatentsLoader) and len(ld) == 10 and ld.dataset.cap_drop_prob == 0.1
    with pytest.raises(FileNotFoundError):
        mdata.build_streaming_latents_dataloader([d, str(tmp_path / "missing")], batch_size=5)
    with pytest.raises(FileNotFoundError):
        mdata.build_streaming_latents_dataloader(str(tmp_path / "missing"), batch_size=5)
    with pytest.raises(ValueError):            # fewer samples than one batch: would spin forever with loop=True
        mdata.LatentsLoader(mdata.StreamingLatentsDataset(streams=[d], shuffle=False, image_size=256, cap_seq_size=77, cap_emb_dim=1024,
                                                          batch_size=64), 64,
                            device="cpu", rank=0, world_size=1)


def test_loader_resume_mid_epoch(shards):
    """state_dict / load_state_dict: a loader restored from (epoch, batches consumed) continues with exactly the batches
    (and caption-drop coins) the original would have produced, across the epoch boundary."""
    d, _ = shards
    ds = mdata.StreamingLatentsDataset(streams=[d], shuffle=True, image_size=256, cap_seq_size=77, cap_emb_dim=1024,
                                       cap_drop_prob=0.2, batch_size=8)
    a = mdata.LatentsLoader(ds, 8, device="cpu", rank=0, world_size=1, seed=9, loop=True)      # 6 batches per epoch
    it = iter(a)
    for _ in range(4):
        next(it)
    state = a.state_dict()
    assert (state["epoch"], state["batch_in_epoch"]) == (0, 4)
    rest = [next(it) for _ in range(5)]                                 # batches 4, 5 of epoch 0 and 0, 1, 2 of epoch 1
    assert (a.state_dict()["epoch"], a.state_dict()["batch_in_epoch"]) == (1, 3)
    it.close()
    b = mdata.LatentsLoader(ds, 8, device="cpu", rank=0, world_size=1, seed=9, loop=True)
    b.load_state_dict(state)
    itb = iter(b)
    for want in rest:
        got = next(itb)
        assert all(torch.equal(got[k], want[k]) for k in want)
    itb.close()
    assert b.state_dict()["epoch"] == 1 and b.state_dict()["batch_in_epoch"] == 3
    with pytest.raises(ValueError):
        mdata.LatentsLoader(ds, 4, device="cpu", rank=0, world_size=1, seed=9).load_state_dict(state)
    # a state saved exactly at an epoch end resumes at the start of the next epoch
    c = mdata.LatentsLoader(ds, 8, device="cpu", rank=0, world_size=1, seed=9)
    c.load_state_dict({"epoch": 2, "batch_in_epoch": 6})
    assert (c.epoch, c.batch_in_epoch) == (3, 0)


def test_hand_assembled_shard_byte_for_byte(tmp_path):
    """An independent pin of the shard layout (§8f-1 was checked against one restatement only).  The shard below is
    assembled BYTE BY BYTE here with struct.pack from the published MDS v2 layout -- it shares no code with oracle/mds_ref.py's
    writer -- and both readers (the native one through the C ABI and the oracle's) must return exactly the values that went in:
      u32 n | u32 absolute offsets[n + 1] | column-config JSON | per sample: u32 size of every VARIABLE-size column (column order),
      then the column values in column order.  Columns are stored sorted by name; "int" is the fixed-size (8-byte) case, which
      has no entry in the per-sample size header."""
    import struct
    names = ["caption", "caption_latents", "idx", "latents_256"]                 # sorted, as MDSWriter stores them
    encs = ["str", "bytes", "int", "bytes"]
    sizes = [None, None, 8, None]
    rng = np.random.default_rng(7)
    rows = []
    for i, cap in enumerate(["a photo of a cat", "", "zwölf Boxkämpfer jagen Viktor"]):     # empty and multi-byte UTF-8 captions
        rows.append({"caption": cap, "caption_latents": rng.standard_normal((1, 77, 8)).astype(np.float16).tobytes(),
                     "idx": 1000 + i, "latents_256": rng.standard_normal((4, 32, 32)).astype(np.float16).tobytes()})
    cfg = json.dumps({"column_encodings": encs, "column_names": names, "column_sizes": sizes}, sort_keys=True).encode()
    blobs = []
    for r in rows:
        vals = [r["caption"].encode("utf-8"), r["caption_latents"], struct.pack("<q", r["idx"]), r["latents_256"]]
        head = b"".join(struct.pack("<I", len(v)) for v, s in zip(vals, sizes) if s is None)      # 3 variable columns -> 12 bytes
        assert len(head) == 12
        blobs.append(head + b"".join(vals))
    n = len(blobs)
    first = 4 + 4 * (n + 1) + len(cfg)
    offs, cur = [], first
    for b in blobs:
        offs.append(cur)
        cur += len(b)
    offs.append(cur)
    shard = struct.pack("<I", n) + struct.pack(f"<{n + 1}I", *offs) + cfg + b"".join(blobs)
    assert len(shard) == offs[-1]
    d = tmp_path / "hand"
    d.mkdir()
    (d / "shard.00000.mds").write_bytes(shard)
    (d / "index.json").write_text(json.dumps({"version": 2, "shards": [{
        "column_encodings": encs, "column_names": names, "column_sizes": sizes, "compression": None, "format": "mds", "hashes": [],
        "raw_data": {"basename": "shard.00000.mds", "bytes": len(shard), "hashes": {}}, "samples": n, "size_limit": 1 << 26,
        "version": 2, "zip_data": None}]}))
    m = mds.MDSDir(str(d))
    r = mds_ref.RefMDSReader(str(d))
    assert len(m) == len(r) == n and m.column_names == names and m.column_encodings == encs
    for i, row in enumerate(rows):
        assert r[i] == row
        assert m.read_value(i, m.column("caption")).decode("utf-8") == row["caption"]
        assert m.read_value(i, m.column("caption_latents")) == row["caption_latents"]
        assert struct.unpack("<q", m.read_value(i, m.column("idx")))[0] == row["idx"]
        assert m.read_value(i, m.column("latents_256")) == row["latents_256"]
    # the batched gather the loader uses: rows 2, 0, 2 of the 8 KiB latents column straight into a strided destination
    dst = np.zeros((3, 4 * 32 * 32 + 5), dtype=np.float16)
    m.read_batch(np.array([2, 0, 2], dtype=np.int64), m.column("latents_256"), dst.ctypes.data, 4 * 32 * 32 * 2, dst.strides[0], 2)
    for k, i in enumerate([2, 0, 2]):
        assert dst[k, :4096].tobytes() == rows[i]["latents_256"] and not dst[k, 4096:].any()
