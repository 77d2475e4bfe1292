"""CPU restatement of the MDS shard format (TEST INFRASTRUCTURE ONLY — never imported by the product path).

The reference reads its precomputed latents through the third-party package `streaming` (mosaicml-streaming; not
vendored under /root/reference and not installed here; the reference's setup.py:17 pins `mosaicml-streaming<=0.9.0`).  Call sites anchored here:
  * writer: micro_diffusion/datasets/prepare/sa1b/precompute.py:158-174 (`MDSWriter(columns={caption: "str",
    caption_latents: "bytes", latents_256: "bytes", latents_512: "bytes"}, compression=None, size_limit=256 MiB)`) and
    :218-227 (`writer.write({...: ndarray.tobytes()})`);
  * reader: micro_diffusion/datasets/latents_loader.py:44-69 (`StreamingDataset.__getitem__` → dict of raw column
    values, then `np.frombuffer(..., float16)`).

**Parity unpinned**: the reference holds no MDS fixtures or golden shards and `streaming` cannot be run here, so this
file restates the published format (MDS version 2: `MDSWriter.encode_joint_shard` / `MDSReader.get_sample_data` /
`decode_sample`) from its documentation; the HIP-side reader is checked against this restatement only.

Format:
  index.json = {"version": 2, "shards": [shard_info, ...]}
  shard_info = {"column_names", "column_encodings", "column_sizes" (None = variable), "compression", "format": "mds",
                "hashes", "raw_data": {"basename", "bytes", "hashes"}, "samples", "size_limit", "version": 2,
                "zip_data"}
  shard file = u32 n | u32 offsets[n+1] (absolute) | config JSON (column_* of the shard) | sample_0 | ... | sample_{n-1}
  sample     = u32 size for every variable-size column, in column order | column values in column order
  encodings used by the reference: "str" (UTF-8), "bytes" (verbatim); "int" (int64, fixed size 8) is restated to cover
  the fixed-size branch of the sample header.  Column names are sorted by the writer.
"""
import json
import os
from typing import Dict, Iterable, List, Optional

import numpy as np

_ENC_SIZE = {"str": None, "bytes": None, "int": 8}   # "int" = np.int64, the fixed-size case of the sample header


def _encode(encoding: str, value) -> bytes:
    if encoding == "str":
        return value.encode("utf-8")
    if encoding == "bytes":
        return bytes(value)
    if encoding == "int":
        return np.int64(value).tobytes()
    raise ValueError(f"encoding {encoding!r} not restated")


def _decode(encoding: str, data: bytes):
    if encoding == "str":
        return data.decode("utf-8")
    if encoding == "bytes":
        return bytes(data)
    if encoding == "int":
        return int(np.frombuffer(data, np.int64)[0])
    raise ValueError(f"encoding {encoding!r} not restated")


class RefMDSWriter:
    """`MDSWriter(out=..., columns=..., compression=None, size_limit=...)`: a new shard is started when adding the next
    sample would push the shard file beyond size_limit."""

    def __init__(self, out: str, columns: Dict[str, str], size_limit: Optional[int] = 1 << 26):
        self.out = out
        os.makedirs(out, exist_ok=True)
        self.column_names = sorted(columns)
        self.column_encodings = [columns[k] for k in self.column_names]
        self.column_sizes = [_ENC_SIZE[e] for e in self.column_encodings]
        self.size_limit = size_limit
        self.config_data = json.dumps({"column_encodings": self.column_encodings, "column_names": self.column_names,
                                       "column_sizes": self.column_sizes}, sort_keys=True).encode("utf-8")
        self.extra_bytes_per_shard = 4 + 4 + len(self.config_data)
        self.extra_bytes_per_sample = 4
        self.new_samples: List[bytes] = []
        self.new_shard_size = self.extra_bytes_per_shard
        self.shards: List[dict] = []

    def _encode_sample(self, sample: dict) -> bytes:
        sizes, data = [], []
        for key, enc, size in zip(self.column_names, self.column_encodings, self.column_sizes):
            datum = _encode(enc, sample[key])
            if size is None:
                sizes.append(len(datum))
            elif size != len(datum):
                raise KeyError(f"unexpected data size for {key}")
            data.append(datum)
        return np.array(sizes, np.uint32).tobytes() + b"".join(data)

    def _flush(self):
        if not self.new_samples and self.shards:
            return
        n = np.uint32(len(self.new_samples))
        sizes = list(map(len, self.new_samples))
        offsets = np.array([0] + sizes).cumsum().astype(np.uint32)
        offsets += len(n.tobytes()) + len(offsets.tobytes()) + len(self.config_data)
        raw = n.tobytes() + offsets.tobytes() + self.config_data + b"".join(self.new_samples)
        basename = f"shard.{len(self.shards):05d}.mds"
        with open(os.path.join(self.out, basename), "wb") as fh:
            fh.write(raw)
        self.shards.append({
            "column_encodings": self.column_encodings, "column_names": self.column_names,
            "column_sizes": self.column_sizes, "compression": None, "format": "mds", "hashes": [],
            "raw_data": {"basename": basename, "bytes": len(raw), "hashes": {}},
            "samples": int(n), "size_limit": self.size_limit, "version": 2, "zip_data": None})
        self.new_samples = []
        self.new_shard_size = self.extra_bytes_per_shard

    def write(self, sample: dict):
        data = self._encode_sample(sample)
        add = self.extra_bytes_per_sample + len(data)
        if self.size_limit and self.new_samples and self.size_limit < self.new_shard_size + add:
            self._flush()
        self.new_samples.append(data)
        self.new_shard_size += add

    def finish(self):
        if self.new_samples or not self.shards:
            self._flush()
        with open(os.path.join(self.out, "index.json"), "w") as fh:
            json.dump({"shards": self.shards, "version": 2}, fh, sort_keys=True)


class RefMDSReader:
    """`StreamingDataset(local=dir)[i]` restricted to local, uncompressed shards: dict of decoded column values."""

    def __init__(self, local: str):
        self.local = local
        with open(os.path.join(local, "index.json")) as fh:
            self.index = json.load(fh)
        self.shards = self.index["shards"]
        self.first = np.concatenate([[0], np.cumsum([s["samples"] for s in self.shards])]).astype(np.int64)

    def __len__(self):
        return int(self.first[-1])

    def _sample_data(self, shard: dict, idx: int) -> bytes:
        with open(os.path.join(self.local, shard["raw_data"]["basename"]), "rb") as fp:
            fp.seek((1 + idx) * 4)
            begin, end = np.frombuffer(fp.read(8), np.uint32)
            fp.seek(int(begin))
            return fp.read(int(end) - int(begin))

    def __getitem__(self, index: int) -> dict:
        if not 0 <= index < len(self):
            raise IndexError(index)
        si = int(np.searchsorted(self.first, index, side="right") - 1)
        shard = self.shards[si]
        data = self._sample_data(shard, index - int(self.first[si]))
        sizes, idx = [], 0
        for size in shard["column_sizes"]:
            if size:
                sizes.append(size)
            else:
                (size,) = np.frombuffer(data[idx:idx + 4], np.uint32)
                sizes.append(int(size))
                idx += 4
        sample = {}
        for key, enc, size in zip(shard["column_names"], shard["column_encodings"], sizes):
            sample[key] = _decode(enc, data[idx:idx + size])
            idx += size
        return sample


def latents_getitem(sample: dict, image_size: int, cap_seq_size: int, cap_emb_dim: int) -> Dict[str, np.ndarray]:
    """The decode half of StreamingLatentsDataset.__getitem__ (latents_loader.py:52-68), without the caption-drop coin."""
    out = {"caption_latents": np.frombuffer(sample["caption_latents"], np.float16).copy().reshape(1, cap_seq_size, cap_emb_dim)}
    if image_size == 256 and "latents_256" in sample:
        out["image_latents"] = np.frombuffer(sample["latents_256"], np.float16).copy().reshape(-1, 32, 32)
    if image_size == 512 and "latents_512" in sample:
        out["image_latents"] = np.frombuffer(sample["latents_512"], np.float16).copy().reshape(-1, 64, 64)
    return out


def write_synthetic_latents(out: str, n: int, seed: int = 0, cap_seq_size: int = 77, cap_emb_dim: int = 1024,
                            size_limit: Optional[int] = 1 << 22, with_512: bool = True) -> List[dict]:
    """Write n synthetic samples with the column set of precompute.py:158-163; returns the samples written."""
    rng = np.random.default_rng(seed)
    cols = {"caption": "str", "caption_latents": "bytes", "latents_256": "bytes"}
    if with_512:
        cols["latents_512"] = "bytes"
    w = RefMDSWriter(out, cols, size_limit=size_limit)
    samples = []
    for i in range(n):
        s = {"caption": f"sample {i} é中 " + "x" * int(rng.integers(0, 40)),
             "caption_latents": rng.standard_normal((cap_seq_size, cap_emb_dim)).astype(np.float16).tobytes(),
             "latents_256": (rng.standard_normal((4, 32, 32)) * 0.8).astype(np.float16).tobytes()}
        if with_512:
            s["latents_512"] = (rng.standard_normal((4, 64, 64)) * 0.8).astype(np.float16).tobytes()
        w.write(s)
        samples.append(s)
    w.finish()
    return samples
