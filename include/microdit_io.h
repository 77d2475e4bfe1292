/* C ABI of libmicrodit_io.so: host-side reader for the precomputed-latents shards that feed the MicroDiT training path.
 *
 * Replaces, for this path, what the reference gets from the third-party `streaming` package (mosaicml-streaming, not
 * vendored in the reference; imported at micro_diffusion/datasets/latents_loader.py:3): `StreamingDataset.__getitem__`
 * (latents_loader.py:44) returning the raw `bytes` columns `caption_latents`, `latents_256`, `latents_512` that
 * micro_diffusion/datasets/prepare/<dataset>/precompute.py:158-174,218-227 wrote with
 * `MDSWriter(columns={...: "bytes"}, compression=None, size_limit=256 MiB)`.
 *
 * MDS shard layout (published format, version 2; restated in oracle/mds_ref.py):
 *   <dir>/index.json : {"version": 2, "shards": [{"format": "mds", "column_names": [...], "column_encodings": [...],
 *                        "column_sizes": [null | int, ...], "compression": null, "samples": S,
 *                        "raw_data": {"basename": "shard.00000.mds", "bytes": N, ...}, ...}, ...]}
 *   shard file       : u32 num_samples | u32 offsets[num_samples + 1] (absolute file offsets) | column-config JSON |
 *                      samples; sample = u32 size per variable-size column (column order) followed by the column data.
 * Only uncompressed shards are supported (the only kind the reference writes).
 *
 * Plain pointers and sizes only; every function returns MD_IO_OK or a negative error code and leaves a message that
 * md_mds_last_error() returns.  The library has no GPU dependency: destination buffers are caller-owned host memory
 * (pinned by the caller for asynchronous H2D copies).
 */
#ifndef MICRODIT_IO_H
#define MICRODIT_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MD_IO_OK 0
#define MD_IO_BAD_ARG (-1)        /* null pointer, sample / column index out of range */
#define MD_IO_NOT_FOUND (-2)      /* index.json or a shard file is missing */
#define MD_IO_BAD_FORMAT (-3)     /* malformed index.json / shard header, truncated shard, inconsistent columns */
#define MD_IO_UNSUPPORTED (-4)    /* compressed shards, formats other than "mds" */
#define MD_IO_SIZE_MISMATCH (-5)  /* a sample's column does not have the requested row size */

typedef struct md_mds md_mds;     /* one local MDS directory (= one `streaming.Stream(local=dir)`) */

int32_t md_io_abi_version(void);

/* Parse <dir>/index.json and validate it; shard files are memory-mapped on first use. */
int md_mds_open(const char* dir, md_mds** out);
void md_mds_close(md_mds* h);
const char* md_mds_last_error(const md_mds* h);   /* h may be NULL: message of the last failed md_mds_open */

int64_t md_mds_num_samples(const md_mds* h);
int32_t md_mds_num_shards(const md_mds* h);
int32_t md_mds_num_columns(const md_mds* h);
/* Column name / encoding by position (pointers stay valid until md_mds_close); NULL when out of range. */
const char* md_mds_column_name(const md_mds* h, int32_t column);
const char* md_mds_column_encoding(const md_mds* h, int32_t column);
int32_t md_mds_column_index(const md_mds* h, const char* name);   /* -1 when absent */

/* Byte size of one sample's column. */
int md_mds_sample_size(md_mds* h, int64_t sample, int32_t column, int64_t* nbytes);

/* Copy one variable-size column value (e.g. the `caption` string) into dst (capacity cap); *nbytes = its size. */
int md_mds_read_sample(md_mds* h, int64_t sample, int32_t column, void* dst, int64_t cap, int64_t* nbytes);

/* Gather a batch: dst[i * row_stride .. + row_bytes) = column bytes of samples[i], i < n.  Every value must be exactly
 * row_bytes long (fp16 latents have a fixed shape), else MD_IO_SIZE_MISMATCH and dst is unspecified.  The copies are
 * spread over n_threads host threads (<= 1: the calling thread). */
int md_mds_read_batch(md_mds* h, const int64_t* samples, int32_t n, int32_t column, void* dst, int64_t row_bytes,
                      int64_t row_stride, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif /* MICRODIT_IO_H */
