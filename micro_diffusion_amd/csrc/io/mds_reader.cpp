// libmicrodit_io.so: memory-mapped reader of uncompressed MDS shards (C ABI in include/microdit_io.h).
//
// The training step consumes ~166 KB of fp16 latents per image (77 x 1024 caption + 4 x 32 x 32 image latents); at the
// step rates of one MI355X that is 0.3-0.8 GB/s per GPU, i.e. a memcpy problem.  The reader therefore keeps every shard
// mapped (page cache = the only cache), resolves a sample with two u32 reads from the shard's offset table and gathers a
// whole batch column straight into the caller's (pinned) staging buffer with a few host threads; there is no per-sample
// Python object, no intermediate `bytes`, and no copy besides the one into the staging buffer.
#include "microdit_io.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <memory>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <utility>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Minimal JSON (index.json only): objects, arrays, strings with escapes, numbers, true/false/null.
// ---------------------------------------------------------------------------------------------------------------------
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;
    const JValue* get(const char* key) const {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char* p;
    const char* end;
    std::string err;
    int depth = 0;

    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool fail(const char* m) {
        if (err.empty()) err = m;
        return false;
    }
    static void utf8(std::string& s, unsigned cp) {
        if (cp < 0x80) s += (char)cp;
        else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
        else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 0x3F)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
    }
    bool hex4(unsigned& v) {
        if (end - p < 4) return fail("truncated \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool string(std::string& s) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("truncated escape");
                const char c = *p++;
                switch (c) {
                    case '"': s += '"'; break;
                    case '\\': s += '\\'; break;
                    case '/': s += '/'; break;
                    case 'b': s += '\b'; break;
                    case 'f': s += '\f'; break;
                    case 'n': s += '\n'; break;
                    case 'r': s += '\r'; break;
                    case 't': s += '\t'; break;
                    case 'u': {
                        unsigned cp;
                        if (!hex4(cp)) return false;
                        if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                            p += 2;
                            unsigned lo;
                            if (!hex4(lo)) return false;
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        utf8(s, cp);
                        break;
                    }
                    default: return fail("bad escape");
                }
            } else {
                s += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    bool value(JValue& v) {
        if (++depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end of input");
        bool ok = true;
        const char c = *p;
        if (c == '{') {
            v.kind = JValue::Obj;
            ++p;
            ws();
            if (p < end && *p == '}') { ++p; }
            else
                for (;;) {
                    ws();
                    std::string k;
                    if (!string(k)) { ok = false; break; }
                    ws();
                    if (p >= end || *p != ':') { ok = fail("expected ':'"); break; }
                    ++p;
                    JValue child;
                    if (!value(child)) { ok = false; break; }
                    v.obj.emplace_back(std::move(k), std::move(child));
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == '}') { ++p; break; }
                    ok = fail("expected ',' or '}'");
                    break;
                }
        } else if (c == '[') {
            v.kind = JValue::Arr;
            ++p;
            ws();
            if (p < end && *p == ']') { ++p; }
            else
                for (;;) {
                    JValue child;
                    if (!value(child)) { ok = false; break; }
                    v.arr.push_back(std::move(child));
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == ']') { ++p; break; }
                    ok = fail("expected ',' or ']'");
                    break;
                }
        } else if (c == '"') {
            v.kind = JValue::Str;
            ok = string(v.str);
        } else if (end - p >= 4 && !strncmp(p, "true", 4)) { v.kind = JValue::Bool; v.b = true; p += 4; }
        else if (end - p >= 5 && !strncmp(p, "false", 5)) { v.kind = JValue::Bool; v.b = false; p += 5; }
        else if (end - p >= 4 && !strncmp(p, "null", 4)) { v.kind = JValue::Null; p += 4; }
        else if (c == '-' || (c >= '0' && c <= '9')) {
            const char* q = p;
            while (q < end && (*q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E' || (*q >= '0' && *q <= '9'))) ++q;
            std::string t(p, q);
            char* stop = nullptr;
            v.kind = JValue::Num;
            v.num = strtod(t.c_str(), &stop);
            if (!stop || *stop) ok = fail("bad number");
            p = q;
        } else {
            ok = fail("unexpected character");
        }
        --depth;
        return ok;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
struct Shard {
    std::string path;
    int64_t samples = 0;
    int64_t bytes = 0;               // raw_data.bytes (0 = not recorded)
    std::vector<int64_t> col_sizes;  // -1 = variable (size prefix in the sample header)
    const uint8_t* map = nullptr;    // published with release semantics once validated
    size_t map_len = 0;
};

thread_local std::string g_open_error;

}  // namespace

struct md_mds {
    std::string dir;
    std::vector<Shard> shards;
    std::vector<int64_t> first;      // first[i] = global index of shard i's sample 0; first[n] = total
    std::vector<std::string> col_names, col_encodings;
    std::mutex mu;                   // guards lazy mapping and `err`
    std::string err;
};

namespace {

int set_err(md_mds* h, int code, const std::string& msg) {
    if (h) {
        std::lock_guard<std::mutex> g(h->mu);
        h->err = msg;
    } else {
        g_open_error = msg;
    }
    return code;
}

inline uint32_t rd_u32(const uint8_t* p) {
    uint32_t v;
    memcpy(&v, p, 4);   // shards are little-endian; so is every host this library targets
    return v;
}

// Map + validate a shard on first use.
int map_shard(md_mds* h, Shard& s) {
    if (__atomic_load_n(&s.map, __ATOMIC_ACQUIRE)) return MD_IO_OK;
    std::unique_lock<std::mutex> g(h->mu);
    if (s.map) return MD_IO_OK;
    const int fd = open(s.path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        h->err = "cannot open shard " + s.path + ": " + strerror(errno);
        return MD_IO_NOT_FOUND;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) {
        close(fd);
        h->err = "shard " + s.path + " is truncated";
        return MD_IO_BAD_FORMAT;
    }
    const size_t len = (size_t)st.st_size;
    if (s.bytes > 0 && (int64_t)len != s.bytes) {
        close(fd);
        h->err = "shard " + s.path + ": size " + std::to_string(len) + " != index raw_data.bytes " + std::to_string(s.bytes);
        return MD_IO_BAD_FORMAT;
    }
    void* m = mmap(nullptr, len, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        h->err = "mmap failed on " + s.path + ": " + strerror(errno);
        return MD_IO_NOT_FOUND;
    }
    const uint8_t* p = static_cast<const uint8_t*>(m);
    const int64_t n = rd_u32(p);
    bool ok = n == s.samples && (size_t)(4 + 4 * (n + 1)) <= len;
    if (ok) {
        uint64_t prev = rd_u32(p + 4);
        ok = prev >= (uint64_t)(4 + 4 * (n + 1));
        for (int64_t i = 1; ok && i <= n; ++i) {
            const uint64_t o = rd_u32(p + 4 + 4 * i);
            ok = o >= prev;
            prev = o;
        }
        ok = ok && prev == len;
    }
    if (!ok) {
        munmap(m, len);
        h->err = "shard " + s.path + ": bad header (sample count / offset table inconsistent with the file)";
        return MD_IO_BAD_FORMAT;
    }
    madvise(m, len, MADV_RANDOM);     // shuffled access: no read-ahead beyond the touched pages
    s.map_len = len;
    __atomic_store_n(&s.map, p, __ATOMIC_RELEASE);
    return MD_IO_OK;
}

// Resolve (sample, column) to a pointer + size inside the mapped shard.
int locate(md_mds* h, int64_t sample, int32_t column, const uint8_t** ptr, int64_t* nbytes) {
    if (!h || sample < 0 || sample >= h->first.back() || column < 0 || column >= (int32_t)h->col_names.size())
        return set_err(h, MD_IO_BAD_ARG, "sample or column index out of range");
    const size_t si = std::upper_bound(h->first.begin(), h->first.end(), sample) - h->first.begin() - 1;
    Shard& s = h->shards[si];
    const int rc = map_shard(h, s);
    if (rc != MD_IO_OK) return rc;
    const int64_t local = sample - h->first[si];
    const uint64_t beg = rd_u32(s.map + 4 + 4 * local), end = rd_u32(s.map + 4 + 4 * (local + 1));
    const uint8_t* p = s.map + beg;
    const uint8_t* const pe = s.map + end;
    const size_t nc = s.col_sizes.size();
    size_t nvar = 0;
    for (size_t c = 0; c < nc; ++c) nvar += s.col_sizes[c] < 0;
    if ((uint64_t)(pe - p) < 4 * nvar) return set_err(h, MD_IO_BAD_FORMAT, "sample header exceeds the sample in " + s.path);
    const uint8_t* body = p + 4 * nvar;
    size_t v = 0;
    for (size_t c = 0; c < nc; ++c) {
        const int64_t sz = s.col_sizes[c] < 0 ? (int64_t)rd_u32(p + 4 * v++) : s.col_sizes[c];
        if (sz > pe - body) return set_err(h, MD_IO_BAD_FORMAT, "column data exceeds the sample in " + s.path);
        if ((int32_t)c == column) {
            *ptr = body;
            *nbytes = sz;
            return MD_IO_OK;
        }
        body += sz;
    }
    return set_err(h, MD_IO_BAD_ARG, "column index out of range");
}

bool read_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

}  // namespace

extern "C" {

int32_t md_io_abi_version(void) { return 1; }

int md_mds_open(const char* dir, md_mds** out) {
    if (!dir || !out) return set_err(nullptr, MD_IO_BAD_ARG, "null argument");
    *out = nullptr;
    const std::string d(dir);
    std::string text;
    if (!read_file(d + "/index.json", text)) return set_err(nullptr, MD_IO_NOT_FOUND, "cannot read " + d + "/index.json");
    JParser jp{text.data(), text.data() + text.size(), {}};
    JValue root;
    if (!jp.value(root)) return set_err(nullptr, MD_IO_BAD_FORMAT, d + "/index.json: " + jp.err);
    jp.ws();
    if (jp.p != jp.end) return set_err(nullptr, MD_IO_BAD_FORMAT, d + "/index.json: trailing characters");
    const JValue* shards = root.get("shards");
    if (!shards || shards->kind != JValue::Arr) return set_err(nullptr, MD_IO_BAD_FORMAT, d + "/index.json: no \"shards\" array");

    std::unique_ptr<md_mds> h(new md_mds);
    h->dir = d;
    h->first.push_back(0);
    for (size_t i = 0; i < shards->arr.size(); ++i) {
        const JValue& js = shards->arr[i];
        const std::string where = d + "/index.json: shard " + std::to_string(i) + ": ";
        const JValue* fmt = js.get("format");
        if (!fmt || fmt->kind != JValue::Str || fmt->str != "mds")
            return set_err(nullptr, MD_IO_UNSUPPORTED, where + "format is not \"mds\"");
        const JValue* comp = js.get("compression");
        if (comp && comp->kind != JValue::Null && !(comp->kind == JValue::Str && comp->str.empty()))
            return set_err(nullptr, MD_IO_UNSUPPORTED, where + "compressed shards are not supported");
        const JValue* names = js.get("column_names");
        const JValue* encs = js.get("column_encodings");
        const JValue* sizes = js.get("column_sizes");
        const JValue* samples = js.get("samples");
        const JValue* raw = js.get("raw_data");
        const JValue* base = raw ? raw->get("basename") : nullptr;
        if (!names || names->kind != JValue::Arr || !encs || encs->kind != JValue::Arr || !sizes || sizes->kind != JValue::Arr ||
            names->arr.size() != encs->arr.size() || names->arr.size() != sizes->arr.size() || !samples ||
            samples->kind != JValue::Num || samples->num < 0 || !base || base->kind != JValue::Str)
            return set_err(nullptr, MD_IO_BAD_FORMAT, where + "missing or inconsistent column / sample / raw_data fields");
        Shard s;
        s.path = d + "/" + base->str;
        s.samples = (int64_t)samples->num;
        const JValue* nb = raw->get("bytes");
        s.bytes = (nb && nb->kind == JValue::Num) ? (int64_t)nb->num : 0;
        std::vector<std::string> cn, ce;
        for (size_t c = 0; c < names->arr.size(); ++c) {
            if (names->arr[c].kind != JValue::Str || encs->arr[c].kind != JValue::Str)
                return set_err(nullptr, MD_IO_BAD_FORMAT, where + "column names / encodings must be strings");
            cn.push_back(names->arr[c].str);
            ce.push_back(encs->arr[c].str);
            const JValue& sz = sizes->arr[c];
            if (sz.kind == JValue::Null) s.col_sizes.push_back(-1);
            else if (sz.kind == JValue::Num && sz.num >= 0) s.col_sizes.push_back((int64_t)sz.num);
            else return set_err(nullptr, MD_IO_BAD_FORMAT, where + "bad column size");
        }
        if (i == 0) {
            h->col_names = cn;
            h->col_encodings = ce;
        } else if (cn != h->col_names || ce != h->col_encodings) {
            return set_err(nullptr, MD_IO_BAD_FORMAT, where + "columns differ from shard 0");
        }
        h->first.push_back(h->first.back() + s.samples);
        h->shards.push_back(std::move(s));
    }
    *out = h.release();
    return MD_IO_OK;
}

void md_mds_close(md_mds* h) {
    if (!h) return;
    for (auto& s : h->shards)
        if (s.map) munmap(const_cast<uint8_t*>(s.map), s.map_len);
    delete h;
}

const char* md_mds_last_error(const md_mds* h) { return h ? h->err.c_str() : g_open_error.c_str(); }

int64_t md_mds_num_samples(const md_mds* h) { return h ? h->first.back() : 0; }
int32_t md_mds_num_shards(const md_mds* h) { return h ? (int32_t)h->shards.size() : 0; }
int32_t md_mds_num_columns(const md_mds* h) { return h ? (int32_t)h->col_names.size() : 0; }

const char* md_mds_column_name(const md_mds* h, int32_t c) {
    return (h && c >= 0 && c < (int32_t)h->col_names.size()) ? h->col_names[c].c_str() : nullptr;
}
const char* md_mds_column_encoding(const md_mds* h, int32_t c) {
    return (h && c >= 0 && c < (int32_t)h->col_encodings.size()) ? h->col_encodings[c].c_str() : nullptr;
}
int32_t md_mds_column_index(const md_mds* h, const char* name) {
    if (!h || !name) return -1;
    for (size_t c = 0; c < h->col_names.size(); ++c)
        if (h->col_names[c] == name) return (int32_t)c;
    return -1;
}

int md_mds_sample_size(md_mds* h, int64_t sample, int32_t column, int64_t* nbytes) {
    if (!h || !nbytes) return set_err(h, MD_IO_BAD_ARG, "null argument");
    const uint8_t* p;
    return locate(h, sample, column, &p, nbytes);
}

int md_mds_read_sample(md_mds* h, int64_t sample, int32_t column, void* dst, int64_t cap, int64_t* nbytes) {
    if (!h || !nbytes || (!dst && cap > 0)) return set_err(h, MD_IO_BAD_ARG, "null argument");
    const uint8_t* p;
    const int rc = locate(h, sample, column, &p, nbytes);
    if (rc != MD_IO_OK) return rc;
    if (*nbytes > cap) return set_err(h, MD_IO_SIZE_MISMATCH, "destination too small for the column value");
    if (*nbytes) memcpy(dst, p, (size_t)*nbytes);
    return MD_IO_OK;
}

int md_mds_read_batch(md_mds* h, const int64_t* samples, int32_t n, int32_t column, void* dst, int64_t row_bytes,
                      int64_t row_stride, int32_t n_threads) {
    if (!h || n < 0 || (n > 0 && (!samples || !dst)) || row_bytes < 0 || row_stride < row_bytes)
        return set_err(h, MD_IO_BAD_ARG, "bad batch arguments");
    if (n == 0) return MD_IO_OK;
    std::atomic<int> status{MD_IO_OK};
    auto work = [&](int32_t lo, int32_t hi) {
        for (int32_t i = lo; i < hi && status.load(std::memory_order_relaxed) == MD_IO_OK; ++i) {
            const uint8_t* p;
            int64_t sz;
            int rc = locate(h, samples[i], column, &p, &sz);
            if (rc == MD_IO_OK && sz != row_bytes)
                rc = set_err(h, MD_IO_SIZE_MISMATCH, "sample " + std::to_string(samples[i]) + ": column has " + std::to_string(sz) +
                                                         " bytes, expected " + std::to_string(row_bytes));
            if (rc != MD_IO_OK) {
                int expect = MD_IO_OK;
                status.compare_exchange_strong(expect, rc);
                return;
            }
            memcpy(static_cast<uint8_t*>(dst) + (int64_t)i * row_stride, p, (size_t)row_bytes);
        }
    };
    int32_t nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    if (nt > n) nt = n;
    if (nt == 1) {
        work(0, n);
    } else {
        std::vector<std::thread> pool;
        pool.reserve(nt - 1);
        const int32_t per = (n + nt - 1) / nt;
        for (int32_t t = 1; t < nt; ++t) pool.emplace_back(work, std::min(n, t * per), std::min(n, (t + 1) * per));
        work(0, std::min(n, per));
        for (auto& th : pool) th.join();
    }
    return status.load();
}

}  // extern "C"
