// wire.cpp -- the reference's wire format for ciphertexts at the C ABI: Ciphertext.Bytes() /
// PublicKey.NewCiphertextFromBytes (paillier.go:374-401) = encoding/gob of
//
//     type Ciphertext struct { C *gmp.Int; Level EncryptionLevel; EncMethod EncryptionMethod }
//
// written by a FRESH gob.Encoder per ciphertext, so every blob carries its type definitions:
//
//     message 1   type -65: struct "Ciphertext" { C: type 66, Level: int, EncMethod: int }
//     message 2   type -66: GobEncoder type "Int"        (gmp.Int implements gob.GobEncoder, as math/big.Int does)
//     message 3   value of type 65: field C = GobEncode() bytes = [version 1 << 1 | sign] ++ big-endian magnitude; zero-valued
//                 Level / EncMethod are omitted, as gob omits them
//
// A batch of blobs <-> the ABI's flat fixed-stride big-endian buffer.  The walk over the gob messages (a few dozen varints per
// blob) runs on host threads; the payload bytes move on the device when the flat buffer lives in HBM (k_bytes_gather_be /
// k_gob_emit) -- blobs that arrive from the network go straight to where pgpu_decrypt(..., PGPU_MEM_DEVICE) reads them.
//
// Status: written from the gob specification and math/big's GobEncode layout, byte-identical to the
// Python restatement paillier_amd/wire.py (tests/test_gpu_wire.py); NOT cross-checked against a Go toolchain (none here): parity
// with Go's own output is unpinned.  The decoder accepts any type ids and field order, as gob does (fields match by NAME), and
// skips value fields Ciphertext lacks when they are of gob's basic types or GobEncoder values; anything else it cannot walk
// (nested structs, maps, slices as extra fields) is an error, where Go's decoder would skip those too.
#include "engine.hpp"

namespace pgi {

namespace {

constexpr uint32_t kFirstUserId = 65;   // gob numbers user types from 65 in order of first use in the process
constexpr uint32_t kTypeInt = 2;        // gob's built-in id of int

void put_uint(std::vector<uint8_t>& o, uint64_t v) {
  if (v < 128) { o.push_back((uint8_t)v); return; }
  int nb = 0;
  for (uint64_t t = v; t; t >>= 8) ++nb;
  o.push_back((uint8_t)(256 - nb));
  for (int i = nb - 1; i >= 0; --i) o.push_back((uint8_t)(v >> (8 * i)));
}
void put_int(std::vector<uint8_t>& o, int64_t v) { put_uint(o, v < 0 ? ((uint64_t)(~v) << 1) | 1 : (uint64_t)v << 1); }
void put_string(std::vector<uint8_t>& o, const char* s) {
  const size_t n = strlen(s);
  put_uint(o, n);
  o.insert(o.end(), s, s + n);
}
void put_common(std::vector<uint8_t>& o, const char* name, int64_t id) {      // CommonType{Name, Id}
  o.push_back(1); put_string(o, name); o.push_back(1); put_int(o, id); o.push_back(0);
}
void put_message(std::vector<uint8_t>& o, const std::vector<uint8_t>& body) {
  put_uint(o, body.size());
  o.insert(o.end(), body.begin(), body.end());
}

// the two type-definition messages every blob starts with
std::vector<uint8_t> gob_prefix() {
  std::vector<uint8_t> out, b;
  put_int(b, -(int64_t)kFirstUserId);                   // (-id, wireType{StructT: ...}): field 3 of wireType
  b.push_back(3); b.push_back(1);
  put_common(b, "Ciphertext", kFirstUserId);
  b.push_back(1); put_uint(b, 3);
  const struct { const char* name; int64_t id; } fields[3] = {{"C", kFirstUserId + 1}, {"Level", kTypeInt}, {"EncMethod", kTypeInt}};
  for (auto& f : fields) { b.push_back(1); put_string(b, f.name); b.push_back(1); put_int(b, f.id); b.push_back(0); }
  b.push_back(0); b.push_back(0);
  put_message(out, b);
  b.clear();
  put_int(b, -(int64_t)(kFirstUserId + 1));             // (-id, wireType{GobEncoderT: ...}): field 5 of wireType
  b.push_back(5); b.push_back(1);
  put_common(b, "Int", kFirstUserId + 1);
  b.push_back(0); b.push_back(0);
  put_message(out, b);
  return out;
}
// value message = head | uint(len GobEncode) | GobEncode | tail
std::vector<uint8_t> gob_head() {
  std::vector<uint8_t> h;
  put_int(h, kFirstUserId);
  h.push_back(1);                                        // field delta: C
  return h;
}
std::vector<uint8_t> gob_tail(int level, int method) {
  std::vector<uint8_t> t;
  uint64_t delta = 1;
  for (int v : {level, method}) {
    if (v) { put_uint(t, delta); put_int(t, v); delta = 1; }
    else ++delta;
  }
  t.push_back(0);
  return t;
}
size_t uint_len(uint64_t v) {
  if (v < 128) return 1;
  size_t nb = 0;
  for (; v; v >>= 8) ++nb;
  return 1 + nb;
}

struct Reader {
  const uint8_t* d;
  size_t n, i = 0;
  Reader(const uint8_t* d_, size_t n_) : d(d_), n(n_) {}
  uint8_t byte() {
    if (i >= n) api_throw(PGPU_ERR_INVALID, "gob: unexpected end of data");
    return d[i++];
  }
  const uint8_t* take(size_t k) {
    if (k > n - i) api_throw(PGPU_ERR_INVALID, "gob: unexpected end of data");
    i += k;
    return d + i - k;
  }
  uint64_t uint() {
    const uint8_t b = byte();
    if (b < 128) return b;
    const int k = 256 - b;
    if (k > 8) api_throw(PGPU_ERR_INVALID, "gob: bad unsigned integer");
    uint64_t v = 0;
    const uint8_t* p = take((size_t)k);
    for (int j = 0; j < k; ++j) v = (v << 8) | p[j];
    return v;
  }
  int64_t sint() {
    const uint64_t u = uint();
    return (u & 1) ? ~(int64_t)(u >> 1) : (int64_t)(u >> 1);
  }
  bool done() const { return i >= n; }
};

// Field deltas are 64-bit integers straight off the wire: the index advances in unsigned arithmetic and is checked against the
// number of fields BEFORE anything is indexed (a delta of 2^64 - 1 must not wrap the index to -2).  f is -1 before the first field.
size_t next_field(long& f, uint64_t d, size_t nfields, const char* what) {
  const uint64_t next = (uint64_t)(f + 1) + d;                                        // = index + 1; f + 1 >= 0, d >= 1
  if (d > (uint64_t)nfields || next > (uint64_t)nfields) api_throw(PGPU_ERR_INVALID, std::string("gob: ") + what + " field index out of range");
  f = (long)(next - 1);
  return (size_t)f;
}

void read_common(Reader& r, std::string* name) {
  long f = -1;
  for (;;) {
    const uint64_t d = r.uint();
    if (d == 0) return;
    next_field(f, d, 2, "CommonType");
    if (f == 0) { const uint64_t k = r.uint(); const uint8_t* p = r.take((size_t)k); if (name) name->assign((const char*)p, (size_t)k); }
    else (void)r.sint();
  }
}

struct Parsed { size_t off = 0, len = 0; int32_t level = 0, method = 0; bool has_c = false; };

// NewCiphertextFromBytes (paillier.go:376-391) for one blob: where the magnitude of C sits, Level, EncMethod
Parsed gob_parse(const uint8_t* data, size_t n) {
  if (n == 0) api_throw(PGPU_ERR_INVALID, "no data provided");                      // paillier.go:377
  Reader r(data, n);
  struct Field { std::string name; int64_t id; };
  std::vector<std::pair<int64_t, std::vector<Field>>> structs;
  std::vector<int64_t> gobenc;
  while (!r.done()) {
    const uint64_t blen = r.uint();
    const uint8_t* bp = r.take((size_t)blen);
    Reader body(bp, (size_t)blen);
    const int64_t tid = body.sint();
    if (tid == INT64_MIN) api_throw(PGPU_ERR_INVALID, "gob: bad type id");
    if (tid < 0) {                                                                    // a type definition
      const uint64_t f = body.uint();
      if (f == 3) {                                                                   // StructT
        std::vector<Field> fields;
        long g = -1;
        for (;;) {
          const uint64_t d = body.uint();
          if (d == 0) break;
          next_field(g, d, 2, "structType");                                          // {CommonType, Field []fieldType}
          if (g == 0) read_common(body, nullptr);
          else {
            const uint64_t nf = body.uint();
            if (nf > blen) api_throw(PGPU_ERR_INVALID, "gob: more fields than bytes");   // (every field takes >= 1 byte)
            for (uint64_t k = 0; k < nf; ++k) {
              Field fd{"", 0};
              long h = -1;
              for (;;) {
                const uint64_t d2 = body.uint();
                if (d2 == 0) break;
                next_field(h, d2, 2, "fieldType");                                    // {Name string, Id typeId}
                if (h == 0) { const uint64_t l = body.uint(); const uint8_t* p = body.take((size_t)l); fd.name.assign((const char*)p, (size_t)l); }
                else fd.id = body.sint();
              }
              fields.push_back(fd);
            }
          }
        }
        structs.push_back({-tid, fields});
      } else if (f == 5) {                                                            // GobEncoderT
        (void)body.uint();
        read_common(body, nullptr);
        gobenc.push_back(-tid);
      } else {
        api_throw(PGPU_ERR_INVALID, "gob: unsupported wire type");
      }
      continue;
    }
    const std::vector<Field>* fields = nullptr;
    for (auto& s : structs) if (s.first == tid) fields = &s.second;
    if (!fields) api_throw(PGPU_ERR_INVALID, "gob: value of an undefined type");
    Parsed out;
    long f = -1;
    for (;;) {
      const uint64_t d = body.uint();
      if (d == 0) break;
      const Field& fd = (*fields)[next_field(f, d, fields->size(), "value")];
      if (fd.name == "C") {
        if (std::find(gobenc.begin(), gobenc.end(), fd.id) == gobenc.end()) api_throw(PGPU_ERR_INVALID, "gob: field C is not a GobEncoder type");
        const uint64_t l = body.uint();
        const uint8_t* p = body.take((size_t)l);
        if (l == 0 || (p[0] >> 1) != 1) api_throw(PGPU_ERR_INVALID, "Int.GobDecode: encoding version not supported");
        if (p[0] & 1) api_throw(PGPU_ERR_INVALID, "gob: negative C (a ciphertext is a residue)");
        size_t z = 1;
        while (z < l && p[z] == 0) ++z;                                               // (a non-minimal magnitude: skip zeros)
        out.off = (size_t)(p - data) + z;
        out.len = (size_t)l - z;
        out.has_c = true;
      } else if (fd.name == "Level") {
        out.level = (int32_t)body.sint();
      } else if (fd.name == "EncMethod") {
        out.method = (int32_t)body.sint();
      } else {
        // a wire field the receiver's struct lacks: Go's decoder skips its value and goes on; so does this one, for the types
        // whose extent is known without their definition (gob's basic types and GobEncoder values)
        if (std::find(gobenc.begin(), gobenc.end(), fd.id) != gobenc.end() || fd.id == 5 || fd.id == 6) (void)body.take((size_t)body.uint());
        else if (fd.id >= 1 && fd.id <= 4) (void)body.uint();                        // bool, int, uint, float: one varint
        else if (fd.id == 7) { (void)body.uint(); (void)body.uint(); }               // complex
        else api_throw(PGPU_ERR_INVALID, "gob: cannot skip field " + fd.name + " (not in Ciphertext, not a basic type)");
      }
    }
    if (!out.has_c) api_throw(PGPU_ERR_INVALID, "gob: the value has no field C");     // (Go: a Ciphertext with a nil C)
    return out;
  }
  api_throw(PGPU_ERR_INVALID, "gob: no value in the data");
}

// f(lo, hi) over [0, total) on up to 16 host threads; the first error text wins
template <class F> void on_host_threads(size_t total, size_t grain, F&& f) {
  const size_t nthreads = std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)std::thread::hardware_concurrency(), total / grain + 1}));
  if (nthreads == 1) { f((size_t)0, total); return; }
  std::vector<std::thread> th;
  std::vector<std::string> errs(nthreads);
  th.reserve(nthreads);
  // a std::thread that cannot be started throws with the earlier ones still joinable: join them on every way out
  struct Joiner { std::vector<std::thread>& th; ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); } } joiner{th};
  for (size_t t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      try { f(total * t / nthreads, total * (t + 1) / nthreads); }
      catch (const ApiError& e) { errs[t] = e.msg.empty() ? "gob error" : e.msg; }
      catch (const std::exception& e) { errs[t] = std::string("gob: ") + e.what(); }
      catch (...) { errs[t] = "gob: unknown error"; }
    });
  for (auto& t : th) t.join();
  for (auto& e : errs) if (!e.empty()) api_throw(PGPU_ERR_INVALID, e);
}

}  // namespace
}  // namespace pgi

extern "C" {

size_t pgpu_gob_max_bytes(size_t value_bytes) {
  // prefix + length of the value message (<= 9) + head + length of GobEncode (<= 9) + version byte + magnitude + tail (<= 23)
  static const size_t fixed = gob_prefix().size() + gob_head().size();
  return fixed + 9 + 9 + 1 + value_bytes + 23;
}

int pgpu_gob_unpack(pgpu_ctx* ctx, size_t batch, const uint8_t* blobs, const size_t* offsets, uint8_t* out, size_t out_stride,
                    int mem, int32_t* levels, int32_t* methods) {
  if (!blobs || !offsets || !out) return fail(PGPU_ERR_INVALID, "null argument");
  if (!ctx && mem != PGPU_MEM_HOST) return fail(PGPU_ERR_INVALID, "a device buffer needs a context");
  return guarded([&] {
    if (batch == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
    if (batch > (1u << 26)) api_throw(PGPU_ERR_INVALID, "batch too large");
    for (size_t i = 0; i < batch; ++i)
      if (offsets[i + 1] < offsets[i]) api_throw(PGPU_ERR_INVALID, "blob offsets must not decrease");
    std::vector<uint64_t> off(batch);
    std::vector<uint32_t> len(batch);
    on_host_threads(batch, 4096, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) {
        const Parsed p = gob_parse(blobs + offsets[i], offsets[i + 1] - offsets[i]);
        if (p.len > out_stride) api_throw(PGPU_ERR_INVALID, "gob: C is wider than the output stride");
        off[i] = (uint64_t)(offsets[i] + p.off);
        len[i] = (uint32_t)p.len;
        if (levels) levels[i] = p.level;
        if (methods) methods[i] = p.method;
      }
    });
    if (mem == PGPU_MEM_HOST) {
      // both sides in host memory: the bytes never need the device
      on_host_threads(batch, 4096, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
          uint8_t* d = out + i * out_stride;
          memset(d, 0, out_stride - len[i]);
          memcpy(d + (out_stride - len[i]), blobs + off[i], len[i]);
        }
      });
      return;
    }
    ctx->bind();
    ctx->reset_ws();
    const size_t total = offsets[batch] - offsets[0];
    uint8_t* d_blobs = (uint8_t*)ctx->ws(total);
    uint64_t* d_off = (uint64_t*)ctx->ws(batch * 8);
    uint32_t* d_len = (uint32_t*)ctx->ws(batch * 4);
    for (auto& o : off) o -= offsets[0];
    HIPCHK(hipMemcpyAsync(d_blobs, blobs + offsets[0], total, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_off, off.data(), batch * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_len, len.data(), batch * 4, hipMemcpyHostToDevice, ctx->stream));
    launch_bytes_gather_be(d_blobs, d_off, d_len, batch, out, out_stride, ctx->stream);
    HIPCHK(hipStreamSynchronize(ctx->stream));      // (off / len are read by the copies until here)
  });
}

int pgpu_gob_pack(pgpu_ctx* ctx, size_t batch, const uint8_t* in, size_t stride, int mem, int level, int enc_method, uint8_t* blobs,
                  size_t blobs_cap, size_t* offsets) {
  if (!in || !blobs || !offsets) return fail(PGPU_ERR_INVALID, "null argument");
  if (!ctx && mem != PGPU_MEM_HOST) return fail(PGPU_ERR_INVALID, "a device buffer needs a context");
  return guarded([&] {
    if (batch == 0 || stride == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
    if (batch > (1u << 26)) api_throw(PGPU_ERR_INVALID, "batch too large");
    const std::vector<uint8_t> prefix = gob_prefix(), head = gob_head(), tail = gob_tail(level, enc_method);
    std::vector<uint32_t> len(batch);
    const uint8_t* d_in = in;
    uint32_t* d_len = nullptr;
    if (mem == PGPU_MEM_HOST) {
      on_host_threads(batch, 4096, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
          const uint8_t* p = in + i * stride;
          size_t z = 0;
          while (z < stride && p[z] == 0) ++z;
          len[i] = (uint32_t)(stride - z);
        }
      });
    } else {
      ctx->bind();
      ctx->reset_ws();
      d_len = (uint32_t*)ctx->ws(batch * 4);
      launch_be_lengths(d_in, stride, batch, d_len, ctx->stream);
      HIPCHK(hipMemcpyAsync(len.data(), d_len, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    // sizes: prefix | uint(L) | head | uint(glen) | GobEncode (glen = 1 + magnitude) | tail, L = everything after uint(L)
    offsets[0] = 0;
    for (size_t i = 0; i < batch; ++i) {
      const size_t glen = (size_t)len[i] + 1;
      const size_t L = head.size() + uint_len(glen) + glen + tail.size();
      offsets[i + 1] = offsets[i] + prefix.size() + uint_len(L) + L;
    }
    if (offsets[batch] > blobs_cap) api_throw(PGPU_ERR_INVALID, "blob buffer too small (pgpu_gob_max_bytes(stride) per ciphertext always fits)");
    if (mem == PGPU_MEM_HOST) {
      on_host_threads(batch, 4096, [&](size_t lo, size_t hi) {
        std::vector<uint8_t> v;
        for (size_t i = lo; i < hi; ++i) {
          const size_t n = len[i], glen = n + 1;
          uint8_t* d = blobs + offsets[i];
          memcpy(d, prefix.data(), prefix.size());
          v.clear();
          put_uint(v, head.size() + uint_len(glen) + glen + tail.size());
          v.insert(v.end(), head.begin(), head.end());
          put_uint(v, glen);
          v.push_back(2);                                                            // version 1 << 1 | sign 0
          memcpy(d + prefix.size(), v.data(), v.size());
          memcpy(d + prefix.size() + v.size(), in + i * stride + (stride - n), n);
          memcpy(d + prefix.size() + v.size() + n, tail.data(), tail.size());
        }
      });
      return;
    }
    std::vector<uint64_t> off64(offsets, offsets + batch);
    uint64_t* d_off = (uint64_t*)ctx->ws(batch * 8);
    uint8_t* d_const = (uint8_t*)ctx->ws(prefix.size() + head.size() + tail.size());
    uint8_t* d_blobs = (uint8_t*)ctx->ws(offsets[batch]);
    std::vector<uint8_t> consts(prefix);
    consts.insert(consts.end(), head.begin(), head.end());
    consts.insert(consts.end(), tail.begin(), tail.end());
    HIPCHK(hipMemcpyAsync(d_off, off64.data(), batch * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_const, consts.data(), consts.size(), hipMemcpyHostToDevice, ctx->stream));
    launch_gob_emit(d_in, stride, d_len, d_off, batch, d_const, (uint32_t)prefix.size(), d_const + prefix.size(), (uint32_t)head.size(),
                    d_const + prefix.size() + head.size(), (uint32_t)tail.size(), d_blobs, ctx->stream);
    HIPCHK(hipMemcpyAsync(blobs, d_blobs, offsets[batch], hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

}  // extern "C"
