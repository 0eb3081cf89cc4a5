// paillier.hpp -- C++ host-side mirror of sachaservan/paillier's exported API for the hot path, above the C ABI
// (include/paillier_hip.h).  Same type and method names as the Go package (PublicKey / SecretKey / Ciphertext /
// ThresholdPublicKey, EncryptWithR, Decrypt, Add, Sub, ConstMult, PartialDecrypt, CombinePartialDecryptions), with
// slice-in / slice-out batch variants -- the shape a cgo shim gives the Go package.
// Big integers cross this layer the way gmp.Int.Bytes() produces them: big-endian magnitudes.
// Header-only; link with -lpaillier_hip.  Errors become exceptions carrying pgpu_last_error().
#pragma once
#include <stdint.h>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/paillier_hip.h"

namespace paillier {

// gmp.Int stand-in at the boundary: big-endian magnitude (minimal, empty for zero), as gmp.Int.Bytes()
struct Int {
  std::vector<uint8_t> be;
  Int() {}
  explicit Int(uint64_t v) { while (v) { be.insert(be.begin(), (uint8_t)v); v >>= 8; } }
  static Int FromHex(const std::string& h) {
    Int r;
    std::string s = (h.size() % 2) ? "0" + h : h;
    for (size_t i = 0; i < s.size(); i += 2) r.be.push_back((uint8_t)std::stoul(s.substr(i, 2), nullptr, 16));
    r.trim();
    return r;
  }
  std::string Hex() const {
    static const char* d = "0123456789abcdef";
    std::string s;
    bool lead = true;
    for (uint8_t b : be) {
      if (lead && (b >> 4) == 0) { if (b == 0) continue; s += d[b & 15]; lead = false; continue; }
      s += d[b >> 4]; s += d[b & 15]; lead = false;
    }
    return s.empty() ? "0" : s;
  }
  void trim() { while (!be.empty() && be[0] == 0) be.erase(be.begin()); }
  bool operator==(const Int& o) const { return be == o.be; }
};

enum EncryptionLevel { EncLevelOne = PGPU_LEVEL_ONE, EncLevelTwo = PGPU_LEVEL_TWO };  // paillier.go:17-23

struct Error : std::runtime_error {
  int code;
  Error(int c) : std::runtime_error(pgpu_last_error()), code(c) {}
};
inline void check(int rc) { if (rc != PGPU_OK) throw Error(rc); }

inline std::vector<uint8_t> pack(const std::vector<Int>& xs, size_t stride) {
  std::vector<uint8_t> buf(xs.size() * stride, 0);
  for (size_t i = 0; i < xs.size(); ++i) {
    if (xs[i].be.size() > stride) throw std::runtime_error("operand wider than its stride");
    std::copy(xs[i].be.begin(), xs[i].be.end(), buf.begin() + (i + 1) * stride - xs[i].be.size());
  }
  return buf;
}
inline std::vector<Int> unpack(const std::vector<uint8_t>& buf, size_t stride) {
  std::vector<Int> out(stride ? buf.size() / stride : 0);
  for (size_t i = 0; i < out.size(); ++i) { out[i].be.assign(buf.begin() + i * stride, buf.begin() + (i + 1) * stride); out[i].trim(); }
  return out;
}

class GPU {
 public:
  explicit GPU(int device = 0) { check(pgpu_ctx_create(device, nullptr, &ctx_)); }
  ~GPU() { pgpu_ctx_destroy(ctx_); }
  GPU(const GPU&) = delete;
  pgpu_ctx* ctx() const { return ctx_; }
 private:
  pgpu_ctx* ctx_ = nullptr;
};

struct Ciphertext { Int C; EncryptionLevel Level = EncLevelOne; };  // paillier.go:65-69

class PublicKey {  // paillier.go:46-57
 public:
  Int N, G;
  PublicKey(GPU& gpu, const Int& n, const Int& g) : N(n), G(g) {
    check(pgpu_pubkey_create(gpu.ctx(), n.be.data(), n.be.size(), g.be.data(), g.be.size(), nullptr, 0, nullptr, 0, &h_));
  }
  virtual ~PublicKey() { pgpu_pubkey_destroy(h_); }
  PublicKey(const PublicKey&) = delete;
  pgpu_pubkey* handle() const { return h_; }
  size_t PlainBytes(EncryptionLevel l = EncLevelOne) const { return pgpu_pubkey_plain_bytes(h_, l); }
  size_t CipherBytes(EncryptionLevel l = EncLevelOne) const { return pgpu_pubkey_cipher_bytes(h_, l); }

  // for i: EncryptWithRAtLevel(ms[i], rs[i], level)          paillier.go:206-218
  std::vector<Ciphertext> EncryptWithRBatch(const std::vector<Int>& ms, const std::vector<Int>& rs,
                                            EncryptionLevel level = EncLevelOne) const {
    size_t pb = PlainBytes(level), cb = CipherBytes(level);
    auto m = pack(ms, pb), r = pack(rs, pb);
    std::vector<uint8_t> c(ms.size() * cb);
    check(pgpu_encrypt_with_r(h_, level, ms.size(), m.data(), pb, r.data(), pb, c.data(), cb, PGPU_MEM_HOST));
    std::vector<Ciphertext> out;
    for (auto& v : unpack(c, cb)) out.push_back({v, level});
    return out;
  }
  // element-wise Add / Sub of two ciphertext vectors         operations.go:11-55
  std::vector<Ciphertext> AddBatch(const std::vector<Ciphertext>& a, const std::vector<Ciphertext>& b) const { return bin(a, b, false); }
  std::vector<Ciphertext> SubBatch(const std::vector<Ciphertext>& a, const std::vector<Ciphertext>& b) const { return bin(a, b, true); }
  // for i: ConstMult(cts[i], k)                              operations.go:58-64
  std::vector<Ciphertext> ConstMultBatch(const std::vector<Ciphertext>& cts, const Int& k) const {
    EncryptionLevel level = cts.at(0).Level;
    size_t cb = CipherBytes(level);
    std::vector<Int> cs;
    for (auto& c : cts) cs.push_back(c.C);
    auto in = pack(cs, cb);
    std::vector<uint8_t> out(cts.size() * cb);
    std::vector<uint8_t> kb = k.be.empty() ? std::vector<uint8_t>{0} : k.be;
    check(pgpu_const_mult(h_, level, cts.size(), in.data(), cb, kb.data(), kb.size(), 0, out.data(), cb, PGPU_MEM_HOST));
    std::vector<Ciphertext> res;
    for (auto& v : unpack(out, cb)) res.push_back({v, level});
    return res;
  }

 protected:
  pgpu_pubkey* h_ = nullptr;

 private:
  std::vector<Ciphertext> bin(const std::vector<Ciphertext>& a, const std::vector<Ciphertext>& b, bool sub) const {
    EncryptionLevel level = a.at(0).Level;
    size_t cb = CipherBytes(level);
    std::vector<Int> as, bs;
    for (auto& c : a) as.push_back(c.C);
    for (auto& c : b) bs.push_back(c.C);
    auto pa = pack(as, cb), pb = pack(bs, cb);
    std::vector<uint8_t> out(a.size() * cb);
    check(sub ? pgpu_sub(h_, level, a.size(), pa.data(), cb, pb.data(), cb, out.data(), cb, PGPU_MEM_HOST, nullptr)
              : pgpu_add(h_, level, a.size(), pa.data(), cb, pb.data(), cb, out.data(), cb, PGPU_MEM_HOST));
    std::vector<Ciphertext> res;
    for (auto& v : unpack(out, cb)) res.push_back({v, level});
    return res;
  }
};

class SecretKey {  // paillier.go:60-63
 public:
  SecretKey(GPU& gpu, const PublicKey& pk, const Int& lambda) : pk_(pk) {
    check(pgpu_seckey_create(gpu.ctx(), pk.handle(), lambda.be.data(), lambda.be.size(), &h_));
  }
  ~SecretKey() { pgpu_seckey_destroy(h_); }
  SecretKey(const SecretKey&) = delete;
  // for i: Decrypt(cts[i])                                   paillier.go:292-303
  std::vector<Int> DecryptBatch(const std::vector<Ciphertext>& cts) const {
    EncryptionLevel level = cts.at(0).Level;
    size_t pb = pk_.PlainBytes(level), cb = pk_.CipherBytes(level);
    std::vector<Int> cs;
    for (auto& c : cts) cs.push_back(c.C);
    auto in = pack(cs, cb);
    std::vector<uint8_t> out(cts.size() * pb);
    check(pgpu_decrypt(h_, level, cts.size(), in.data(), cb, out.data(), pb, PGPU_MEM_HOST, PGPU_DECRYPT_DEFAULT, nullptr));
    return unpack(out, pb);
  }
 private:
  const PublicKey& pk_;
  pgpu_seckey* h_ = nullptr;
};

struct PartialDecryption { int ID; std::vector<Int> Decryption; };  // thresholdkey.go:44-47 (one value per ciphertext)

class ThresholdPublicKey : public PublicKey {  // thresholdkey.go:26-32
 public:
  int TotalNumberOfDecryptionServers, Threshold;
  ThresholdPublicKey(GPU& gpu, const Int& n, const Int& g, int total, int threshold)
      : PublicKey(gpu, n, g), TotalNumberOfDecryptionServers(total), Threshold(threshold) {}
  // ThresholdSecretKey.PartialDecrypt for each ciphertext     thresholdkey.go:192-201
  PartialDecryption PartialDecryptBatch(int id, const Int& share, const std::vector<Int>& cs) const {
    size_t cb = CipherBytes();
    auto in = pack(cs, cb);
    std::vector<uint8_t> out(cs.size() * cb);
    check(pgpu_partial_decrypt(h_, TotalNumberOfDecryptionServers, share.be.data(), share.be.size(), cs.size(), in.data(), cb,
                               out.data(), cb, PGPU_MEM_HOST));
    return {id, unpack(out, cb)};
  }
  // CombinePartialDecryptions for each ciphertext             thresholdkey.go:149-161
  std::vector<Int> CombinePartialDecryptionsBatch(const std::vector<PartialDecryption>& shares) const {
    size_t cb = CipherBytes(), pb = PlainBytes();
    std::vector<std::vector<uint8_t>> bufs;
    std::vector<const uint8_t*> ptrs;
    std::vector<int> ids;
    for (auto& s : shares) { bufs.push_back(pack(s.Decryption, cb)); ids.push_back(s.ID); }
    for (auto& b : bufs) ptrs.push_back(b.data());
    size_t batch = shares.empty() ? 0 : shares[0].Decryption.size();
    std::vector<uint8_t> out(batch * pb);
    check(pgpu_combine_partial_decryptions(h_, TotalNumberOfDecryptionServers, Threshold, (int)shares.size(), ids.data(), batch,
                                           ptrs.data(), cb, out.data(), pb, PGPU_MEM_HOST, nullptr));
    return unpack(out, pb);
  }
};

}  // namespace paillier
