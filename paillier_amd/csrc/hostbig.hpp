// hostbig.hpp -- small arbitrary-precision unsigned integer for SETUP-TIME host work only:
// Montgomery constants (R^2 mod N, -N^-1 mod 2^28), CRT constants, key-derived exponents,
// Lagrange coefficients.  Nothing here runs per ciphertext; per-ciphertext arithmetic is
// exclusively in the HIP kernels.  Not a product CPU path: every batch entry point of the C ABI
// launches kernels and fails if there is no GPU.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

namespace hostbig {

struct BigU {
  std::vector<uint32_t> d;  // little-endian 32-bit digits, no leading zeros (empty == 0)

  BigU() {}
  BigU(uint64_t v) {
    while (v) { d.push_back((uint32_t)v); v >>= 32; }
  }
  void trim() { while (!d.empty() && d.back() == 0) d.pop_back(); }
  bool is_zero() const { return d.empty(); }
  bool is_odd() const { return !d.empty() && (d[0] & 1); }
  size_t bit_length() const {
    if (d.empty()) return 0;
    return (d.size() - 1) * 32 + (32 - __builtin_clz(d.back()));
  }
  bool bit(size_t i) const { return (i / 32 < d.size()) && ((d[i / 32] >> (i % 32)) & 1); }
  uint64_t low64() const { return (d.size() > 0 ? d[0] : 0) | ((uint64_t)(d.size() > 1 ? d[1] : 0) << 32); }

  static BigU from_be(const uint8_t* p, size_t n) {
    BigU r;
    r.d.assign((n + 3) / 4, 0);
    for (size_t i = 0; i < n; ++i) {
      size_t pos = n - 1 - i;  // byte significance
      r.d[pos / 4] |= (uint32_t)p[i] << (8 * (pos % 4));
    }
    r.trim();
    return r;
  }
  // fixed-width big-endian; throws if it does not fit
  void to_be(uint8_t* p, size_t n) const {
    if ((bit_length() + 7) / 8 > n) throw std::runtime_error("to_be: value does not fit");
    for (size_t i = 0; i < n; ++i) {
      size_t pos = n - 1 - i;
      p[i] = (pos / 4 < d.size()) ? (uint8_t)(d[pos / 4] >> (8 * (pos % 4))) : 0;
    }
  }
  std::vector<uint8_t> to_be_min() const {
    std::vector<uint8_t> v((bit_length() + 7) / 8);
    to_be(v.data(), v.size());
    return v;
  }
  // limbs of `lb` bits (lb <= 32), exactly `n` of them; throws if it does not fit
  std::vector<uint32_t> to_limbs(int lb, size_t n) const {
    if (bit_length() > (size_t)lb * n) throw std::runtime_error("to_limbs: value does not fit");
    std::vector<uint32_t> out(n, 0);
    for (size_t i = 0; i < n; ++i) {
      size_t bitpos = i * lb;
      size_t w = bitpos / 32, s = bitpos % 32;
      uint64_t v = 0;
      if (w < d.size()) v = d[w];
      if (w + 1 < d.size()) v |= (uint64_t)d[w + 1] << 32;
      out[i] = (uint32_t)((v >> s) & ((1ull << lb) - 1));
    }
    return out;
  }
  static BigU from_limbs(const uint32_t* l, int lb, size_t n) {
    BigU r;
    r.d.assign((n * lb + 31) / 32 + 1, 0);
    for (size_t i = 0; i < n; ++i) {
      size_t bitpos = i * lb;
      size_t w = bitpos / 32, s = bitpos % 32;
      uint64_t v = (uint64_t)l[i] << s;  // limbs may be lazy (> 2^lb): add with carry
      size_t k = w;
      while (v) {
        uint64_t sum = (uint64_t)r.d[k] + (v & 0xFFFFFFFFu);
        r.d[k] = (uint32_t)sum;
        v = (v >> 32) + (sum >> 32);
        ++k;
        if (k >= r.d.size() && v) r.d.push_back(0);
      }
    }
    r.trim();
    return r;
  }
};

inline int cmp(const BigU& a, const BigU& b) {
  if (a.d.size() != b.d.size()) return a.d.size() < b.d.size() ? -1 : 1;
  for (size_t i = a.d.size(); i-- > 0;)
    if (a.d[i] != b.d[i]) return a.d[i] < b.d[i] ? -1 : 1;
  return 0;
}
inline bool operator==(const BigU& a, const BigU& b) { return cmp(a, b) == 0; }
inline bool operator!=(const BigU& a, const BigU& b) { return cmp(a, b) != 0; }
inline bool operator<(const BigU& a, const BigU& b) { return cmp(a, b) < 0; }
inline bool operator<=(const BigU& a, const BigU& b) { return cmp(a, b) <= 0; }

inline BigU operator+(const BigU& a, const BigU& b) {
  BigU r;
  size_t n = std::max(a.d.size(), b.d.size());
  r.d.resize(n + 1);
  uint64_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c += (uint64_t)(i < a.d.size() ? a.d[i] : 0) + (i < b.d.size() ? b.d[i] : 0);
    r.d[i] = (uint32_t)c;
    c >>= 32;
  }
  r.d[n] = (uint32_t)c;
  r.trim();
  return r;
}
// a - b, requires a >= b
inline BigU operator-(const BigU& a, const BigU& b) {
  if (cmp(a, b) < 0) throw std::runtime_error("BigU: negative result");
  BigU r;
  r.d.resize(a.d.size());
  int64_t br = 0;
  for (size_t i = 0; i < a.d.size(); ++i) {
    int64_t v = (int64_t)a.d[i] - (i < b.d.size() ? b.d[i] : 0) - br;
    br = v < 0;
    r.d[i] = (uint32_t)(v + (br ? (1ll << 32) : 0));
  }
  r.trim();
  return r;
}
inline BigU operator*(const BigU& a, const BigU& b) {
  BigU r;
  if (a.is_zero() || b.is_zero()) return r;
  r.d.assign(a.d.size() + b.d.size(), 0);
  for (size_t i = 0; i < a.d.size(); ++i) {
    uint64_t c = 0;
    for (size_t j = 0; j < b.d.size(); ++j) {
      c += (uint64_t)a.d[i] * b.d[j] + r.d[i + j];
      r.d[i + j] = (uint32_t)c;
      c >>= 32;
    }
    r.d[i + b.d.size()] = (uint32_t)c;
  }
  r.trim();
  return r;
}
inline BigU shl(const BigU& a, size_t s) {
  if (a.is_zero()) return a;
  BigU r;
  size_t w = s / 32, b = s % 32;
  r.d.assign(a.d.size() + w + 1, 0);
  for (size_t i = 0; i < a.d.size(); ++i) {
    uint64_t v = (uint64_t)a.d[i] << b;
    r.d[i + w] |= (uint32_t)v;
    r.d[i + w + 1] |= (uint32_t)(v >> 32);
  }
  r.trim();
  return r;
}
inline BigU shr(const BigU& a, size_t s) {
  size_t w = s / 32, b = s % 32;
  BigU r;
  if (w >= a.d.size()) return r;
  r.d.assign(a.d.size() - w, 0);
  for (size_t i = w; i < a.d.size(); ++i) {
    uint64_t v = a.d[i];
    if (i + 1 < a.d.size()) v |= (uint64_t)a.d[i + 1] << 32;
    r.d[i - w] = (uint32_t)(v >> b);
  }
  r.trim();
  return r;
}

// Knuth algorithm D.  q = floor(a / b), r = a mod b.
inline void divmod(const BigU& a, const BigU& b, BigU& q, BigU& r) {
  if (b.is_zero()) throw std::runtime_error("BigU: division by zero");
  if (cmp(a, b) < 0) { q = BigU(); r = a; return; }
  if (b.d.size() == 1) {
    uint64_t rem = 0, dv = b.d[0];
    q.d.assign(a.d.size(), 0);
    for (size_t i = a.d.size(); i-- > 0;) {
      uint64_t cur = (rem << 32) | a.d[i];
      q.d[i] = (uint32_t)(cur / dv);
      rem = cur % dv;
    }
    q.trim();
    r = BigU(rem);
    return;
  }
  int s = __builtin_clz(b.d.back());
  BigU u = shl(a, s), v = shl(b, s);
  const size_t n = v.d.size();          // == b.d.size(): the shift is < 32 bits and b's top digit becomes >= 2^31
  u.d.resize(a.d.size() + 1, 0);        // exactly one extra digit
  const size_t m = a.d.size() - n;
  q.d.assign(m + 1, 0);
  const uint64_t B = 1ull << 32;
  for (size_t jj = m + 1; jj-- > 0;) {
    size_t j = jj;
    uint64_t num = ((uint64_t)u.d[j + n] << 32) | u.d[j + n - 1];
    uint64_t qhat = num / v.d[n - 1], rhat = num % v.d[n - 1];
    while (qhat >= B || qhat * v.d[n - 2] > ((rhat << 32) | u.d[j + n - 2])) {
      --qhat;
      rhat += v.d[n - 1];
      if (rhat >= B) break;
    }
    int64_t borrow = 0;
    uint64_t carry = 0;
    for (size_t i = 0; i < n; ++i) {
      uint64_t p = qhat * v.d[i] + carry;
      carry = p >> 32;
      int64_t t = (int64_t)u.d[i + j] - borrow - (int64_t)(p & 0xFFFFFFFFu);
      borrow = t < 0;
      u.d[i + j] = (uint32_t)t;
    }
    int64_t t = (int64_t)u.d[j + n] - borrow - (int64_t)carry;
    borrow = t < 0;
    u.d[j + n] = (uint32_t)t;
    if (borrow) {
      --qhat;
      uint64_t c = 0;
      for (size_t i = 0; i < n; ++i) {
        c += (uint64_t)u.d[i + j] + v.d[i];
        u.d[i + j] = (uint32_t)c;
        c >>= 32;
      }
      u.d[j + n] += (uint32_t)c;
    }
    q.d[j] = (uint32_t)qhat;
  }
  q.trim();
  u.d.resize(n);
  u.trim();
  r = shr(u, s);
}
inline BigU operator/(const BigU& a, const BigU& b) { BigU q, r; divmod(a, b, q, r); return q; }
inline BigU operator%(const BigU& a, const BigU& b) { BigU q, r; divmod(a, b, q, r); return r; }

inline BigU mulmod(const BigU& a, const BigU& b, const BigU& m) { return (a * b) % m; }

inline BigU powmod(const BigU& base, const BigU& e, const BigU& m) {
  BigU r(1), b = base % m;
  r = r % m;
  for (size_t i = e.bit_length(); i-- > 0;) {
    r = mulmod(r, r, m);
    if (e.bit(i)) r = mulmod(r, b, m);
  }
  return r;
}

inline BigU gcd(BigU a, BigU b) {
  while (!b.is_zero()) { BigU t = a % b; a = b; b = t; }
  return a;
}

// inverse of a modulo m (m > 1); returns false if gcd(a, m) != 1
inline bool modinv(const BigU& a_in, const BigU& m, BigU& out) {
  // extended Euclid with coefficients tracked as (magnitude, sign)
  BigU r0 = m, r1 = a_in % m;
  BigU t0, t1(1);
  bool n0 = false, n1 = false;
  while (!r1.is_zero()) {
    BigU q, r2;
    divmod(r0, r1, q, r2);
    // t2 = t0 - q*t1
    BigU qt = q * t1;
    BigU t2;
    bool n2;
    if (n0 == n1) {            // same sign: t0 - q t1 may flip
      if (cmp(t0, qt) >= 0) { t2 = t0 - qt; n2 = n0; }
      else { t2 = qt - t0; n2 = !n0; }
    } else {                   // opposite signs: magnitudes add, sign of t0
      t2 = t0 + qt; n2 = n0;
    }
    r0 = r1; r1 = r2;
    t0 = t1; n0 = n1;
    t1 = t2; n1 = n2;
  }
  if (!(r0 == BigU(1))) return false;
  BigU t = t0 % m;
  out = (n0 && !t.is_zero()) ? (m - t) : t;
  return true;
}

inline BigU isqrt(const BigU& n) {
  if (n.is_zero()) return n;
  BigU x = shl(BigU(1), (n.bit_length() + 1) / 2);
  for (;;) {
    BigU y = shr(x + n / x, 1);
    if (cmp(y, x) >= 0) return x;
    x = y;
  }
}

}  // namespace hostbig
