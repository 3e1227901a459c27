// ORACLE -- test infrastructure only (see ref_fft.h).
#include "ref_fft.h"
#include <utility>
#include "ref_field_x8.h"   // AVX-512 butterflies of the tuned cpu_baseline leg (functions with target attributes)

// The tuned leg (bench.py cpu_baseline, ref_set_tuned): stages whose half-length is >= 8 run eight butterflies per pass on
// AVX-512 -- same butterflies, same canonical values, same order of results; the checker leaves it off.
static bool g_fft_tuned = false;
void ref_fft_set_tuned(bool on) { g_fft_tuned = on; }

// one decimation-in-time stage (len >= 8): u = d[s+j], v = d[s+j+len] * w[j]; d[s+j] = u + v, d[s+j+len] = u - v
__attribute__((target("avx512f,avx512dq"))) static void dit_stage_x8(u64* d, size_t n, size_t len, const u64* w) {
  for (size_t s = 0; s < n; s += 2 * len)
    for (size_t j = 0; j < len; j += 8) {
      V u = _mm512_loadu_si512(d + s + j);
      V v = v_mul(_mm512_loadu_si512(d + s + j + len), _mm512_loadu_si512(w + j));
      _mm512_storeu_si512(d + s + j, v_add(u, v));
      _mm512_storeu_si512(d + s + j + len, v_sub(u, v));
    }
}
// one decimation-in-frequency stage (len >= 8): d[s+j] = u + v, d[s+j+len] = (u - v) * w[j]
__attribute__((target("avx512f,avx512dq"))) static void dif_stage_x8(u64* d, size_t n, size_t len, const u64* w) {
  for (size_t s = 0; s < n; s += 2 * len)
    for (size_t j = 0; j < len; j += 8) {
      V u = _mm512_loadu_si512(d + s + j), v = _mm512_loadu_si512(d + s + j + len);
      _mm512_storeu_si512(d + s + j, v_add(u, v));
      _mm512_storeu_si512(d + s + j + len, v_mul(v_sub(u, v), _mm512_loadu_si512(w + j)));
    }
}

static unsigned log2_exact(size_t n) {
  unsigned l = 0;
  while (((size_t)1 << l) < n) l++;
  return l;
}

// Twiddles in per-stage contiguous layout, tw[len + j] = root_{2 len}^j (j < len), cached per root so
// the timed CPU baseline does not rebuild them (or stride through one big table) on every transform.
#include <map>
#include <memory>
#include <mutex>
static const std::vector<u64>& stage_twiddles(size_t n, u64 root) {
  static std::mutex m;
  static std::map<std::pair<size_t, u64>, std::unique_ptr<std::vector<u64>>> cache;
  std::lock_guard<std::mutex> lk(m);
  auto& slot = cache[{n, root}];
  if (!slot) {
    slot.reset(new std::vector<u64>(n < 2 ? 2 : n));
    std::vector<u64>& tw = *slot;
    // top stage: len = n/2, root itself; lower stages take every other entry
    for (size_t len = n / 2; len >= 1; len >>= 1) {
      if (len == n / 2) {
        u64 w = 1;
        for (size_t j = 0; j < len; j++) {
          tw[len + j] = w;
          w = rf_mul(w, root);
        }
      } else {
        for (size_t j = 0; j < len; j++) tw[len + j] = tw[2 * len + 2 * j];
      }
    }
  }
  return *slot;
}

static void fft_core(std::vector<u64>& a, u64 root) {
  const size_t n = a.size();
  if (n < 2) return;
  const unsigned lg = log2_exact(n);
  for (size_t i = 0; i < n; i++) {
    size_t j = rbits(i, lg);
    if (i < j) std::swap(a[i], a[j]);
  }
  const std::vector<u64>& tw = stage_twiddles(n, root);
  u64* d = a.data();
  for (size_t s = 0; s < n; s += 2) {  // len = 1: twiddle 1
    u64 u = d[s], v = d[s + 1];
    d[s] = rf_add(u, v);
    d[s + 1] = rf_sub(u, v);
  }
  for (size_t len = 2; len < n; len <<= 1) {
    const u64* w = tw.data() + len;
    if (g_fft_tuned && len >= 8) {
      dit_stage_x8(d, n, len, w);
      continue;
    }
    for (size_t s = 0; s < n; s += 2 * len)
      for (size_t j = 0; j < len; j++) {
        u64 u = d[s + j], v = rf_mul(d[s + j + len], w[j]);
        d[s + j] = rf_add(u, v);
        d[s + j + len] = rf_sub(u, v);
      }
  }
}

// natural order in, BIT-REVERSED order out (decimation in frequency): out[rbits(i)] = sum_k a[k] root^(i k).  Used for
// the LDE, whose values are wanted in bit-reversed (Merkle leaf) order anyway: no permutation pass at all.
static void fft_dif_core(std::vector<u64>& a, u64 root) {
  const size_t n = a.size();
  if (n < 2) return;
  const std::vector<u64>& tw = stage_twiddles(n, root);
  u64* d = a.data();
  for (size_t len = n / 2; len >= 2; len >>= 1) {
    const u64* w = tw.data() + len;
    if (g_fft_tuned && len >= 8) {
      dif_stage_x8(d, n, len, w);
      continue;
    }
    for (size_t s = 0; s < n; s += 2 * len)
      for (size_t j = 0; j < len; j++) {
        u64 u = d[s + j], v = d[s + j + len];
        d[s + j] = rf_add(u, v);
        d[s + j + len] = rf_mul(rf_sub(u, v), w[j]);
      }
  }
  for (size_t s = 0; s < n; s += 2) {
    u64 u = d[s], v = d[s + 1];
    d[s] = rf_add(u, v);
    d[s + 1] = rf_sub(u, v);
  }
}
std::vector<u64> ref_lde_values_bitrev(const std::vector<u64>& coeffs, unsigned rate_bits, u64 shift) {
  std::vector<u64> v(coeffs);
  v.resize(coeffs.size() << rate_bits, 0);
  u64 p = 1;
  for (size_t k = 0; k < coeffs.size(); k++) {
    v[k] = rf_mul(v[k], p);
    p = rf_mul(p, shift);
  }
  fft_dif_core(v, rf_root_of_unity(log2_exact(v.size())));
  return v;
}

void ref_fft(std::vector<u64>& a) { fft_core(a, rf_root_of_unity(log2_exact(a.size()))); }
void ref_ifft(std::vector<u64>& a) {
  fft_core(a, rf_inv(rf_root_of_unity(log2_exact(a.size()))));
  u64 ninv = rf_inv((u64)a.size());
  for (auto& x : a) x = rf_mul(x, ninv);
}
void ref_coset_fft(std::vector<u64>& a, u64 shift) {
  u64 p = 1;
  for (auto& x : a) {
    x = rf_mul(x, p);
    p = rf_mul(p, shift);
  }
  ref_fft(a);
}
void ref_coset_ifft(std::vector<u64>& a, u64 shift) {
  ref_ifft(a);
  u64 si = rf_inv(shift), p = 1;
  for (auto& x : a) {
    x = rf_mul(x, p);
    p = rf_mul(p, si);
  }
}
void ref_fft_ext(std::vector<RE2>& a) {
  std::vector<u64> x(a.size()), y(a.size());
  for (size_t i = 0; i < a.size(); i++) x[i] = a[i].a, y[i] = a[i].b;
  ref_fft(x);
  ref_fft(y);
  for (size_t i = 0; i < a.size(); i++) a[i] = RE2{x[i], y[i]};
}
void ref_coset_fft_ext(std::vector<RE2>& a, u64 shift) {
  std::vector<u64> x(a.size()), y(a.size());
  for (size_t i = 0; i < a.size(); i++) x[i] = a[i].a, y[i] = a[i].b;
  ref_coset_fft(x, shift);
  ref_coset_fft(y, shift);
  for (size_t i = 0; i < a.size(); i++) a[i] = RE2{x[i], y[i]};
}
std::vector<u64> ref_lde_values(const std::vector<u64>& coeffs, unsigned rate_bits, u64 shift) {
  std::vector<u64> v(coeffs);
  v.resize(coeffs.size() << rate_bits, 0);
  ref_coset_fft(v, shift);
  return v;
}
