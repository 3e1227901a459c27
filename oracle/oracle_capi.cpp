// ORACLE -- test infrastructure only.  C entry points (ctypes) over the CPU restatement, used by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product.
#include <string.h>
#include "ref_fft.h"
#include "ref_hash.h"

extern "C" {

void p25o_poseidon_permute(u64* states, size_t n) {
  for (size_t i = 0; i < n; i++) ref_poseidon(states + 12 * i);
}
void p25o_poseidon2_permute(u64* states, size_t n) {
  for (size_t i = 0; i < n; i++) ref_poseidon2(states + 12 * i);
}
void p25o_poseidon2_trace(u64* state, u64* trace) { ref_poseidon2_trace(state, trace); }
void p25o_hash_no_pad(const u64* in, size_t n, u64* out4) {
  RHash h = ref_hash_no_pad(in, n);
  memcpy(out4, h.e, 32);
}
u64 p25o_mul(u64 a, u64 b) { return rf_mul(a, b); }
u64 p25o_inv(u64 a) { return rf_inv(a); }

// leaves row-major [n_leaves][width] (upstream orientation).  tree_out nullable:
// all levels, leaf digests first, 4 words per node.
void p25o_merkle_commit(const u64* leaves_rm, size_t n_leaves, size_t width, unsigned cap_height,
                        u64* cap_out, u64* tree_out) {
  std::vector<std::vector<u64>> leaves(n_leaves);
  for (size_t i = 0; i < n_leaves; i++) leaves[i].assign(leaves_rm + i * width, leaves_rm + (i + 1) * width);
  RMerkleTree t = ref_merkle_build(leaves, cap_height);
  memcpy(cap_out, t.cap().data(), t.cap().size() * 32);
  if (tree_out)
    for (auto& lv : t.levels) {
      memcpy(tree_out, lv.data(), lv.size() * 32);
      tree_out += lv.size() * 4;
    }
}

// PolynomialBatch::from_values / from_coeffs.  polys[n_polys][n].
// coeffs_out[n_polys][n]; lde_out[n_polys][n << rate_bits] in BIT-REVERSED index order (the order
// upstream's leaves are in after transpose + reverse_index_bits); cap_out[2^cap][4].
void p25o_lde_commit(const u64* polys, unsigned log_n, size_t n_polys, int from_coeffs,
                     unsigned rate_bits, unsigned cap_height, u64* coeffs_out, u64* lde_out,
                     u64* cap_out) {
  const size_t n = (size_t)1 << log_n, big = n << rate_bits;
  std::vector<std::vector<u64>> leaves(big, std::vector<u64>(n_polys));
  for (size_t p = 0; p < n_polys; p++) {
    std::vector<u64> c(polys + p * n, polys + (p + 1) * n);
    if (!from_coeffs) ref_ifft(c);
    if (coeffs_out) memcpy(coeffs_out + p * n, c.data(), n * 8);
    std::vector<u64> v = ref_lde_values(c, rate_bits, 7);
    for (size_t i = 0; i < big; i++) {
      size_t r = rbits(i, log_n + rate_bits);
      leaves[r][p] = v[i];
      if (lde_out) lde_out[p * big + r] = v[i];
    }
  }
  if (cap_out) {
    RMerkleTree t = ref_merkle_build(leaves, cap_height);
    memcpy(cap_out, t.cap().data(), t.cap().size() * 32);
  }
}

}  // extern "C"
