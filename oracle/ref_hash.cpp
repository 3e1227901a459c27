// ORACLE -- test infrastructure only (see ref_hash.h for what each function restates).
#include "ref_hash.h"
#include <string.h>

static const u64 POSEIDON_RC[360] = {
#include "poseidon_constants.inc"
};
static const u64 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const u64 MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#include "poseidon2_constants.inc"

static u64 pow7(u64 x) {
  u64 x2 = rf_mul(x, x), x4 = rf_mul(x2, x2), x3 = rf_mul(x, x2);
  return rf_mul(x3, x4);
}

static void poseidon_mds(u64 s[12]) {
  // out[r] = sum_i s[(i+r)%12]*CIRC[i] + s[r]*DIAG[r].  The 32-bit halves are accumulated separately
  // in 64-bit integers (entries <= 41, 13 terms: no overflow) -- same value as the 128-bit sum, but
  // without 128-bit multiplies, so the CPU baseline is not needlessly slow.
  u64 lo[24], hi[24], o[12];
  for (int i = 0; i < 12; i++) {
    lo[i] = lo[i + 12] = s[i] & 0xFFFFFFFFull;
    hi[i] = hi[i + 12] = s[i] >> 32;
  }
  for (int r = 0; r < 12; r++) {
    u64 al = lo[r] * MDS_DIAG[r], ah = hi[r] * MDS_DIAG[r];
    for (int i = 0; i < 12; i++) {
      al += lo[i + r] * MDS_CIRC[i];
      ah += hi[i + r] * MDS_CIRC[i];
    }
    o[r] = rf_reduce((u128)al + ((u128)ah << 32));
  }
  memcpy(s, o, sizeof(o));
}

// Same permutation with the 22 partial rounds in the "optimised" form of the Poseidon paper (what
// upstream's CPU code does: partial_first_constant_layer, mds_partial_layer_init, then per round one
// S-box, one scalar constant and a sparse matrix).  Constants derived from the naive definition and
// verified against it by tools/gen_poseidon_fast.py; ref_poseidon_naive below is the definition.
namespace pf {
#include "poseidon_fast_constants.inc"
}
static inline u64 red_acc(u128 acc, u32 ov) {
  // acc + ov * 2^128, with 2^128 = -2^32 (mod p)
  return rf_sub(rf_reduce(acc), rf_mul(ov, (u64)1 << 32));
}
// partial_in (nullable): receives lane 0's S-box input of each of the 22 partial rounds in THIS (fast) formulation --
// upstream's PoseidonGate stores exactly these as wires; tests check they equal the naive form's (ref_poseidon_trace).
static void poseidon_fast(u64 s[12], u64* partial_in) {
  int r = 0;
  for (; r < 4; r++) {
    for (int i = 0; i < 12; i++) s[i] = pow7(rf_add(s[i], POSEIDON_RC[12 * r + i]));
    poseidon_mds(s);
  }
  for (int i = 0; i < 12; i++) s[i] = rf_add(s[i], pf::PF_FIRST[i]);
  {
    u64 t[11];
    for (int c = 0; c < 11; c++) {
      u128 acc = 0;
      u32 ov = 0;
      for (int k = 0; k < 11; k++) {
        u128 p = (u128)s[k + 1] * pf::PF_INIT[k * 11 + c];
        acc += p;
        ov += acc < p;
      }
      t[c] = red_acc(acc, ov);
    }
    memcpy(s + 1, t, sizeof(t));
  }
  for (int i = 0; i < 22; i++) {
    if (partial_in) partial_in[i] = s[0];
    u64 s0 = rf_add(pow7(s[0]), pf::PF_SCALAR[i]);
    u128 acc = (u128)s0 * 25;  // m00 = MDS_CIRC[0] + MDS_DIAG[0]
    u32 ov = 0;
    for (int j = 0; j < 11; j++) {
      u128 p = (u128)s[j + 1] * pf::PF_VHAT[i * 11 + j];
      acc += p;
      ov += acc < p;
    }
    for (int j = 0; j < 11; j++) s[j + 1] = rf_reduce((u128)s0 * pf::PF_W[i * 11 + j] + s[j + 1]);
    s[0] = red_acc(acc, ov);
  }
  r += 22;
  for (; r < 30; r++) {
    for (int i = 0; i < 12; i++) s[i] = pow7(rf_add(s[i], POSEIDON_RC[12 * r + i]));
    poseidon_mds(s);
  }
}

void ref_poseidon(u64 s[12]) { poseidon_fast(s, nullptr); }
void ref_poseidon_fast_partial_inputs(u64 s[12], u64 partial_in[22]) { poseidon_fast(s, partial_in); }

void ref_poseidon_naive(u64 s[12]) {
  for (int r = 0; r < 30; r++) {
    for (int i = 0; i < 12; i++) s[i] = rf_add(s[i], POSEIDON_RC[12 * r + i]);
    if (r < 4 || r >= 26) {
      for (int i = 0; i < 12; i++) s[i] = pow7(s[i]);
    } else {
      s[0] = pow7(s[0]);
    }
    poseidon_mds(s);
  }
}

void ref_poseidon_trace(u64 s[12], u64* trace) {
  int k = 0;
  for (int r = 0; r < 30; r++) {
    for (int i = 0; i < 12; i++) s[i] = rf_add(s[i], POSEIDON_RC[12 * r + i]);
    if (r < 4 || r >= 26) {
      if (r != 0 && trace)
        for (int i = 0; i < 12; i++) trace[k++] = s[i];
      for (int i = 0; i < 12; i++) s[i] = pow7(s[i]);
    } else {
      if (trace) trace[k++] = s[0];
      s[0] = pow7(s[0]);
    }
    poseidon_mds(s);
  }
}

// poseidon2.rs:184-213 (matmul_m4) + :126-147 (matmul_external)
static void p2_external(u64 s[12]) {
  for (int b = 0; b < 3; b++) {
    u64* x = s + 4 * b;
    u64 t0 = rf_add(x[0], x[1]);
    u64 t1 = rf_add(x[2], x[3]);
    u64 t2 = rf_add(t1, rf_mul(2, x[1]));
    u64 t3 = rf_add(t0, rf_mul(2, x[3]));
    u64 t4 = rf_add(t3, rf_mul(4, t1));
    u64 t5 = rf_add(t2, rf_mul(4, t0));
    x[0] = rf_add(t3, t5);
    x[1] = t5;
    x[2] = rf_add(t2, t4);
    x[3] = t4;
  }
  u64 st[4];
  for (int l = 0; l < 4; l++) st[l] = rf_add(rf_add(s[l], s[4 + l]), s[8 + l]);
  for (int i = 0; i < 12; i++) s[i] = rf_add(s[i], st[i % 4]);
}
// poseidon2.rs:163-182
static void p2_internal(u64 s[12]) {
  u64 sum = 0;
  for (int i = 0; i < 12; i++) sum = rf_add(sum, s[i]);
  for (int i = 0; i < 12; i++) s[i] = rf_add(rf_mul(s[i], P2_MAT_DIAG_M_1[i] - 1), sum);
}

void ref_poseidon2_trace(u64 s[12], u64* trace) {
  p2_external(s);
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 12; i++) s[i] = rf_add(s[i], P2_RC[12 * r + i]);
    if (trace && r) memcpy(trace + 12 * (r - 1), s, 96);
    for (int i = 0; i < 12; i++) s[i] = pow7(s[i]);
    p2_external(s);
  }
  for (int r = 0; r < 22; r++) {
    s[0] = rf_add(s[0], P2_RC_MID[r]);
    if (trace) trace[36 + r] = s[0];
    s[0] = pow7(s[0]);
    p2_internal(s);
  }
  for (int r = 4; r < 8; r++) {
    for (int i = 0; i < 12; i++) s[i] = rf_add(s[i], P2_RC[12 * r + i]);
    if (trace) memcpy(trace + 58 + 12 * (r - 4), s, 96);
    for (int i = 0; i < 12; i++) s[i] = pow7(s[i]);
    p2_external(s);
  }
}
void ref_poseidon2(u64 s[12]) { ref_poseidon2_trace(s, nullptr); }

RHash ref_hash_no_pad(const u64* in, size_t n) {
  u64 s[12] = {0};
  for (size_t off = 0; off < n; off += 8) {
    size_t m = n - off < 8 ? n - off : 8;
    for (size_t i = 0; i < m; i++) s[i] = in[off + i];
    ref_poseidon(s);
  }
  RHash h;
  memcpy(h.e, s, 32);
  return h;
}
RHash ref_hash_or_noop(const u64* in, size_t n) {
  if (n <= 4) {
    RHash h = {{0, 0, 0, 0}};
    for (size_t i = 0; i < n; i++) h.e[i] = in[i];
    return h;
  }
  return ref_hash_no_pad(in, n);
}
RHash ref_two_to_one(const RHash& l, const RHash& r) {
  u64 s[12] = {0};
  memcpy(s, l.e, 32);
  memcpy(s + 4, r.e, 32);
  ref_poseidon(s);
  RHash h;
  memcpy(h.e, s, 32);
  return h;
}

RMerkleTree ref_merkle_build(const std::vector<std::vector<u64>>& leaves, unsigned cap_height) {
  RMerkleTree t;
  t.cap_height = cap_height;
  std::vector<RHash> cur(leaves.size());
  for (size_t i = 0; i < leaves.size(); i++) cur[i] = ref_hash_or_noop(leaves[i].data(), leaves[i].size());
  t.levels.push_back(cur);
  while (cur.size() > ((size_t)1 << cap_height)) {
    std::vector<RHash> nxt(cur.size() / 2);
    for (size_t i = 0; i < nxt.size(); i++) nxt[i] = ref_two_to_one(cur[2 * i], cur[2 * i + 1]);
    t.levels.push_back(nxt);
    cur.swap(nxt);
  }
  return t;
}
std::vector<RHash> RMerkleTree::prove(size_t leaf) const {
  std::vector<RHash> sib;
  for (size_t k = 0; k + 1 < levels.size(); k++) sib.push_back(levels[k][(leaf >> k) ^ 1]);
  return sib;
}
bool ref_merkle_verify(const std::vector<u64>& leaf, size_t index, const std::vector<RHash>& cap,
                       const std::vector<RHash>& siblings) {
  RHash d = ref_hash_or_noop(leaf.data(), leaf.size());
  for (const RHash& s : siblings) {
    d = (index & 1) ? ref_two_to_one(s, d) : ref_two_to_one(d, s);
    index >>= 1;
  }
  if (index >= cap.size()) return false;
  return memcmp(d.e, cap[index].e, 32) == 0;
}

RChallenger::RChallenger() { memset(state, 0, sizeof(state)); }
void RChallenger::duplex() {
  for (size_t i = 0; i < in.size(); i++) state[i] = in[i];
  in.clear();
  ref_poseidon(state);
  out.assign(state, state + 8);
}
void RChallenger::observe(u64 x) {
  out.clear();
  in.push_back(x);
  if (in.size() == 8) duplex();
}
void RChallenger::observe_hash(const RHash& h) {
  for (int i = 0; i < 4; i++) observe(h.e[i]);
}
void RChallenger::observe_cap(const std::vector<RHash>& cap) {
  for (const RHash& h : cap) observe_hash(h);
}
void RChallenger::observe_ext(RE2 x) {
  observe(x.a);
  observe(x.b);
}
u64 RChallenger::challenge() {
  if (!in.empty() || out.empty()) duplex();
  u64 v = out.back();
  out.pop_back();
  return v;
}
RE2 RChallenger::ext_challenge() {
  u64 a = challenge();
  u64 b = challenge();
  return RE2{a, b};
}
