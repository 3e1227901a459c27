// Host-side code of the product (circuit builder, plonky3 verifier-circuit restatement, native plonky3
// prover, AIR programs, circuit blob I/O) and the CPU oracle (witness, prover, verifier) in one
// executable for AddressSanitizer + UBSan (GPU sanitizers are not available on the pool; the device
// code is covered by tools/asmcheck and the parity tests).  Built and run by tests/test_sanitizers_cpu.py.
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "builder.h"
#include "circuit_io.h"
#include "circuit_bytes.h"
#include "ref_fft.h"
#include "p3_circuit.h"
#include "p3_prover.h"
#include "witness_program.h"
#include "recursion.h"
typedef uint64_t u64;
extern "C" {
void* p25o_circuit_load(const unsigned char* blob, size_t len);
void p25o_circuit_free(void* h);
int p25o_witness(void* h, const u64* inputs, u64 seed, u64* wires_out, char* msg, size_t msglen);
long p25o_check_constraints(void* h, const u64* wires, char* msg, size_t msglen);
void p25o_circuit_digest(void* h, u64* digest4, u64* cs_cap);
size_t p25o_proof_words(void* h);
int p25o_prove(void* h, const u64* inputs, u64 seed, u64* proof_out, double* timings_out, char* msg, size_t msglen);
int p25o_verify(void* h, const u64* digest4, const u64* cs_cap, const u64* proof_words, char* msg, size_t msglen);
void p25o_set_threads(int n);
int p25o_set_tuned(int on);
}
using namespace p25;
#define CHECK(c) do { if (!(c)) { printf("CHECK FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
  // product host code
  P3ProveParams prm;
  prm.log_n = 5; prm.num_queries = 6; prm.pow_bits = 6; prm.threads = 2;
  P3Config cfg;
  std::vector<u64> inp = p3_prove_fibonacci(prm, cfg);
  std::string js = p3_inputs_to_json(inp, cfg);
  CHECK(js.size() > inp.size());
  CircuitBuilder cb;
  FibonacciAir fib;
  p3_verify_proof(cb, cfg, fib);
  Circuit c = cb.build();
  std::vector<uint8_t> blob = circuit_to_blob(c);
  Circuit c2 = circuit_from_blob(blob.data(), blob.size());
  CHECK(c2.degree() == c.degree());
  CircuitBuilder cb2;
  ProgramAir pa(AirProgram::fibonacci());
  p3_verify_proof(cb2, cfg, pa);
  CHECK(circuit_to_blob(cb2.build()) == blob);
  for (int k = 0; k < 7; k++) CHECK(build_gadget_circuit(k, 5).degree() >= 4);
  // truncated / corrupted blobs must be rejected, not crash
  for (size_t cut : {(size_t)0, (size_t)7, blob.size() / 2, blob.size() - 1}) {
    bool threw = false;
    try { circuit_from_blob(blob.data(), cut); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
  }
  // bit-flip fuzz of a valid blob (a circuit handed over by another process is untrusted input): every
  // mutant is either rejected with an exception or yields a circuit whose witness program can be built
  // without touching memory out of bounds (ASan is the judge).
  {
    CircuitBuilder cbf;
    P3ProveParams ps;
    ps.log_n = 3; ps.num_queries = 2; ps.pow_bits = 4; ps.threads = 1;
    P3Config cf;
    p3_prove_fibonacci(ps, cf);
    p3_verify_proof(cbf, cf, fib);
    std::vector<uint8_t> small = circuit_to_blob(cbf.build());
    const size_t header_bytes = 8 * (1 + 32 + 4 * 16 + 8);
    u64 rng = 0x9E3779B97F4A7C15ull;
    auto next = [&] { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    int rejected = 0, accepted = 0;
    for (int trial = 0; trial < 400; trial++) {
      std::vector<uint8_t> m(small);
      const size_t span = trial % 2 ? header_bytes : m.size();
      const int flips = 1 + (int)(next() % 3);
      for (int f = 0; f < flips; f++) m[next() % span] ^= (uint8_t)(1u << (next() % 8));
      try {
        Circuit cm = circuit_from_blob(m.data(), m.size());
        WitnessProgram wpm = build_witness_program(cm);
        CHECK(wpm.input_slots.size() == cm.input_targets.size());
        accepted++;
      } catch (const std::exception&) {
        rejected++;
      }
    }
    printf("blob fuzz: %d rejected, %d accepted\n", rejected, accepted);
    CHECK(rejected > 50);
  }
  // upstream's binary circuit form (circuit_bytes.cpp): write -> read gives the circuit back; truncation and bit
  // flips are rejected or parse to something the witness scheduler can walk (the commitment part -- LDE leaves,
  // Merkle digests -- is opaque to the reader, so placeholders of the right size stand in for the device's tables)
  {
    CircuitBuilder cbb;
    P3ProveParams ps;
    ps.log_n = 3; ps.num_queries = 2; ps.pow_bits = 4; ps.threads = 1;
    P3Config cf;
    p3_prove_fibonacci(ps, cf);
    p3_verify_proof(cbb, cf, fib);
    Circuit small = cbb.build();
    const size_t n = small.degree(), big = n << small.cfg.rate_bits, ncs = small.constants_sigmas.size();
    std::vector<u64> coeffs(ncs * n), lde(ncs * big, 1), tree(8 * big, 2);
    for (size_t p = 0; p < ncs; p++) {
      std::vector<u64> v = small.constants_sigmas[p];
      ref_ifft(v);
      std::copy(v.begin(), v.end(), coeffs.begin() + p * n);
    }
    CircuitCommitment cm;
    cm.coeffs = coeffs.data(); cm.lde = lde.data(); cm.tree = tree.data();
    for (int i = 0; i < 4; i++) cm.digest[i] = 1000 + i;
    std::vector<uint8_t> bytes = circuit_data_to_bytes(small, cm);
    std::vector<uint32_t> in_idx;
    for (auto& t : small.input_targets) in_idx.push_back((uint32_t)small.target_index(t));
    u64 dg[4];
    Circuit back = circuit_data_from_bytes(bytes.data(), bytes.size(), in_idx.data(), in_idx.size(), dg);
    CHECK(circuit_to_blob(back) == circuit_to_blob(small) && dg[3] == 1003);
    u64 rng = 0x243F6A8885A308D3ull;
    auto next = [&] { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    int rejected = 0, accepted = 0;
    for (int trial = 0; trial < 60; trial++) {
      std::vector<uint8_t> m(bytes);
      if (trial % 3 == 0) {
        m.resize(next() % m.size());
      } else {
        const size_t span = trial % 3 == 1 ? 4096 : m.size();
        for (int f = 0; f < 1 + (int)(next() % 3); f++) m[next() % span] ^= (uint8_t)(1u << (next() % 8));
      }
      try {
        Circuit cm2 = circuit_data_from_bytes(m.data(), m.size(), in_idx.data(), in_idx.size(), dg);
        WitnessProgram wpm = build_witness_program(cm2);
        CHECK(wpm.input_slots.size() == cm2.input_targets.size());
        accepted++;
      } catch (const std::exception&) {
        rejected++;
      }
    }
    printf("CircuitData bytes fuzz: %d rejected, %d accepted\n", rejected, accepted);
    CHECK(rejected > 20);
  }
  // round 6: FriConfig.log_blowup 2 with a degree-4 AIR (four quotient chunks): y = x^4 + 3 x + 5 on every row, next x = y + 7 x
  {
    AirProgram q;
    q.width = 2;
    auto nd = [&](uint32_t op, uint32_t a, uint32_t b, u64 v) { q.nodes.push_back(AirProgram::Node{op, a, b, v}); return (uint32_t)q.nodes.size() - 1; };
    const uint32_t x = nd(AirProgram::LOCAL, 0, 0, 0), y = nd(AirProgram::LOCAL, 1, 0, 0), nx = nd(AirProgram::NEXT, 0, 0, 0);
    const uint32_t x2 = nd(AirProgram::MUL, x, x, 0), x4 = nd(AirProgram::MUL, x2, x2, 0);
    const uint32_t c3 = nd(AirProgram::CONST, 0, 0, 3), c5 = nd(AirProgram::CONST, 0, 0, 5), c7 = nd(AirProgram::CONST, 0, 0, 7), c2 = nd(AirProgram::CONST, 0, 0, 2);
    const uint32_t rhs = nd(AirProgram::ADD, nd(AirProgram::ADD, x4, nd(AirProgram::MUL, c3, x, 0), 0), c5, 0);
    q.constraints.push_back({nd(AirProgram::SUB, rhs, y, 0), AirProgram::ALWAYS});
    q.constraints.push_back({nd(AirProgram::SUB, x, c2, 0), AirProgram::FIRST_ROW});
    q.constraints.push_back({nd(AirProgram::SUB, nx, nd(AirProgram::ADD, y, nd(AirProgram::MUL, c7, x, 0), 0), 0), AirProgram::TRANSITION});
    q.validate();
    CHECK(q.max_constraint_degree() == 4 && q.log_quotient_degree() == 2);
    const size_t n = 16;
    std::vector<std::vector<u64>> col(2, std::vector<u64>(n));
    u64 xv = 2;
    for (size_t i = 0; i < n; i++) {
      const u64 x2v = gl::mul(xv, xv), yv = gl::add(gl::add(gl::mul(x2v, x2v), gl::mul(3, xv)), 5);
      col[0][i] = xv;
      col[1][i] = yv;
      xv = gl::add(yv, gl::mul(7, xv));
    }
    P3ProveParams p2;
    p2.log_n = 4; p2.log_blowup = 2; p2.num_queries = 4; p2.pow_bits = 4; p2.threads = 2;
    P3Config c2cfg;
    std::vector<u64> in2 = p3_prove_air(q, col, p2, c2cfg);
    CHECK(c2cfg.fri_config.log_blowup == 2 && c2cfg.log_quotient_degree == 2 && c2cfg.opening_matrix_log_max_height == 6);
    CHECK(in2.size() == c2cfg.num_inputs());
    std::string j2 = p3_inputs_to_json(in2, c2cfg);
    CHECK(j2.size() > in2.size());
    CircuitBuilder cbq;
    ProgramAir paq(q);
    p3_verify_proof(cbq, c2cfg, paq);
    std::vector<uint8_t> bq = circuit_to_blob(cbq.build());
    void* hq = p25o_circuit_load(bq.data(), bq.size());
    CHECK(hq != nullptr);
    char m2[256] = {0};
    std::vector<u64> wq((size_t)circuit_from_blob(bq.data(), bq.size()).degree() * 135);
    CHECK(p25o_witness(hq, in2.data(), 1, wq.data(), m2, sizeof m2) == 0);
    CHECK(p25o_check_constraints(hq, wq.data(), m2, sizeof m2) == 0);
    in2[8 + 4 * 2 + 5] ^= 1;      // one of the four chunks' openings
    CHECK(p25o_witness(hq, in2.data(), 1, wq.data(), m2, sizeof m2) != 0);
    bool threw = false;
    p2.log_blowup = 1;              // four chunks do not fit the LDE domain of log_blowup 1
    try { p3_prove_air(q, col, p2, c2cfg); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
    p25o_circuit_free(hq);
  }
  // oracle on a small verifier circuit
  prm.log_n = 3; prm.num_queries = 3; prm.pow_bits = 4;
  inp = p3_prove_fibonacci(prm, cfg);
  CircuitBuilder cb3;
  p3_verify_proof(cb3, cfg, fib);
  Circuit c3 = cb3.build();
  blob = circuit_to_blob(c3);
  p25o_set_threads(4);
  void* h = p25o_circuit_load(blob.data(), blob.size());
  CHECK(h != nullptr);
  char msg[256] = {0};
  std::vector<u64> wires((size_t)c3.degree() * 135);
  CHECK(p25o_witness(h, inp.data(), 1, wires.data(), msg, sizeof msg) == 0);
  CHECK(p25o_check_constraints(h, wires.data(), msg, sizeof msg) == 0);
  u64 dg[4];
  std::vector<u64> cap(16 * 4);
  p25o_circuit_digest(h, dg, cap.data());
  std::vector<u64> proof(p25o_proof_words(h));
  double tm[16];
  CHECK(p25o_prove(h, inp.data(), 1, proof.data(), tm, msg, sizeof msg) == 0);
  CHECK(p25o_verify(h, dg, cap.data(), proof.data(), msg, sizeof msg) == 0);
  if (p25o_set_tuned(1)) {   // the AVX-512 leg of the cpu_baseline (ref_hash_x8.cpp, ref_quotient_x8.cpp): same bytes
    std::vector<u64> tuned(proof.size());
    CHECK(p25o_prove(h, inp.data(), 1, tuned.data(), tm, msg, sizeof msg) == 0);
    CHECK(tuned == proof);
    p25o_set_tuned(0);
  }
  proof[100] ^= 1;
  CHECK(p25o_verify(h, dg, cap.data(), proof.data(), msg, sizeof msg) != 0);
  std::vector<u64> bad(inp);
  bad[9] ^= 1;
  CHECK(p25o_witness(h, bad.data(), 1, wires.data(), msg, sizeof msg) != 0);
  // recursion: gate-level evaluator circuits and the recursive verifier of the small circuit above, witness
  // generated from the oracle's proof
  for (int k : {3, 9, 10, 11}) CHECK(build_gate_eval_circuit((GateKind)k).degree() >= 4);
  {
    proof[100] ^= 1;  // undo the tampering above
    std::vector<u64> capv(cap);
    Circuit outer = build_recursive_verifier(c3, dg, capv, 1);
    std::vector<uint8_t> ob = circuit_to_blob(outer);
    void* ho = p25o_circuit_load(ob.data(), ob.size());
    CHECK(ho != nullptr);
    std::vector<u64> ow((size_t)outer.degree() * 135);
    CHECK(p25o_witness(ho, proof.data(), 2, ow.data(), msg, sizeof msg) == 0);
    CHECK(p25o_check_constraints(ho, ow.data(), msg, sizeof msg) == 0);
    proof[200] ^= 1;
    CHECK(p25o_witness(ho, proof.data(), 2, ow.data(), msg, sizeof msg) != 0);
    p25o_circuit_free(ho);
  }
  p25o_circuit_free(h);
  printf("SANITIZE OK\n");
  return 0;
}
