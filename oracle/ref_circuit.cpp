// ORACLE -- test infrastructure only (see ref_circuit.h).
#include "ref_circuit.h"
#include <stdexcept>
#include <string.h>
#include "ref_hash.h"

namespace {
struct Rd {
  const unsigned char* p;
  size_t len, off;
  u64 w() {
    if (off + 8 > len) throw std::runtime_error("blob truncated");
    u64 v;
    memcpy(&v, p + off, 8);
    off += 8;
    return v;
  }
  void a32(u32* out, size_t n) {
    size_t bytes = (n * 4 + 7) & ~(size_t)7;
    if (off + bytes > len) throw std::runtime_error("blob truncated");
    if (n) memcpy(out, p + off, n * 4);
    off += bytes;
  }
  void a64(u64* out, size_t n) {
    if (off + 8 * n > len) throw std::runtime_error("blob truncated");
    if (n) memcpy(out, p + off, 8 * n);
    off += 8 * n;
  }
};
}  // namespace

RCircuit ref_circuit_parse(const unsigned char* blob, size_t len) {
  Rd r{blob, len, 0};
  if (r.w() != 0x3143524943353250ull) throw std::runtime_error("bad magic");
  u64 h[32];
  r.a64(h, 32);
  RCircuit c;
  c.degree_bits = (int)h[0]; c.num_wires = (int)h[1]; c.num_routed = (int)h[2]; c.num_constants = (int)h[3];
  c.num_challenges = (int)h[4]; c.quotient_degree_factor = (int)h[5]; c.rate_bits = (int)h[6];
  c.cap_height = (int)h[7]; c.pow_bits = (int)h[8]; c.num_queries = (int)h[9];
  size_t n_arity = h[10];
  c.num_selectors = (int)h[11]; c.num_gate_constraints = (int)h[12]; c.num_partial_products = (int)h[13];
  size_t ng = h[14];
  c.pi_row = (int)h[15]; c.num_virtual = h[16]; c.num_inputs = h[17];
  size_t n_gen = h[18], n_cs = h[19];
  for (size_t i = 0; i < ng; i++) {
    RGateType g;
    g.kind = (u32)r.w(); g.selector_index = (int)r.w(); g.group_start = (int)r.w(); g.group_end = (int)r.w();
    c.gates.push_back(g);
  }
  for (size_t i = 0; i < n_arity; i++) c.arity_bits.push_back((int)r.w());
  const size_t n = c.n();
  c.row_kind.resize(n);
  r.a32(c.row_kind.data(), n);
  c.constants_sigmas.assign(n_cs, std::vector<u64>(n));
  for (auto& p : c.constants_sigmas) r.a64(p.data(), n);
  c.k_is.resize(c.num_routed);
  r.a64(c.k_is.data(), c.k_is.size());
  c.input_targets.resize(c.num_inputs);
  r.a32(c.input_targets.data(), c.num_inputs);
  c.rep.resize(c.num_targets());
  r.a32(c.rep.data(), c.rep.size());
  c.gens.resize(n_gen);
  for (auto& g : c.gens) {
    g.kind = (u32)r.w(); g.c0 = r.w(); g.c1 = r.w(); g.aux = (int)r.w();
    if (g.kind == RGEN_RANDOM) g.c1 = c.num_random_fill++;  // ordinal among the RandomValueGenerators
    g.n_deps = (u32)r.w(); g.n_outs = (u32)r.w();
    g.arg_off = c.gen_args.size();
    c.gen_args.resize(g.arg_off + g.n_deps + g.n_outs);
    r.a32(c.gen_args.data() + g.arg_off, g.n_deps + g.n_outs);
  }
  if (h[22] > 4096) throw std::runtime_error("too many public inputs");
  c.public_inputs.resize(h[22]);   // header word 22: registered public inputs, listed after the generator table
  r.a32(c.public_inputs.data(), c.public_inputs.size());
  for (u32 t : c.public_inputs)
    if (t >= c.num_targets()) throw std::runtime_error("public-input target out of range");
  return c;
}

// Deterministic stand-in for upstream's RandomValueGenerator (F::rand() from the OS RNG, which makes
// the real prover's PublicInputGate filler wires non-reproducible): SplitMix64 of (seed, k), reduced.
u64 ref_random_fill(u64 seed, u64 k) {
  u64 z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return z >= RP ? z - RP : z;
}

namespace {
struct PartitionWitness {
  const RCircuit& c;
  std::vector<u64> val;
  std::vector<unsigned char> set;
  bool conflict = false;
  std::string msg;
  explicit PartitionWitness(const RCircuit& cc) : c(cc), val(cc.num_targets(), 0), set(cc.num_targets(), 0) {}
  bool has(u32 t) const { return set[c.rep[t]]; }
  u64 get(u32 t) const { return val[c.rep[t]]; }
  // returns true if newly set
  bool put(u32 t, u64 v) {
    u32 r = c.rep[t];
    if (set[r]) {
      if (val[r] != v && !conflict) {
        conflict = true;
        msg = "Partition containing target " + std::to_string(t) + " was set twice with different values: " +
              std::to_string(val[r]) + " != " + std::to_string(v);
      }
      return false;
    }
    set[r] = 1;
    val[r] = v;
    return true;
  }
};

// Runs one generator; outputs appended to `out` in the generator's `outs` order.
void run_generator(const RCircuit& c, const RGenerator& g, const PartitionWitness& w, u64 seed, const u64* filler,
                   std::vector<u64>& out) {
  const u32* dep = c.gen_args.data() + g.arg_off;
  auto d = [&](int i) { return w.get(dep[i]); };
  out.clear();
  switch (g.kind) {
    case RGEN_CONSTANT:
      out.push_back(g.c0);
      break;
    case RGEN_RANDOM:
      // explicit filler (e.g. the values a real upstream run drew from the OS RNG), indexed by the generator's
      // ordinal among the RandomValueGenerators, or the deterministic SplitMix64 stand-in
      out.push_back(filler ? filler[g.c1] : ref_random_fill(seed, (u64)g.aux));
      break;
    case RGEN_ARITHMETIC:  // out = c0*m0*m1 + c1*addend
      out.push_back(rf_add(rf_mul(rf_mul(d(0), d(1)), g.c0), rf_mul(d(2), g.c1)));
      break;
    case RGEN_MUL_EXT: {
      RE2 r = re_muls(re_mul(RE2{d(0), d(1)}, RE2{d(2), d(3)}), g.c0);
      out.push_back(r.a);
      out.push_back(r.b);
      break;
    }
    case RGEN_QUOTIENT_EXT: {
      RE2 r = re_mul(RE2{d(0), d(1)}, re_inv(RE2{d(2), d(3)}));
      out.push_back(r.a);
      out.push_back(r.b);
      break;
    }
    case RGEN_BASE_SPLIT: {  // sum -> 63 little-endian bits
      u64 s = d(0);
      for (u32 i = 0; i < g.n_outs; i++) {
        out.push_back(s & 1);
        s >>= 1;
      }
      break;
    }
    case RGEN_WIRE_SPLIT: {  // integer -> 63-bit chunks, one per BaseSum row
      u64 v = d(0);
      for (u32 i = 0; i < g.n_outs; i++) {
        out.push_back(v & (((u64)1 << 63) - 1));
        v >>= 63;
      }
      break;
    }
    case RGEN_BASE_SUM: {  // limbs (little-endian bits) -> sum
      u64 s = 0;
      for (int i = (int)g.n_deps - 1; i >= 0; i--) s = rf_add(rf_mul(s, 2), d(i) & 1);
      out.push_back(s);
      break;
    }
    case RGEN_LOW_HIGH: {
      u64 v = d(0);
      out.push_back(v & (((u64)1 << g.aux) - 1));
      out.push_back(v >> g.aux);
      break;
    }
    case RGEN_EXPONENTIATION: {  // deps: base, 66 bits (LE); outs: 66 intermediates then output
      u64 base = d(0);
      int nb = (int)g.n_deps - 1;
      u64 cur = 1;
      for (int i = 0; i < nb; i++) {
        u64 prev = i == 0 ? 1 : rf_mul(cur, cur);
        u64 bit = d(1 + (nb - 1 - i));
        cur = bit ? rf_mul(prev, base) : prev;
        out.push_back(cur);
      }
      out.push_back(cur);
      break;
    }
    case RGEN_POSEIDON2: {  // poseidon2_gate.rs:447-523
      u64 st[12];
      for (int i = 0; i < 12; i++) st[i] = d(i);
      u64 swap = d(12);
      for (int i = 0; i < 4; i++) out.push_back(rf_mul(swap, rf_sub(st[i + 4], st[i])));
      if (swap == 1)
        for (int i = 0; i < 4; i++) {
          u64 t = st[i];
          st[i] = st[i + 4];
          st[i + 4] = t;
        }
      u64 tr[106];
      ref_poseidon2_trace(st, tr);
      for (int i = 0; i < 106; i++) out.push_back(tr[i]);
      for (int i = 0; i < 12; i++) out.push_back(st[i]);
      break;
    }
    case RGEN_ARITH_EXT: {  // upstream ArithmeticExtensionGenerator: out = c0*m0*m1 + c1*addend in F_p^2
      RE2 m0{d(0), d(1)}, m1{d(2), d(3)}, ad{d(4), d(5)};
      RE2 r = re_add(re_muls(re_mul(m0, m1), g.c0), re_muls(ad, g.c1));
      out.push_back(r.a);
      out.push_back(r.b);
      break;
    }
    case RGEN_POSEIDON: {  // upstream PoseidonGenerator (hash/poseidon.rs + gates/poseidon.rs)
      u64 st[12];
      for (int i = 0; i < 12; i++) st[i] = d(i);
      u64 swap = d(12);
      for (int i = 0; i < 4; i++) out.push_back(rf_mul(swap, rf_sub(st[i + 4], st[i])));
      if (swap == 1)
        for (int i = 0; i < 4; i++) {
          u64 t = st[i];
          st[i] = st[i + 4];
          st[i + 4] = t;
        }
      u64 tr[106];
      ref_poseidon_trace(st, tr);
      for (int i = 0; i < 106; i++) out.push_back(tr[i]);
      for (int i = 0; i < 12; i++) out.push_back(st[i]);
      break;
    }
    case RGEN_U32_ARITHMETIC: {  // arithmetic_u32.rs:389-439: outs low, high, inverse, 32 two-bit limbs
      u64 o = rf_add(rf_mul(d(0), d(1)), d(2));
      u64 hi = o >> 32, lo = o & 0xFFFFFFFFull;
      out.push_back(lo);
      out.push_back(hi);
      u64 diff = 0xFFFFFFFFull - hi;
      out.push_back(diff == 0 ? 0 : rf_inv(diff));
      for (int j = 0; j < 32; j++) {
        out.push_back(o & 3);
        o >>= 2;
      }
      break;
    }
    case RGEN_U32_INTERLEAVE: {  // interleave_u32.rs:305-334: 32 bits big-endian, then x_interleaved
      u64 x = d(0);
      u64 xi = 0;
      for (int i = 0; i < 32; i++) {
        u64 bit = (x >> (32 - i - 1)) & 1;
        out.push_back(bit);
        xi += bit << (2 * (32 - i - 1));
      }
      out.push_back(xi);
      break;
    }
    case RGEN_U32_UNINTERLEAVE: {  // uninterleave_to_u32.rs:353-390: 64 bits big-endian, evens, odds
      u64 x = d(0);
      u64 ev = 0, od = 0;
      for (int j = 0; j < 32; j++) {
        int shift = 2 * (32 - j - 1);
        u64 e = (x >> (shift + 1)) & 1, o = (x >> shift) & 1;
        out.push_back(e);
        out.push_back(o);
        ev += e << (32 - j - 1);
        od += o << (32 - j - 1);
      }
      out.push_back(ev);
      out.push_back(od);
      break;
    }
    case RGEN_RANDOM_ACCESS: {  // upstream RandomAccessGenerator: index, 16 items -> claimed element, 4 bits
      const u64 idx = d(0);
      if (idx >= 16) throw std::runtime_error("random access index out of range");
      out.push_back(d(1 + (int)idx));
      for (int i = 0; i < 4; i++) out.push_back((idx >> i) & 1);
      break;
    }
    case RGEN_REDUCING:
    case RGEN_REDUCING_EXT: {  // upstream ReducingGenerator: alpha, old_acc, coefficients -> all accumulators
      const bool ext = g.kind == RGEN_REDUCING_EXT;
      const int nco = ext ? 32 : 43;
      RE2 alpha{d(0), d(1)}, acc{d(2), d(3)};
      for (int i = 0; i < nco; i++) {
        acc = re_mul(acc, alpha);
        acc.a = rf_add(acc.a, d(ext ? 4 + 2 * i : 4 + i));
        if (ext) acc.b = rf_add(acc.b, d(5 + 2 * i));
        out.push_back(acc.a);
        out.push_back(acc.b);
      }
      break;
    }
    case RGEN_POSEIDON_MDS: {  // upstream PoseidonMdsGenerator
      static const u64 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
      for (int r = 0; r < 12; r++)
        for (int c2 = 0; c2 < 2; c2++) {
          u64 acc = r == 0 ? rf_mul(d(c2), 8) : 0;
          for (int i = 0; i < 12; i++) acc = rf_add(acc, rf_mul(d(2 * ((i + r) % 12) + c2), CIRC[i]));
          out.push_back(acc);
        }
      break;
    }
    case RGEN_COSET_INTERP: {  // upstream InterpolationGenerator: shift, 16 values, point -> shifted point, states, value
      const u64 shift = d(0);
      const RE2 x = re_muls(RE2{d(33), d(34)}, rf_inv(shift));
      out.push_back(x.a);
      out.push_back(x.b);
      const u64 gen = rf_root_of_unity(4), inv16 = rf_inv(16);
      RE2 eval{0, 0}, prod{1, 0};
      u64 xi = 1;
      for (int c = 0; c < 3; c++) {
        if (c > 0) {
          out.push_back(eval.a);
          out.push_back(eval.b);
          out.push_back(prod.a);
          out.push_back(prod.b);
        }
        const int begin = c == 0 ? 0 : 1 + 5 * c, end = c == 0 ? 6 : (1 + 5 * (c + 1) < 16 ? 1 + 5 * (c + 1) : 16);
        for (int i = begin; i < end; i++) {
          const RE2 v = re_muls(RE2{d(1 + 2 * i), d(2 + 2 * i)}, rf_mul(xi, inv16));
          const RE2 term{rf_sub(x.a, xi), x.b};
          eval = re_add(re_mul(eval, term), re_mul(v, prod));
          prod = re_mul(prod, term);
          xi = rf_mul(xi, gen);
        }
      }
      out.push_back(eval.a);
      out.push_back(eval.b);
      break;
    }
    default:
      throw std::runtime_error("unknown generator kind");
  }
  if (out.size() != g.n_outs) throw std::runtime_error("generator output arity mismatch");
}
}  // namespace

RWitnessResult ref_generate_witness(const RCircuit& c, const u64* inputs, u64 seed, const u64* filler, bool keep_going) {
  RWitnessResult res;
  res.status = 0;
  PartitionWitness w(c);
  for (size_t i = 0; i < c.num_inputs; i++) w.put(c.input_targets[i], inputs[i]);
  const size_t G = c.gens.size();
  // watch lists: representative -> generators depending on it
  std::vector<u32> watch_count(c.num_targets(), 0);
  for (auto& g : c.gens)
    for (u32 i = 0; i < g.n_deps; i++) watch_count[c.rep[c.gen_args[g.arg_off + i]]]++;
  std::vector<size_t> watch_off(c.num_targets() + 1, 0);
  for (size_t i = 0; i < c.num_targets(); i++) watch_off[i + 1] = watch_off[i] + watch_count[i];
  std::vector<u32> watchers(watch_off.back());
  {
    std::vector<size_t> pos(watch_off.begin(), watch_off.end() - 1);
    for (size_t gi = 0; gi < G; gi++) {
      auto& g = c.gens[gi];
      for (u32 i = 0; i < g.n_deps; i++) watchers[pos[c.rep[c.gen_args[g.arg_off + i]]]++] = (u32)gi;
    }
  }
  std::vector<unsigned char> expired(G, 0);
  size_t remaining = G;
  std::vector<u32> pending(G), next;
  for (size_t i = 0; i < G; i++) pending[i] = (u32)i;
  std::vector<u64> out;
  while (!pending.empty()) {
    next.clear();
    for (u32 gi : pending) {
      if (expired[gi]) continue;
      auto& g = c.gens[gi];
      bool ready = true;
      for (u32 i = 0; i < g.n_deps && ready; i++) ready = w.has(c.gen_args[g.arg_off + i]);
      if (!ready) continue;
      run_generator(c, g, w, seed, filler, out);
      expired[gi] = 1;
      remaining--;
      for (u32 i = 0; i < g.n_outs; i++) {
        u32 t = c.gen_args[g.arg_off + g.n_deps + i];
        if (w.put(t, out[i])) {
          u32 r = c.rep[t];
          for (size_t k = watch_off[r]; k < watch_off[r + 1]; k++)
            if (!expired[watchers[k]]) next.push_back(watchers[k]);
        }
      }
    }
    pending.swap(next);
  }
  if (w.conflict) {
    res.status = 4;
    res.message = w.msg;
    if (!keep_going) return res;
  }
  if (remaining) {
    res.status = 5;
    res.message = std::to_string(remaining) + " generators weren't run";
    return res;
  }
  const size_t n = c.n();
  res.wires.assign(c.num_wires, std::vector<u64>(n, 0));
  for (size_t row = 0; row < n; row++)
    for (int col = 0; col < c.num_wires; col++) {
      u32 t = (u32)(row * c.num_wires + col);
      if (w.has(t)) res.wires[col][row] = w.get(t);
    }
  for (u32 t : c.public_inputs) res.public_inputs.push_back(w.has(t) ? w.get(t) : 0);
  return res;
}
