#include "witness_program.h"
#include <algorithm>
#include <stdexcept>
#include <string>

namespace p25 {

WitnessProgram build_witness_program(const Circuit& c) {
  WitnessProgram wp;
  const size_t NT = c.num_targets();
  const size_t G = c.generators.size();
  std::vector<uint32_t> slot_of_rep(NT, 0);       // 0 = unassigned
  std::vector<uint32_t> set_round(NT, UINT32_MAX);  // round in which the rep gets its value
  auto rep_of = [&](Target t) { return c.rep[c.target_index(t)]; };

  // inputs are assigned before round 1
  {
    std::vector<uint32_t> first_input(NT, UINT32_MAX);
    for (size_t i = 0; i < c.input_targets.size(); i++) {
      uint32_t r = rep_of(c.input_targets[i]);
      if (slot_of_rep[r] == 0) {
        slot_of_rep[r] = wp.num_slots++;
        set_round[r] = 0;
        first_input[r] = (uint32_t)i;
        wp.input_slots.push_back(slot_of_rep[r]);
        wp.input_first.push_back((uint32_t)i);
      } else {
        wp.input_slots.push_back(slot_of_rep[r] | WIT_CHECK_FLAG);
        wp.input_first.push_back(first_input[r]);
      }
    }
  }

  // watch lists
  std::vector<uint32_t> wcount(NT + 1, 0);
  for (const auto& g : c.generators)
    for (const Target& t : g.deps) wcount[rep_of(t) + 1]++;
  for (size_t i = 0; i < NT; i++) wcount[i + 1] += wcount[i];
  std::vector<uint32_t> watchers(wcount[NT]);
  {
    std::vector<uint32_t> pos(wcount.begin(), wcount.end() - 1);
    for (size_t gi = 0; gi < G; gi++)
      for (const Target& t : c.generators[gi].deps) watchers[pos[rep_of(t)]++] = (uint32_t)gi;
  }
  std::vector<uint32_t> missing(G, 0);
  for (size_t gi = 0; gi < G; gi++)
    for (const Target& t : c.generators[gi].deps)
      if (set_round[rep_of(t)] == UINT32_MAX) missing[gi]++;

  std::vector<uint32_t> level_of(G, UINT32_MAX);
  std::vector<std::vector<uint32_t>> out_flags(G);
  std::vector<uint32_t> ready;
  for (size_t gi = 0; gi < G; gi++)
    if (missing[gi] == 0) ready.push_back((uint32_t)gi);
  size_t done = 0;
  uint32_t round = 0;
  std::vector<uint32_t> next_ready, deferred, newly_set;
  while (!ready.empty()) {
    round++;
    std::sort(ready.begin(), ready.end());
    next_ready.clear();
    deferred.clear();
    newly_set.clear();
    for (uint32_t gi : ready) {
      const Generator& g = c.generators[gi];
      bool clash = false;  // an output partition was first-assigned by another generator this round
      for (const Target& t : g.outs)
        if (set_round[rep_of(t)] == round) clash = true;
      // (a generator writing the same partition through two of its own outputs is handled below)
      if (clash) {
        deferred.push_back(gi);
        continue;
      }
      level_of[gi] = round;
      out_flags[gi].resize(g.outs.size());
      std::vector<uint32_t> mine;
      for (size_t k = 0; k < g.outs.size(); k++) {
        uint32_t r = rep_of(g.outs[k]);
        bool own = std::find(mine.begin(), mine.end(), r) != mine.end();
        if (set_round[r] == UINT32_MAX) {
          slot_of_rep[r] = wp.num_slots++;
          set_round[r] = round;
          mine.push_back(r);
          newly_set.push_back(r);
          out_flags[gi][k] = 0;
        } else if (own) {
          // second output of the same generator into one partition: cannot compare in-kernel
          // without ordering; defer semantics: treat as CHECK against the value this lane wrote
          out_flags[gi][k] = WIT_CHECK_FLAG;
        } else {
          out_flags[gi][k] = WIT_CHECK_FLAG;
        }
      }
      done++;
    }
    for (uint32_t r : newly_set)
      for (uint32_t k = wcount[r]; k < wcount[r + 1]; k++) {
        uint32_t w = watchers[k];
        if (--missing[w] == 0) next_ready.push_back(w);
      }
    next_ready.insert(next_ready.end(), deferred.begin(), deferred.end());
    ready.swap(next_ready);
  }
  if (done != G) throw std::runtime_error(std::to_string(G - done) + " generators weren't run");

  // RandomValueGenerators are numbered in generator-list order (the order an explicit filler vector is given in)
  std::vector<uint32_t> random_ordinal(G, 0);
  for (size_t gi = 0; gi < G; gi++)
    if (c.generators[gi].kind == GEN_RANDOM) random_ordinal[gi] = wp.num_random_fill++;
  // emit generators sorted by (level, kind, original index)
  std::vector<uint32_t> order(G);
  for (size_t i = 0; i < G; i++) order[i] = (uint32_t)i;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    if (level_of[a] != level_of[b]) return level_of[a] < level_of[b];
    if (c.generators[a].kind != c.generators[b].kind) return c.generators[a].kind < c.generators[b].kind;
    return a < b;
  });
  wp.level_start.push_back(0);
  uint32_t cur = 1;
  for (uint32_t gi : order) {
    while (level_of[gi] > cur) {
      wp.level_start.push_back((uint32_t)wp.gens.size());
      cur++;
    }
    const Generator& g = c.generators[gi];
    WitGen w;
    w.kind = g.kind;
    w.aux = (uint32_t)g.aux;
    w.arg_off = (uint32_t)wp.args.size();
    w.n_deps = (uint16_t)g.deps.size();
    w.n_outs = (uint16_t)g.outs.size();
    w.c0 = g.c0;
    w.c1 = g.c1;
    if (g.kind == GEN_RANDOM) w.c1 = random_ordinal[gi];  // index into an explicit filler vector
    for (const Target& t : g.deps) wp.args.push_back(slot_of_rep[rep_of(t)]);
    for (size_t k = 0; k < g.outs.size(); k++) wp.args.push_back(slot_of_rep[rep_of(g.outs[k])] | out_flags[gi][k]);
    wp.gens.push_back(w);
  }
  wp.level_start.push_back((uint32_t)wp.gens.size());

  for (const Target& t : c.public_inputs) wp.pi_slots.push_back(slot_of_rep[rep_of(t)]);
  const size_t n = c.degree();
  const int W = c.cfg.num_wires;
  wp.wire_slot_cm.resize((size_t)W * n);
  for (size_t row = 0; row < n; row++)
    for (int col = 0; col < W; col++) wp.wire_slot_cm[(size_t)col * n + row] = slot_of_rep[c.rep[row * W + col]];
  return wp;
}

}  // namespace p25
