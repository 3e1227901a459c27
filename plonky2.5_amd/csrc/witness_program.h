// Levelised witness-generation program.
//
// Upstream plonky2 generates the witness with a sequential, data-dependent work-list
// (generate_partial_witness; called from /root/reference/src/p3/mod.rs:260; generator bodies are the
// reference's SimpleGenerator::run_once implementations cited in kernels_witgen.hip).  For a FIXED
// circuit the schedule is static, so it is computed once here: every generator gets a level such
// that all its dependencies are produced at earlier levels, every copy-constraint class
// ("partition") that is ever assigned gets a compact SLOT, and each generator output is classified
//   WRITE -- first assignment of its slot, or
//   CHECK -- the slot was assigned earlier; the GPU compares instead of writing and raises
//            P25_ERR_WITNESS_CONFLICT (upstream panics "was set twice with different values").
// One kernel launch per level then runs (generators of the level) x (proofs of the batch) lanes.
#pragma once
#include <stdint.h>
#include <vector>
#include "builder.h"

namespace p25 {

struct WitGen {       // 32 bytes, mirrored in the kernel
  uint32_t kind;      // GenKind
  uint32_t aux;
  uint32_t arg_off;   // into WitnessProgram::args: dep slots, then out slots
  uint16_t n_deps, n_outs;
  u64 c0, c1;
};
constexpr uint32_t WIT_CHECK_FLAG = 0x80000000u;  // on an out slot: compare, do not write

struct WitnessProgram {
  uint32_t num_random_fill = 0;           // RandomValueGenerators; WitGen::c1 of each = its ordinal
  uint32_t num_slots = 1;                 // slot 0 is the constant 0 (value of every unset wire)
  std::vector<WitGen> gens;               // sorted by (level, kind)
  std::vector<uint32_t> args;
  std::vector<uint32_t> level_start;      // gens[level_start[l] .. level_start[l+1]) is level l
  std::vector<uint32_t> input_slots;      // slot of each per-proof input
  // An input whose copy-constraint partition was already assigned by an EARLIER input (two connected
  // inputs; never the case for the p3 circuit) is compared with that first input instead of stored:
  // input_slots[i] carries WIT_CHECK_FLAG and input_first[i] is the index of the first writer.
  std::vector<uint32_t> input_first;
  std::vector<uint32_t> wire_slot_cm;     // [num_wires][degree] column-major: slot of each wire
  std::vector<uint32_t> pi_slots;         // slot of each registered public input (Circuit::public_inputs order)
};

// Throws std::runtime_error("N generators weren't run") if the circuit is not fully determined.
WitnessProgram build_witness_program(const Circuit& c);

}  // namespace p25
