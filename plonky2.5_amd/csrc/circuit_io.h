// Flat binary form of a built `Circuit` ("circuit blob"): what crosses the C ABI
// (p25_circuit_export / p25_circuit_import) so a circuit is built once and persisted, and what the
// CPU oracle loads to prove the same circuit independently.  Layout (little-endian u64 words unless
// noted; arrays of u32 are padded to a multiple of 8 bytes):
//   magic "P25CIRC1", header[32], sorted gates [num_gates][4] = {kind, selector_index, group_start,
//   group_end}, fri arity bits [n_arity], row kinds u32[n], constants_sigmas [num_cs_polys][n],
//   k_is [num_routed], input target indices u32[num_inputs], representative map u32[num_targets],
//   generator table: per generator {kind, c0, c1, aux, n_deps, n_outs} then u32 target indices.
#pragma once
#include <vector>
#include "builder.h"
namespace p25 {
std::vector<uint8_t> circuit_to_blob(const Circuit& c);
Circuit circuit_from_blob(const uint8_t* data, size_t len);
}
