#pragma once
#include <string>
#include <vector>
#include "p3_circuit.h"
#include "prover.h"
namespace p25 {
void p3_proof_from_json(const char* json, size_t len, std::vector<u64>& inputs, P3Config& cfg);
std::string proof_to_json(const Circuit& c, const ProofLayout& L, const u64* proof_words);
std::vector<uint8_t> proof_to_bytes(const Circuit& c, const ProofLayout& L, const u64* proof_words);
void proof_from_bytes(const Circuit& c, const ProofLayout& L, const uint8_t* data, size_t len, u64* proof_words_out);
}
