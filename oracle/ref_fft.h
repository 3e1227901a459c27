// ORACLE -- test infrastructure only.  Nothing in plonky2.5_amd/ may include, link or call this.
//
// Textbook radix-2 NTT over Goldilocks restating what upstream plonky2_field @ 3de92d9 computes
// (polynomial.rs / fft.rs): fft(c)[i] = sum_k c[k] w^(ik) in natural order,
// ifft its inverse, coset_fft(c, s)[i] = f(s*w^i), lde = zero-pad then coset_fft.
#pragma once
#include <vector>
#include "ref_field.h"
void ref_fft(std::vector<u64>& a);            // coefficients -> values (natural order)
void ref_ifft(std::vector<u64>& a);           // values -> coefficients
void ref_coset_fft(std::vector<u64>& a, u64 shift);
void ref_coset_ifft(std::vector<u64>& a, u64 shift);
void ref_fft_ext(std::vector<RE2>& a);
void ref_coset_fft_ext(std::vector<RE2>& a, u64 shift);
// lde: coefficient vector -> values of the rate-2^rate_bits LDE on shift*<w>, natural order
std::vector<u64> ref_lde_values(const std::vector<u64>& coeffs, unsigned rate_bits, u64 shift);
// the same values at BIT-REVERSED index (out[rbits(i)] = f(shift w^i)): the Merkle-leaf order, computed without a
// permutation pass (decimation in frequency)
std::vector<u64> ref_lde_values_bitrev(const std::vector<u64>& coeffs, unsigned rate_bits, u64 shift);

// tuned cpu_baseline leg: AVX-512 butterflies in the stages with >= 8 butterflies per block (same values); off for the checker
void ref_fft_set_tuned(bool on);
