//! Builds the circuit of `p3::tests::test_verify_plonky3_proof` (reference src/p3/mod.rs:226-269) with the real
//! plonky2 `CircuitBuilder`, proves it with the real prover, and dumps what tests/test_upstream_golden.py compares:
//!
//!   upstream_circuit.json  degree_bits, circuit_digest, constants_sigmas_cap, gate ids in `common.gates` order,
//!                          rows per gate (from the selector polynomials), selector groups, k_is[0..4]
//!   upstream_proof.json    serde_json::to_string(&proof)          (what src/p3/mod.rs:261 writes)
//!   upstream_filler.json   the values the 131 RandomValueGenerators of the PublicInputGate row drew for THAT proof,
//!                          in wire order 4..134 -- with them p25_prove_batch_filler must reproduce the proof bytes
//!
//! The witness is generated with `generate_partial_witness` so that the filler values can be read back before
//! `prove_with_partition_witness` consumes it (both are public in plonky2 @ 3de92d9: plonky2/src/iop/generator.rs,
//! plonky2/src/plonk/prover.rs).  If a later revision hides them, drop the filler file: the other two still pin
//! the circuit and the verifier semantics.
mod export_blob; // CircuitData -> libp25 circuit blob
mod p25_serializers; // the gate / generator serializers of INTEGRATION.md section 5a

use anyhow::Result;
use plonky2::iop::generator::generate_partial_witness;
use plonky2::iop::target::Target;
use plonky2::iop::witness::{PartialWitness, Witness};
use plonky2::plonk::circuit_builder::CircuitBuilder;
use plonky2::plonk::circuit_data::CircuitConfig;
use plonky2::plonk::config::{GenericConfig, PoseidonGoldilocksConfig};
use plonky2::plonk::prover::prove_with_partition_witness;
use plonky2::util::timing::TimingTree;
use plonky2_field::types::PrimeField64;
use serde_json::json;

// names as in the reference (src/p3/mod.rs:29-94, 151-269).  `FibonacciAir` lives in the reference's `#[cfg(test)] mod tests`
// (src/p3/mod.rs:151-221): README.md "visibility patch" makes that module `pub mod tests` in the checkout this harness builds against.
use plonky2_5::p3::serde::fri::FriConfig;
use plonky2_5::p3::serde::proof::{P3ProofField, Proof};
use plonky2_5::p3::tests::FibonacciAir;
use plonky2_5::p3::CircuitBuilderP3Arithmetic;             // the trait providing `p3_verify_proof` (src/p3/mod.rs:29-48)
use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::hash::poseidon::PoseidonHash;

/// The per-proof input targets in the order `Proof::<Target>::add_virtual_to` creates them (src/p3/serde/proof.rs:357-373 and the
/// nested `add_virtual_to`s :29-343) = the order libp25 takes a proof's words in (`p25_p3_proof_from_json`, tests/p3json.py).
fn flat_targets(p: &Proof<Target>) -> Vec<Target> {
    let mut t = Vec::new();
    t.extend(p.commitments.trace.value);
    t.extend(p.commitments.quotient_chunks.value);
    for e in &p.opened_values.trace_local { t.extend(e.value); }
    for e in &p.opened_values.trace_next { t.extend(e.value); }
    for chunk in &p.opened_values.quotient_chunks { for e in chunk { t.extend(e.value); } }
    let fp = &p.opening_proof.fri_proof;
    for c in &fp.commit_phase_commits { t.extend(c.value); }
    for qp in &fp.query_proofs {
        for step in &qp.commit_phase_openings {
            t.extend(step.sibling_value.value);
            for sib in &step.opening_proof { t.extend(sib.iter().copied()); }
        }
    }
    t.extend(fp.final_poly.value);
    t.push(fp.pow_witness);
    for qo in &p.opening_proof.query_openings {
        for batch in qo {
            for row in &batch.opened_values { t.extend(row.iter().copied()); }
            for sib in &batch.opening_proof { t.extend(sib.iter().copied()); }
        }
    }
    t
}

const D: usize = 2;
type C = PoseidonGoldilocksConfig;
type F = <C as GenericConfig<D>>::F;

fn main() -> Result<()> {
    let artifact = std::env::args().nth(1).unwrap_or("artifacts/proof_fibonacci.json".into());
    let p3_proof: P3ProofField = serde_json::from_str(&std::fs::read_to_string(artifact)?)?;

    let mut builder = CircuitBuilder::<F, D>::new(CircuitConfig::standard_recursion_config());
    // creates the proof's virtual targets itself (Proof::add_virtual_to, proof.rs:357-373) and returns them: src/p3/mod.rs:66-94
    let proof_t: Proof<Target> = builder.p3_verify_proof::<PoseidonHash>(
        p3_proof.clone(),
        &FibonacciAir {},
        FriConfig { log_blowup: 1, num_queries: 100, proof_of_work_bits: 16 },
    );
    let data = builder.build::<C>();
    let (common, prover_only) = (&data.common, &data.prover_only);
    let n = common.degree();

    // ---- upstream_circuit.json
    let cs_values: Vec<Vec<F>> = prover_only.constants_sigmas_commitment.polynomials.iter()
        .map(|p| p.clone().fft().values).collect();
    let mut rows_per_gate = vec![0usize; common.gates.len()];
    let mut pi_row = None;
    for r in 0..n {
        for (g, gate) in common.gates.iter().enumerate() {
            let s = common.selectors_info.selector_indices[g];
            if cs_values[s][r].to_canonical_u64() == g as u64 {
                rows_per_gate[g] += 1;
                if gate.0.id() == "PublicInputGate" { pi_row = Some(r); }
            }
        }
    }
    let circuit = json!({
        "plonky2_rev": "3de92d9ed1721cec133e4e1e1b3ec7facb756ccf",
        "degree_bits": common.degree_bits(),
        "num_gate_constraints": common.num_gate_constraints,
        "num_partial_products": common.num_partial_products,
        "quotient_degree_factor": common.quotient_degree_factor,
        "circuit_digest": data.verifier_only.circuit_digest.elements.iter().map(|e| e.to_canonical_u64()).collect::<Vec<_>>(),
        "constants_sigmas_cap": data.verifier_only.constants_sigmas_cap.0.iter()
            .map(|h| h.elements.iter().map(|e| e.to_canonical_u64()).collect::<Vec<_>>()).collect::<Vec<_>>(),
        "gate_ids": common.gates.iter().map(|g| g.0.id()).collect::<Vec<_>>(),
        "rows_per_gate": rows_per_gate,
        "selector_indices": common.selectors_info.selector_indices,
        "selector_groups": common.selectors_info.groups.iter().map(|r| vec![r.start, r.end]).collect::<Vec<_>>(),
        "fri_reduction_arity_bits": common.fri_params.reduction_arity_bits,
        "k_is_head": common.k_is.iter().take(4).map(|e| e.to_canonical_u64()).collect::<Vec<_>>(),
        "num_generators": prover_only.generators.len(),
    });
    std::fs::write("upstream_circuit.json", serde_json::to_string_pretty(&circuit)?)?;

    // ---- witness, filler, proof
    let mut pw = PartialWitness::new();
    // as the reference's test does (src/p3/mod.rs:254-257): Value<GoldilocksField> and GoldilocksField have the same layout
    let p: Proof<GoldilocksField> = unsafe { std::mem::transmute(p3_proof) };
    proof_t.set_witness::<F, D, _>(&mut pw, &p);                              // proof.rs:374-383
    let partition_witness = generate_partial_witness(pw, prover_only, common);
    if let Some(r) = pi_row {
        let filler: Vec<u64> = (4..common.config.num_wires)
            .map(|c| partition_witness.get_target(Target::wire(r, c)).to_canonical_u64()).collect();
        std::fs::write("upstream_filler.json", serde_json::to_string(&json!({"pi_row": r, "filler": filler}))?)?;
    }
    let proof = prove_with_partition_witness::<F, C, D>(prover_only, common, partition_witness,
                                                          &mut TimingTree::default())?;
    std::fs::write("upstream_proof.json", serde_json::to_string(&proof)?)?;
    // the binary forms libp25 restates: ProofWithPublicInputs::to_bytes (p25_proof_to_bytes) and CircuitData::to_bytes
    // with the serializers of INTEGRATION.md section 5a (p25_circuit_to_bytes / p25_circuit_from_bytes).  The circuit
    // bytes are ~600 MB; only their length, a SHA-256 and the first 64 KiB (config, FRI params, selectors, gate list,
    // the head of the generator list) are kept as a fixture.
    std::fs::write("upstream_proof.bin", proof.to_bytes())?;
    {
        use sha2::{Digest, Sha256};
        let bytes = data.to_bytes(&p25_serializers::P25GateSerializer, &p25_serializers::P25GeneratorSerializer::<C, D>::default())
            .map_err(|e| anyhow::anyhow!("CircuitData::to_bytes: {e:?}"))?;
        let head = &bytes[..bytes.len().min(1 << 16)];
        std::fs::write("upstream_circuit_data_head.bin", head)?;
        std::fs::write("upstream_circuit_data.json", serde_json::to_string(&json!({
            "len": bytes.len(), "sha256": format!("{:x}", Sha256::digest(&bytes)), "head_len": head.len() }))?)?;
    }
    data.verify(proof.clone())?;                                              // src/p3/mod.rs:266
    // ---- the recursive verifier of that proof: upstream's own `builder.verify_proof`, so that libp25's
    // `p25_circuit_build_recursive_verifier` -- restated from memory, NOT claimed row for row -- becomes
    // checkable: gate table, rows per gate and circuit digest of the circuit upstream builds for one inner proof with the
    // inner verifier data as constants (tests/test_upstream_golden.py::test_recursive_verifier_shape_vs_upstream).
    {
        let mut rb = CircuitBuilder::<F, D>::new(CircuitConfig::standard_recursion_config());
        let pt = rb.add_virtual_proof_with_pis(&data.common);
        let vd = rb.constant_verifier_data(&data.verifier_only);
        rb.verify_proof::<C>(&pt, &vd, &data.common);
        let rdata = rb.build::<C>();
        let rc = &rdata.common;
        let rn = rc.degree();
        let rvals: Vec<Vec<F>> = rdata.prover_only.constants_sigmas_commitment.polynomials.iter()
            .map(|p| p.clone().fft().values).collect();
        let mut rrows = vec![0usize; rc.gates.len()];
        for r in 0..rn {
            for g in 0..rc.gates.len() {
                let s = rc.selectors_info.selector_indices[g];
                if rvals[s][r].to_canonical_u64() == g as u64 { rrows[g] += 1; }
            }
        }
        let mut rpw = PartialWitness::new();
        rpw.set_proof_with_pis_target(&pt, &proof);
        let rproof = rdata.prove(rpw)?;
        std::fs::write("upstream_recursive_circuit.json", serde_json::to_string_pretty(&json!({
            "degree_bits": rc.degree_bits(),
            "gate_ids": rc.gates.iter().map(|g| g.0.id()).collect::<Vec<_>>(),
            "rows_per_gate": rrows,
            "num_gate_constraints": rc.num_gate_constraints,
            "num_generators": rdata.prover_only.generators.len(),
            "circuit_digest": rdata.verifier_only.circuit_digest.elements.iter().map(|e| e.to_canonical_u64()).collect::<Vec<_>>(),
        }))?)?;
        std::fs::write("upstream_recursive_proof.json", serde_json::to_string(&rproof)?)?;
        rdata.verify(rproof)?;
    }
    // the circuit as libp25 takes it (p25_circuit_import): prove it on the GPU, feed the proof back to data.verify
    std::fs::write("upstream_circuit.p25blob", export_blob::export_p25_blob(&data, &flat_targets(&proof_t)))?;
    println!("wrote upstream_circuit.json, upstream_proof.json, upstream_filler.json (n = 2^{})", common.degree_bits());
    Ok(())
}
