//! Exporter of an upstream-built `CircuitData` to the circuit blob of INTEGRATION.md section 5, so that a host which
//! keeps `builder.build::<C>()` in Rust (reference src/p3/mod.rs:250) can hand the circuit to libp25
//! (`p25_circuit_import`).  Text only -- never compiled in the build container (no Rust toolchain there); field names
//! follow plonky2 @ 3de92d9 as the author remembers them and may need small adjustments.  The format itself is pinned
//! by tests/blob_writer.py (an independent writer) and the validation in plonky2.5_amd/csrc/circuit_io.cpp.
//!
//! Generators are recovered from their own `serialize` output (every upstream / reference generator writes the row,
//! the op index and its constants; e.g. reference poseidon2_gate.rs:529-539, arithmetic_u32.rs:445-464) and expanded to
//! wire targets with the wire layouts tabulated in INTEGRATION.md.
use plonky2::field::extension::Extendable;
use plonky2::field::types::{Field, PrimeField64};
use plonky2::hash::hash_types::RichField;
use plonky2::iop::target::Target;
use plonky2::plonk::circuit_data::{CircuitData, CommonCircuitData};
use plonky2::plonk::config::GenericConfig;
use plonky2::util::serialization::{Buffer, Read};

pub struct Blob(pub Vec<u8>);
impl Blob {
    fn u64(&mut self, v: u64) { self.0.extend_from_slice(&v.to_le_bytes()); }
    fn u64s(&mut self, v: &[u64]) { for &x in v { self.u64(x); } }
    fn u32s(&mut self, v: &[u32]) {
        for &x in v { self.0.extend_from_slice(&x.to_le_bytes()); }
        if v.len() % 2 == 1 { self.0.extend_from_slice(&[0u8; 4]); }          // u32 arrays are padded to 8 bytes
    }
    fn fields<F: PrimeField64>(&mut self, v: &[F]) { for x in v { self.u64(x.to_canonical_u64()); } }
}

fn gate_kind(id: &str) -> u64 {
    match id {
        "NoopGate" => 0,
        s if s.starts_with("ConstantGate") => 1,
        "PublicInputGate" => 2,
        s if s.starts_with("BaseSumGate") => 3,
        s if s.starts_with("U32InterleaveGate") => 4,
        s if s.starts_with("UninterleaveToU32Gate") => 5,
        s if s.starts_with("ArithmeticGate") => 6,
        s if s.starts_with("MulExtensionGate") => 7,
        s if s.starts_with("ExponentiationGate") => 8,
        s if s.starts_with("U32ArithmeticGate") => 9,
        s if s.starts_with("Poseidon2Gate") => 10,
        s if s.starts_with("ArithmeticExtensionGate") => 11,
        s if s.starts_with("PoseidonGate") => 12,
        other => panic!("gate {other} is not supported by libp25"),
    }
}

/// `inputs`: the targets the host assigns per proof, in the order it will pass their values
/// (for the plonky3 verifier circuit: `Proof::<Target>::add_virtual_to` order, reference proof.rs:357-373).
pub fn export_p25_blob<F, C, const D: usize>(data: &CircuitData<F, C, D>, inputs: &[Target]) -> Vec<u8>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
{
    assert_eq!(D, 2);
    let (c, p) = (&data.common, &data.prover_only);
    let n = c.degree();
    let nw = c.config.num_wires;
    let tidx = |t: &Target| -> u32 {
        (match t {
            Target::Wire(w) => w.row * nw + w.column,
            Target::VirtualTarget { index } => n * nw + index,
        }) as u32
    };
    let cs: Vec<Vec<F>> = p.constants_sigmas_commitment.polynomials.iter().map(|poly| poly.clone().fft().values).collect();
    // row kinds from the selector values; the PublicInputGate row on the way
    let mut kinds = vec![0u32; n];
    let mut pi_row = u64::MAX;
    for r in 0..n {
        for (g, gate) in c.gates.iter().enumerate() {
            if cs[c.selectors_info.selector_indices[g]][r].to_canonical_u64() == g as u64 {
                kinds[r] = gate_kind(&gate.0.id()) as u32;
                if kinds[r] == 2 { pi_row = r as u64; }
            }
        }
    }
    let num_virtual = p.representative_map.len() - n * nw;

    let mut b = Blob(b"P25CIRC1".to_vec());
    let mut h = [0u64; 32];
    h[0] = c.degree_bits() as u64;
    h[1] = nw as u64;
    h[2] = c.config.num_routed_wires as u64;
    h[3] = c.config.num_constants as u64;
    h[4] = c.config.num_challenges as u64;
    h[5] = c.config.max_quotient_degree_factor as u64;
    h[6] = c.config.fri_config.rate_bits as u64;
    h[7] = c.config.fri_config.cap_height as u64;
    h[8] = c.config.fri_config.proof_of_work_bits as u64;
    h[9] = c.config.fri_config.num_query_rounds as u64;
    h[10] = c.fri_params.reduction_arity_bits.len() as u64;
    h[11] = c.selectors_info.num_selectors() as u64;
    h[12] = c.num_gate_constraints as u64;
    h[13] = c.num_partial_products as u64;
    h[14] = c.gates.len() as u64;
    h[15] = pi_row;
    h[16] = num_virtual as u64;
    h[17] = inputs.len() as u64;
    h[18] = p.generators.len() as u64;
    h[19] = cs.len() as u64;
    h[20] = 4; // FriReductionStrategy::ConstantArityBits(4, 5) of standard_recursion_config (informational)
    h[21] = 5;
    b.u64s(&h);
    for (i, g) in c.gates.iter().enumerate() {
        let s = c.selectors_info.selector_indices[i];
        let r = &c.selectors_info.groups[s];
        b.u64s(&[gate_kind(&g.0.id()), s as u64, r.start as u64, r.end as u64]);
    }
    b.u64s(&c.fri_params.reduction_arity_bits.iter().map(|&a| a as u64).collect::<Vec<_>>());
    b.u32s(&kinds);
    for v in &cs { b.fields(v); }
    b.fields(&c.k_is);
    b.u32s(&inputs.iter().map(tidx).collect::<Vec<_>>());
    b.u32s(&p.representative_map.iter().map(|&r| r as u32).collect::<Vec<_>>());
    for g in &p.generators {
        let mut bytes = Vec::new();
        g.0.serialize(&mut bytes, c).expect("generator serialisation");
        write_generator(&mut b, &g.0.id(), &mut Buffer::new(&bytes), c, &tidx);
    }
    b.0
}

/// One record of the generator table: u64[6] = kind, c0, c1, aux, n_deps, n_outs, then the target indices.
fn write_generator<F: RichField + Extendable<D>, const D: usize>(
    b: &mut Blob, id: &str, src: &mut Buffer, c: &CommonCircuitData<F, D>, tidx: &dyn Fn(&Target) -> u32,
) {
    let w = |row: usize, col: usize| Target::wire(row, col);
    let f = |x: F| x.to_canonical_u64();
    let (kind, c0, c1, aux, deps, outs): (u64, u64, u64, u64, Vec<Target>, Vec<Target>) = match id {
        "ConstantGenerator" => {
            let (row, _ci, wi, k) = (src.read_usize().unwrap(), src.read_usize().unwrap(), src.read_usize().unwrap(), src.read_field::<F>().unwrap());
            (0, f(k), 0, 0, vec![], vec![w(row, wi)])
        }
        "RandomValueGenerator" => {
            let t = src.read_target().unwrap();
            let col = match t { Target::Wire(x) => x.column, _ => 0 };
            (1, 0, 0, col as u64, vec![], vec![t])
        }
        "ArithmeticBaseGenerator" => {
            let (row, k0, k1, i) = (src.read_usize().unwrap(), src.read_field::<F>().unwrap(), src.read_field::<F>().unwrap(), src.read_usize().unwrap());
            (2, f(k0), f(k1), 0, vec![w(row, 4 * i), w(row, 4 * i + 1), w(row, 4 * i + 2)], vec![w(row, 4 * i + 3)])
        }
        "MulExtensionGenerator" => {
            let (row, k0, i) = (src.read_usize().unwrap(), src.read_field::<F>().unwrap(), src.read_usize().unwrap());
            (3, f(k0), 0, 0, (0..4).map(|k| w(row, 6 * i + k)).collect(), vec![w(row, 6 * i + 4), w(row, 6 * i + 5)])
        }
        "QuotientGeneratorExtension" => {
            let (num, den, quo) = (src.read_target_ext::<D>().unwrap(), src.read_target_ext::<D>().unwrap(), src.read_target_ext::<D>().unwrap());
            (4, 0, 0, 0, vec![num.0[0], num.0[1], den.0[0], den.0[1]], vec![quo.0[0], quo.0[1]])
        }
        s if s.starts_with("BaseSplitGenerator") => {
            let (row, limbs) = (src.read_usize().unwrap(), src.read_usize().unwrap());
            (5, 0, 0, 0, vec![w(row, 0)], (0..limbs).map(|l| w(row, 1 + l)).collect())
        }
        "WireSplitGenerator" => {
            let (integer, gates, _limbs) = (src.read_target().unwrap(), src.read_usize_vec().unwrap(), src.read_usize().unwrap());
            (6, 0, 0, 0, vec![integer], gates.iter().map(|&g| w(g, 0)).collect())
        }
        s if s.starts_with("BaseSumGenerator") => {
            let (row, limbs) = (src.read_usize().unwrap(), src.read_target_bool_vec().unwrap());
            (7, 0, 0, 0, limbs.iter().map(|l| l.target).collect(), vec![w(row, 0)])
        }
        "LowHighGenerator" => {
            let (x, n_log, lo, hi) = (src.read_target().unwrap(), src.read_usize().unwrap(), src.read_target().unwrap(), src.read_target().unwrap());
            (8, 0, 0, n_log as u64, vec![x], vec![lo, hi])
        }
        s if s.starts_with("ExponentiationGenerator") => {
            let row = src.read_usize().unwrap();
            let bits = 66; // ExponentiationGate::new_from_config(standard_recursion_config)
            let mut outs: Vec<Target> = (0..bits).map(|i| w(row, 2 + bits + i)).collect();
            outs.push(w(row, 1 + bits));
            (9, 0, 0, 0, (0..=bits).map(|i| w(row, i)).collect(), outs)
        }
        "Poseidon2Generator" | "PoseidonGenerator" => {
            let row = src.read_usize().unwrap();
            let mut deps: Vec<Target> = (0..12).map(|i| w(row, i)).collect();
            deps.push(w(row, 24));
            let mut outs: Vec<Target> = (0..4).map(|i| w(row, 25 + i)).collect();
            outs.extend((0..106).map(|i| w(row, 29 + i)));
            outs.extend((0..12).map(|i| w(row, 12 + i)));
            (if id == "Poseidon2Generator" { 10 } else { 15 }, 0, 0, 0, deps, outs)
        }
        "U32ArithmeticGenerator" => {
            let (_ops, row, i) = (src.read_usize().unwrap(), src.read_usize().unwrap(), src.read_usize().unwrap());
            let mut outs = vec![w(row, 6 * i + 3), w(row, 6 * i + 4), w(row, 6 * i + 5)];
            outs.extend((0..32).map(|j| w(row, 18 + 32 * i + j)));
            (11, 0, 0, 0, vec![w(row, 6 * i), w(row, 6 * i + 1), w(row, 6 * i + 2)], outs)
        }
        "U32InterleaveGenerator" => {
            let (_ops, row, i) = (src.read_usize().unwrap(), src.read_usize().unwrap(), src.read_usize().unwrap());
            let mut outs: Vec<Target> = (0..32).map(|j| w(row, 6 + 32 * i + j)).collect();
            outs.push(w(row, 2 * i + 1));
            (12, 0, 0, 0, vec![w(row, 2 * i)], outs)
        }
        "UninterleaveToU32Generator" => {
            let (_ops, row, i) = (src.read_usize().unwrap(), src.read_usize().unwrap(), src.read_usize().unwrap());
            let mut outs: Vec<Target> = (0..64).map(|j| w(row, 6 + 64 * i + j)).collect();
            outs.push(w(row, 3 * i + 1));
            outs.push(w(row, 3 * i + 2));
            (13, 0, 0, 0, vec![w(row, 3 * i)], outs)
        }
        "ArithmeticExtensionGenerator" => {
            let (row, k0, k1, i) = (src.read_usize().unwrap(), src.read_field::<F>().unwrap(), src.read_field::<F>().unwrap(), src.read_usize().unwrap());
            (14, f(k0), f(k1), 0, (0..6).map(|k| w(row, 8 * i + k)).collect(), vec![w(row, 8 * i + 6), w(row, 8 * i + 7)])
        }
        other => panic!("generator {other} is not supported by libp25"),
    };
    let _ = c;
    b.u64s(&[kind, c0, c1, aux, deps.len() as u64, outs.len() as u64]);
    let mut args: Vec<u32> = deps.iter().map(tidx).collect();
    args.extend(outs.iter().map(tidx));
    b.u32s(&args);
}
