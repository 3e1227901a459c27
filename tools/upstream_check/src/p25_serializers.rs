//! The serializers `CircuitData::to_bytes` / `from_bytes` need for the reference's circuits: upstream's default type
//! lists followed by the reference's gates and generators -- the tag order libp25's `p25_circuit_to_bytes` /
//! `p25_circuit_from_bytes` use (plonky2.5_amd/csrc/circuit_bytes.cpp).  Never compiled here (no Rust toolchain).
use plonky2::gates::arithmetic_base::ArithmeticGate;
use plonky2::gates::arithmetic_extension::ArithmeticExtensionGate;
use plonky2::gates::base_sum::BaseSumGate;
use plonky2::gates::constant::ConstantGate;
use plonky2::gates::coset_interpolation::CosetInterpolationGate;
use plonky2::gates::exponentiation::ExponentiationGate;
use plonky2::gates::lookup::LookupGate;
use plonky2::gates::lookup_table::LookupTableGate;
use plonky2::gates::multiplication_extension::MulExtensionGate;
use plonky2::gates::noop::NoopGate;
use plonky2::gates::poseidon::PoseidonGate;
use plonky2::gates::poseidon_mds::PoseidonMdsGate;
use plonky2::gates::public_input::PublicInputGate;
use plonky2::gates::random_access::RandomAccessGate;
use plonky2::gates::reducing::ReducingGate;
use plonky2::gates::reducing_extension::ReducingExtensionGate;
use plonky2::util::serialization::{GateSerializer, WitnessGeneratorSerializer};
use plonky2::{get_gate_tag_impl, get_generator_tag_impl, impl_gate_serializer, impl_generator_serializer, read_gate_impl, read_generator_impl};
// ... and every generator type of plonky2::util::generator_serialization::default::DefaultGeneratorSerializer,
// imported as there (ArithmeticBaseGenerator ... WireSplitGenerator)

use plonky2_5::common::poseidon2::poseidon2_gate::{Poseidon2Gate, Poseidon2Generator};
use plonky2_5::common::u32::gates::arithmetic_u32::{U32ArithmeticGate, U32ArithmeticGenerator};
use plonky2_5::common::u32::gates::interleave_u32::{U32InterleaveGate, U32InterleaveGenerator};
use plonky2_5::common::u32::gates::uninterleave_to_u32::{UninterleaveToU32Gate, UninterleaveToU32Generator};

pub struct P25GateSerializer;
impl<F: plonky2::hash::hash_types::RichField + plonky2_field::extension::Extendable<D>, const D: usize> GateSerializer<F, D> for P25GateSerializer {
    impl_gate_serializer! {
        P25GateSerializer,
        ArithmeticGate, ArithmeticExtensionGate<D>, BaseSumGate<2>, ConstantGate, CosetInterpolationGate<F, D>,
        ExponentiationGate<F, D>, LookupGate, LookupTableGate, MulExtensionGate<D>, NoopGate, PoseidonMdsGate<F, D>,
        PoseidonGate<F, D>, PublicInputGate, RandomAccessGate<F, D>, ReducingExtensionGate<D>, ReducingGate<D>,   // 0..15
        Poseidon2Gate<F, D>, U32ArithmeticGate<F, D>, U32InterleaveGate, UninterleaveToU32Gate                     // 16..19
    }
}

#[derive(Default)]
pub struct P25GeneratorSerializer<C, const D: usize> { _c: core::marker::PhantomData<C> }
// impl<F, C, const D: usize> WitnessGeneratorSerializer<F, D> for P25GeneratorSerializer<C, D> {
//     impl_generator_serializer! { P25GeneratorSerializer,
//         <the 24 generators of DefaultGeneratorSerializer, in its order>,                                          // 0..23
//         Poseidon2Generator<F, D>, U32ArithmeticGenerator<F, D>, U32InterleaveGenerator, UninterleaveToU32Generator } // 24..27
// }
