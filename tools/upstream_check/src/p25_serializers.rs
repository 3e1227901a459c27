//! The serializers `CircuitData::to_bytes` / `from_bytes` need for the reference's circuits: upstream's default type
//! lists followed by the reference's gates and generators -- the tag order libp25's `p25_circuit_to_bytes` /
//! `p25_circuit_from_bytes` use (plonky2.5_amd/csrc/circuit_bytes.cpp).  Never compiled here (no Rust toolchain): written against
//! the module paths of plonky2 @ 3de92d9; `cargo check` is the first thing to run where cargo exists (README.md).
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
// every generator type of plonky2::util::serialization::generator_serialization::default::DefaultGeneratorSerializer, imported
// as that module imports them (plonky2 @ 3de92d9: plonky2/src/util/serialization/generator_serialization.rs `pub mod default`)
use plonky2::gadgets::arithmetic::EqualityGenerator;
use plonky2::gadgets::arithmetic_extension::QuotientGeneratorExtension;
use plonky2::gadgets::range_check::LowHighGenerator;
use plonky2::gadgets::split_base::BaseSumGenerator;
use plonky2::gadgets::split_join::{SplitGenerator, WireSplitGenerator};
use plonky2::gates::arithmetic_base::ArithmeticBaseGenerator;
use plonky2::gates::arithmetic_extension::ArithmeticExtensionGenerator;
use plonky2::gates::base_sum::BaseSplitGenerator;
use plonky2::gates::coset_interpolation::InterpolationGenerator;
use plonky2::gates::exponentiation::ExponentiationGenerator;
use plonky2::gates::lookup::LookupGenerator;
use plonky2::gates::lookup_table::LookupTableGenerator;
use plonky2::gates::multiplication_extension::MulExtensionGenerator;
use plonky2::gates::poseidon::PoseidonGenerator;
use plonky2::gates::poseidon_mds::PoseidonMdsGenerator;
use plonky2::gates::random_access::RandomAccessGenerator;
use plonky2::gates::reducing::ReducingGenerator;
use plonky2::gates::reducing_extension::ReducingGenerator as ReducingExtensionGenerator;
use plonky2::hash::hash_types::RichField;
use plonky2::iop::generator::{ConstantGenerator, CopyGenerator, NonzeroTestGenerator, RandomValueGenerator};
use plonky2::plonk::config::{AlgebraicHasher, GenericConfig};
use plonky2::recursion::dummy_circuit::DummyProofGenerator;
use plonky2_field::extension::Extendable;

// Two of the reference's generator types are private to their modules -- `struct Poseidon2Generator`
// (src/common/poseidon2/poseidon2_gate.rs:431) and `struct U32ArithmeticGenerator` (src/common/u32/gates/arithmetic_u32.rs:369)
// -- and need `pub` in the checkout the harness builds against (README.md: "visibility patch"); the other two
// (interleave_u32.rs:290, uninterleave_to_u32.rs:338) are public already.
use plonky2_5::common::poseidon2::poseidon2::Poseidon2;
use plonky2_5::common::poseidon2::poseidon2_gate::{Poseidon2Gate, Poseidon2Generator};
use plonky2_5::common::u32::gates::arithmetic_u32::{U32ArithmeticGate, U32ArithmeticGenerator};
use plonky2_5::common::u32::gates::interleave_u32::{U32InterleaveGate, U32InterleaveGenerator};
use plonky2_5::common::u32::gates::uninterleave_to_u32::{UninterleaveToU32Gate, UninterleaveToU32Generator};

pub struct P25GateSerializer;
impl<F: RichField + Extendable<D> + Poseidon2, const D: usize> GateSerializer<F, D> for P25GateSerializer {
    impl_gate_serializer! {
        P25GateSerializer,
        ArithmeticGate, ArithmeticExtensionGate<D>, BaseSumGate<2>, ConstantGate, CosetInterpolationGate<F, D>,
        ExponentiationGate<F, D>, LookupGate, LookupTableGate, MulExtensionGate<D>, NoopGate, PoseidonMdsGate<F, D>,
        PoseidonGate<F, D>, PublicInputGate, RandomAccessGate<F, D>, ReducingExtensionGate<D>, ReducingGate<D>,   // 0..15
        Poseidon2Gate<F, D>, U32ArithmeticGate<F, D>, U32InterleaveGate, UninterleaveToU32Gate                     // 16..19
    }
}

/// Tags 0..23 are upstream's `DefaultGeneratorSerializer` list IN ITS ORDER (the macro numbers the types by position, so the
/// order is the wire format); 24..27 are the reference's generators.  The same numbering as the `GT_*` enum of
/// plonky2.5_amd/csrc/circuit_bytes.cpp -- keep the two in step.
pub struct P25GeneratorSerializer<C: GenericConfig<D>, const D: usize> {
    pub _phantom: core::marker::PhantomData<C>,
}
impl<C: GenericConfig<D>, const D: usize> Default for P25GeneratorSerializer<C, D> {
    fn default() -> Self {
        Self { _phantom: core::marker::PhantomData }
    }
}
impl<F, C, const D: usize> WitnessGeneratorSerializer<F, D> for P25GeneratorSerializer<C, D>
where
    F: RichField + Extendable<D> + Poseidon2,      // Poseidon2Gate / Poseidon2Generator are bounded by the reference's trait
    C: GenericConfig<D, F = F> + 'static,
    C::Hasher: AlgebraicHasher<F>,
{
    impl_generator_serializer! {
        P25GeneratorSerializer,
        ArithmeticBaseGenerator<F, D>,        //  0
        ArithmeticExtensionGenerator<F, D>,   //  1
        BaseSplitGenerator<2>,                //  2
        BaseSumGenerator<2>,                  //  3
        ConstantGenerator<F>,                 //  4
        CopyGenerator,                        //  5
        DummyProofGenerator<F, C, D>,         //  6
        EqualityGenerator,                    //  7
        ExponentiationGenerator<F, D>,        //  8
        InterpolationGenerator<F, D>,         //  9
        LookupGenerator,                      // 10
        LookupTableGenerator,                 // 11
        LowHighGenerator,                     // 12
        MulExtensionGenerator<F, D>,          // 13
        NonzeroTestGenerator,                 // 14
        PoseidonGenerator<F, D>,              // 15
        PoseidonMdsGenerator<D>,              // 16
        QuotientGeneratorExtension<D>,        // 17
        RandomAccessGenerator<F, D>,          // 18
        RandomValueGenerator,                 // 19
        ReducingGenerator<D>,                 // 20
        ReducingExtensionGenerator<D>,        // 21
        SplitGenerator,                       // 22
        WireSplitGenerator,                   // 23
        Poseidon2Generator<F, D>,             // 24  the reference's: src/common/poseidon2/poseidon2_gate.rs:430-540
        U32ArithmeticGenerator<F, D>,         // 25  src/common/u32/gates/arithmetic_u32.rs:370-464
        U32InterleaveGenerator,               // 26  src/common/u32/gates/interleave_u32.rs:290-360
        UninterleaveToU32Generator            // 27  src/common/u32/gates/uninterleave_to_u32.rs:338-412
    }
}
