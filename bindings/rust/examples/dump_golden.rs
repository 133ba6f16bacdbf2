//! dump_golden.rs -- builds, on the REAL `StandardComposer`, the circuits the MI355X repository freezes under
//! tests/golden/*.npz (inputs: tests/refcases.py, i.e. the reference's own test cases, and tests/golden/make_golden.py)
//! and dumps every array with `StandardComposer::dump_columns` (bindings/rust/fork/dump_columns.rs, added to a fork of
//! dusk-plonk 0.8 -- the columns are `pub(crate)` there).  Drop this file into `examples/` of a checkout of
//! dusk-network/plonk_gadgets v0.6.0 whose Cargo.toml points `dusk-plonk` at that fork (`[patch.crates-io]`), then
//!     cargo run --release --features std --example dump_golden -- /tmp/plonk_gadgets_dump
//!     python tests/golden/compare_dump.py /tmp/plonk_gadgets_dump          # in the MI355X repository
//! A clean comparison turns "parity unpinned" into pinned; a difference names the first array/row/limb and the
//! recollection of dusk-plonk it falsifies.
//!
//! NOT compiled in the environment that produced it (no Rust toolchain there).
use dusk_plonk::prelude::*;
use plonk_gadgets::{AllocatedScalar, RangeGadgets, ScalarGadgets};
use std::path::PathBuf;

fn s(x: u128) -> BlsScalar {
    BlsScalar::from_raw([x as u64, (x >> 64) as u64, 0, 0]) // from_raw: canonical limbs -> Montgomery form
}
fn pow2(k: u64) -> BlsScalar {
    BlsScalar::pow_of_2(k)
}
fn q_minus(k: u64) -> BlsScalar {
    -BlsScalar::from(k)
}

fn range_cases(dir: &PathBuf, name: &str, min: BlsScalar, max: BlsScalar, witnesses: &[BlsScalar]) {
    let mut c = StandardComposer::new();
    for w in witnesses {
        let a = AllocatedScalar::allocate(&mut c, *w);
        RangeGadgets::range_check(&mut c, min, max, a);
    }
    c.dump_columns(&dir.join(name));
}

fn main() {
    let dir = PathBuf::from(std::env::args().nth(1).expect("usage: dump_golden <output directory>"));

    // tests/refcases.py RANGE_CHECK_CASES (tests/range_gadgets_tests.rs:120-169): the seven [50 000, 250 000) cases ...
    let ws: Vec<BlsScalar> = [50_001u128, 250_001, 250_000, 249_000, 50_000, 49_999, 18_598].iter().map(|&w| s(w)).collect();
    range_cases(&dir, "range_check_ref_50k_250k", s(50_000), s(250_000), &ws);
    // ... and case 7: [2^126, 2^127 + 1), witness 2^127 - 1
    range_cases(&dir, "range_check_ref_2p126_2p127", pow2(126), pow2(127) + BlsScalar::one(), &[pow2(127) - BlsScalar::one()]);
    // BASELINE shapes C1 (n = 65) and C2 (n = 255), three witnesses each
    range_cases(&dir, "range_check_c1_n65", BlsScalar::zero(), pow2(64), &[BlsScalar::zero(), pow2(64) - BlsScalar::one(), pow2(64) + pow2(59)]);
    range_cases(&dir, "range_check_c2_n255", BlsScalar::zero(), pow2(254), &[s(5), pow2(254) - BlsScalar::one(), q_minus(1)]);

    // MAX_BOUND_CASES (tests/range_gadgets_tests.rs:57-78)
    {
        let mut c = StandardComposer::new();
        for (max, w) in [(pow2(128) - BlsScalar::one(), pow2(127)), (s(200), s(100)), (s(100), s(200)), (pow2(128) - BlsScalar::one(), pow2(130))] {
            let a = AllocatedScalar::allocate(&mut c, w);
            RangeGadgets::max_bound(&mut c, max, a);
        }
        c.dump_columns(&dir.join("max_bound_ref"));
    }
    // MAYBE_EQUAL_CASES (tests/scalar_gadgets_tests.rs:36,53 + the verifier's (0, 0))
    {
        let mut c = StandardComposer::new();
        for (a, b) in [(100u128, 100u128), (20, 3330), (0, 0)] {
            let (aa, bb) = (AllocatedScalar::allocate(&mut c, s(a)), AllocatedScalar::allocate(&mut c, s(b)));
            ScalarGadgets::maybe_equal(&mut c, aa, bb);
        }
        c.dump_columns(&dir.join("maybe_equal_ref"));
    }
    // the fused mix of BASELINE config C3 (tests/golden/make_golden.py section 4): v, y, s, a, b per item; item 2 has v = 0
    {
        let mut c = StandardComposer::new();
        let items = [
            (s(7), s(1_234_567), s(1), s(100), s(100)),
            (q_minus(1), s(42), s(0), s(20), s(3330)),
            (s(0), s(5), s(1), s(0), s(0)),
            (pow2(200) + BlsScalar::one(), q_minus(5), s(1), s(9), q_minus(9)),
        ];
        for (v, y, sel, a, b) in items {
            let (vv, yv, sv) = (c.add_input(v), c.add_input(y), c.add_input(sel));
            let (aa, bb) = (AllocatedScalar::allocate(&mut c, a), AllocatedScalar::allocate(&mut c, b));
            let _ = ScalarGadgets::is_non_zero(&mut c, vv, v); // Err(NonExistingInverse) for item 2, after one variable + one row
            ScalarGadgets::conditionally_select_one(&mut c, yv, sv);
            ScalarGadgets::maybe_equal(&mut c, aa, bb);
        }
        c.dump_columns(&dir.join("scalar_mix"));
    }
    // one whole composer: tests/refcases.py full_circuit, call for call
    {
        let mut c = StandardComposer::new();
        let x = c.add_input(s(9));
        let mut r = Vec::new();
        for w in [50_001u128, 250_000, 7] {
            let a = AllocatedScalar::allocate(&mut c, s(w));
            r.push(RangeGadgets::range_check(&mut c, s(50_000), s(250_000), a));
        }
        ScalarGadgets::is_non_zero(&mut c, x, s(9)).unwrap();
        let y = ScalarGadgets::conditionally_select_one(&mut c, x, r[0]);
        c.constrain_to_constant(y, s(20), Some(s(11)));
        let (a5, b5) = (AllocatedScalar::allocate(&mut c, s(5)), AllocatedScalar::allocate(&mut c, s(5)));
        let m = ScalarGadgets::maybe_equal(&mut c, a5, b5);
        let z = ScalarGadgets::conditionally_select_zero(&mut c, m, r[1]);
        c.constrain_to_constant(z, BlsScalar::zero(), None);
        let big = AllocatedScalar::allocate(&mut c, pow2(64) + s(5));
        let (b, _) = RangeGadgets::max_bound(&mut c, pow2(64), big);
        c.constrain_to_constant(b, BlsScalar::zero(), None);
        c.boolean_gate(m);
        c.constrain_to_constant(r[0], BlsScalar::one(), None);
        c.dump_columns(&dir.join("composer_full"));
    }
    println!("dumped 8 circuits under {}", dir.display());
}
