//! dump_columns.rs -- what `tests/golden/compare_dump.py` of the MI355X repository reads: every array a
//! `StandardComposer` holds after gate emission, as raw little-endian u64 files.  A module to ADD TO A FORK OF
//! dusk-plonk 0.8 (`src/constraint_system/dump_columns.rs`, `pub mod dump_columns;` in `constraint_system/mod.rs`): the
//! columns, the variable map and the permutation are `pub(crate)` in that crate, so only code inside it can read them.
//! Field names are those of dusk-plonk 0.8's `constraint_system/composer.rs` and `permutation/mod.rs`
//! ([DEP-RECALL] in the MI355X repository's SURVEY.md: if one does not exist under this name the compiler says so, and the
//! mismatch is itself a finding to report).  Needs `std` (the reference's `std` feature forwards to dusk-plonk/std).
//!
//! NOT compiled in the environment that produced it (no Rust toolchain there).
//!
//! Layout written to `<dir>/`:
//!   q_m q_l q_r q_o q_c q_4 q_arith   .u64   n x 4   Montgomery limbs of each row's selector
//!   w_l w_r w_o w_4                   .u64   n       Variable indices
//!   var_values                        .u64   V x 4   assignment of Variable(i) at row i (indices are dense: 0..V)
//!   dense_pi                          .u64   n x 4   construct_dense_pi_vec()
//!   sigma                             .u64   4 x padded_n   position (wire, gate) -> wire' * padded_n + gate'
//!   meta                              .u64   [n, V, zero_var, padded_n]
use super::composer::StandardComposer;
use super::Variable;
use crate::permutation::WireData;
use dusk_bls12_381::BlsScalar;
use std::fs::{create_dir_all, File};
use std::io::Write;
use std::path::Path;

fn put(dir: &Path, name: &str, words: &[u64]) {
    let mut f = File::create(dir.join(format!("{}.u64", name))).expect("create dump file");
    for w in words {
        f.write_all(&w.to_le_bytes()).expect("write dump file");
    }
}
fn limbs(col: &[BlsScalar]) -> Vec<u64> {
    col.iter().flat_map(|s| s.0.iter().copied()).collect()
}
fn indices(col: &[Variable]) -> Vec<u64> {
    col.iter().map(|v| v.0 as u64).collect()
}

impl StandardComposer {
    /// Writes the composer's state as it is after gate emission (before `preprocess` pads anything).
    pub fn dump_columns(&self, dir: &Path) {
        create_dir_all(dir).expect("create dump directory");
        let n = self.n;
        for (name, col) in [("q_m", &self.q_m), ("q_l", &self.q_l), ("q_r", &self.q_r), ("q_o", &self.q_o), ("q_c", &self.q_c),
                            ("q_4", &self.q_4), ("q_arith", &self.q_arith)] {
            assert_eq!(col.len(), n, "{} has {} rows, circuit_size is {}", name, col.len(), n);
            put(dir, name, &limbs(col));
        }
        for (name, col) in [("w_l", &self.w_l), ("w_r", &self.w_r), ("w_o", &self.w_o), ("w_4", &self.w_4)] {
            put(dir, name, &indices(col));
        }
        // variables: HashMap<Variable, BlsScalar>; indices are handed out sequentially from 0
        let nv = self.variables.len();
        let mut vals = vec![0u64; 4 * nv];
        for (var, s) in self.variables.iter() {
            assert!(var.0 < nv, "Variable({}) outside 0..{}: numbering is not dense", var.0, nv);
            vals[4 * var.0..4 * var.0 + 4].copy_from_slice(&s.0);
        }
        put(dir, "var_values", &vals);
        put(dir, "dense_pi", &limbs(&self.construct_dense_pi_vec()));
        // sigma as positions: Permutation::compute_sigma_permutations(n) -> [Vec<WireData>; 4]
        let padded_n = n.next_power_of_two();
        let sigmas = self.perm.compute_sigma_permutations(padded_n);
        let mut sigma = Vec::with_capacity(4 * padded_n);
        for col in sigmas.iter() {
            for w in col.iter() {
                sigma.push(match *w {
                    WireData::Left(i) => i as u64,
                    WireData::Right(i) => (padded_n + i) as u64,
                    WireData::Output(i) => (2 * padded_n + i) as u64,
                    WireData::Fourth(i) => (3 * padded_n + i) as u64,
                });
            }
        }
        put(dir, "sigma", &sigma);
        put(dir, "meta", &[n as u64, nv as u64, self.zero_var.0 as u64, padded_n as u64]);
    }
}
