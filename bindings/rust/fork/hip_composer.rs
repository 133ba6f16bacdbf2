//! hip_composer.rs -- the reference's gadget API (dusk-network/plonk_gadgets `src/lib.rs:42-45`) on a device-resident
//! composer, as a module to ADD TO A FORK OF dusk-plonk 0.8 (`src/constraint_system/hip_composer.rs`, declared in
//! `src/constraint_system/mod.rs` as `pub mod hip_composer;`, with `ffi.rs` beside it).
//!
//! Why a fork and not an external crate: dusk-plonk 0.8 keeps `Variable`'s index crate-private
//! (`pub struct Variable(pub(crate) usize)`) and every column of `StandardComposer` `pub(crate)`; it exports no
//! constructor from an index and no accessor for one.  A shim that has to turn the `u64` the C ABI returns into a
//! `Variable` (and back) can only be written INSIDE the crate.  Everything below uses nothing but that: `Variable(i)`,
//! `v.0`, and `BlsScalar`'s public inner array (`dusk_bls12_381::BlsScalar(pub [u64; 4])`, the four Montgomery limbs
//! the library calls `pg_scalar`).  No accessor names are invented here.
//!
//! NOT compiled in the environment that produced it (no Rust toolchain there): source text for a maintainer.
//! Names, argument order, return shapes and the one error follow the reference:
//!   AllocatedScalar::allocate   src/allocated_scalar.rs:27        range_check   src/range.rs:27-32
//!   max_bound                   src/range.rs:82-86                maybe_equal   src/scalar.rs:105-109
//!   conditionally_select_zero   src/scalar.rs:21-25               is_non_zero   src/scalar.rs:63-67
//!   conditionally_select_one    src/scalar.rs:36-40
//! The reference crate (plonk_gadgets) then re-exports these behind a cargo feature, e.g.
//!   #[cfg(feature = "hip")] pub use dusk_plonk::constraint_system::hip_composer::{range_check, max_bound, ...};
#![allow(clippy::missing_safety_doc)]
use super::ffi::*;
use super::Variable;          // pub struct Variable(pub(crate) usize)  -- constructible here, inside the crate
use dusk_bls12_381::BlsScalar; // pub struct BlsScalar(pub [u64; 4])     -- Montgomery limbs

/// `plonk_gadgets::Error` has the single variant `NonExistingInverse` (src/errors.rs:13-18); it is mirrored here so
/// that this module does not depend on the crate that depends on it.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum HipError {
    NonExistingInverse,
}

#[inline]
fn to_pg(s: &BlsScalar) -> PgScalar {
    PgScalar { l: s.0 }
}
#[inline]
fn from_pg(s: &PgScalar) -> BlsScalar {
    BlsScalar(s.l)
}
fn status(st: i32, what: &str) -> Result<(), HipError> {
    match st {
        PG_OK => Ok(()),
        PG_ERR_NON_EXISTING_INVERSE => Err(HipError::NonExistingInverse),
        _ => panic!("{}: plonk_gadgets_hip status {}", what, st), // the reference panics where the library returns an error
    }
}

/// One per GPU and host thread.
pub struct Engine(*mut PgEngine);
impl Engine {
    pub fn new(device: i32) -> Self {
        let mut e = core::ptr::null_mut();
        status(unsafe { pg_engine_create(device, &mut e) }, "pg_engine_create").unwrap();
        Engine(e)
    }
}
impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { pg_engine_destroy(self.0) }
    }
}

/// Stands where the reference takes `&mut StandardComposer`: columns and variable table live in HBM.  Single calls are
/// recorded and flushed as few launches by the library (its command queue), so the reference's call-by-call style costs
/// about as much as a batched append.
pub struct HipComposer(*mut PgComposer);
impl HipComposer {
    /// `StandardComposer::new()`; appends double the capacity when they do not fit, like the reference's Vecs.
    pub fn new(engine: &Engine) -> Self {
        let mut c = core::ptr::null_mut();
        status(unsafe { pg_composer_create(engine.0, 1 << 16, 1 << 16, 1, core::ptr::null_mut(), &mut c) }, "pg_composer_create").unwrap();
        unsafe { pg_composer_auto_grow(c, 1) };
        HipComposer(c)
    }
    pub fn circuit_size(&self) -> usize {
        unsafe { pg_composer_circuit_size(self.0) as usize }
    }
    pub fn add_input(&mut self, s: BlsScalar) -> Variable {
        let mut v = 0u64;
        status(unsafe { pg_composer_add_input(self.0, &to_pg(&s), &mut v) }, "add_input").unwrap();
        Variable(v as usize)
    }
    pub fn constrain_to_constant(&mut self, a: Variable, constant: BlsScalar, pi: Option<BlsScalar>) {
        let p = pi.map(|x| to_pg(&x));
        let pp = p.as_ref().map_or(core::ptr::null(), |x| x as *const PgScalar);
        status(unsafe { pg_composer_constrain_to_constant(self.0, a.0 as u64, &to_pg(&constant), pp) }, "constrain_to_constant").unwrap();
    }
    /// -1 when every row satisfies its gate equation, else the first row that does not
    pub fn check(&mut self) -> i64 {
        let mut bad = 0i64;
        status(unsafe { pg_composer_check(self.0, &mut bad) }, "check").unwrap();
        bad
    }
    /// the loop `for w in witnesses { allocate; range_check }` as one append; `d_witness` / `d_result` are device pointers
    pub unsafe fn range_check_batch(&mut self, min_range: BlsScalar, max_range: BlsScalar, d_witness: *const PgScalar, batch: u64,
                                    d_result: *mut u64) {
        status(pg_composer_range_check_batch(self.0, &to_pg(&min_range), &to_pg(&max_range), d_witness, batch, d_result),
               "range_check_batch").unwrap();
    }
}
impl Drop for HipComposer {
    fn drop(&mut self) {
        unsafe { pg_composer_destroy(self.0) }
    }
}

/// src/allocated_scalar.rs:17-30
#[derive(Clone, Copy)]
pub struct AllocatedScalar {
    pub var: Variable,
    pub scalar: BlsScalar,
}
impl AllocatedScalar {
    pub fn allocate(composer: &mut HipComposer, scalar: BlsScalar) -> AllocatedScalar {
        let mut out = PgAllocatedScalar { var: 0, scalar: PgScalar::default() };
        status(unsafe { pg_allocated_scalar_allocate(composer.0, &to_pg(&scalar), &mut out) }, "allocate").unwrap();
        AllocatedScalar { var: Variable(out.var as usize), scalar: from_pg(&out.scalar) }
    }
    fn c(&self) -> PgAllocatedScalar {
        PgAllocatedScalar { var: self.var.0 as u64, scalar: to_pg(&self.scalar) }
    }
}

pub fn range_check(composer: &mut HipComposer, min_range: BlsScalar, max_range: BlsScalar, witness: AllocatedScalar) -> Variable {
    let mut out = 0u64;
    status(unsafe { pg_range_check(composer.0, &to_pg(&min_range), &to_pg(&max_range), &witness.c(), &mut out) }, "range_check").unwrap();
    Variable(out as usize)
}
pub fn max_bound(composer: &mut HipComposer, max_range: BlsScalar, witness: AllocatedScalar) -> (Variable, u64) {
    let (mut out, mut num_bits) = (0u64, 0u64);
    status(unsafe { pg_max_bound(composer.0, &to_pg(&max_range), &witness.c(), &mut out, &mut num_bits) }, "max_bound").unwrap();
    (Variable(out as usize), num_bits)
}
pub fn conditionally_select_zero(composer: &mut HipComposer, x: Variable, select: Variable) -> Variable {
    let mut out = 0u64;
    status(unsafe { pg_conditionally_select_zero(composer.0, x.0 as u64, select.0 as u64, &mut out) }, "conditionally_select_zero").unwrap();
    Variable(out as usize)
}
pub fn conditionally_select_one(composer: &mut HipComposer, y: Variable, selector: Variable) -> Variable {
    let mut out = 0u64;
    status(unsafe { pg_conditionally_select_one(composer.0, y.0 as u64, selector.0 as u64, &mut out) }, "conditionally_select_one").unwrap();
    Variable(out as usize)
}
pub fn is_non_zero(composer: &mut HipComposer, var: Variable, value_assigned: BlsScalar) -> Result<(), HipError> {
    status(unsafe { pg_is_non_zero(composer.0, var.0 as u64, &to_pg(&value_assigned)) }, "is_non_zero")
}
pub fn maybe_equal(composer: &mut HipComposer, a: AllocatedScalar, b: AllocatedScalar) -> Variable {
    let mut out = 0u64;
    status(unsafe { pg_maybe_equal(composer.0, &a.c(), &b.c(), &mut out) }, "maybe_equal").unwrap();
    Variable(out as usize)
}
