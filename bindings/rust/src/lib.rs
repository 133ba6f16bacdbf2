//! The reference's gadget API (dusk-network/plonk_gadgets `src/lib.rs:42-45`) on a device-resident composer.
//!
//! NOT compiled in the environment that produced it (no Rust toolchain there): source text for a maintainer, kept
//! next to the generated `ffi.rs`.  Names, argument order, return shapes and the one error follow the reference:
//!   AllocatedScalar::allocate   src/allocated_scalar.rs:27
//!   range_check                 src/range.rs:27-32
//!   max_bound                   src/range.rs:82-86
//!   conditionally_select_zero   src/scalar.rs:21-25
//!   conditionally_select_one    src/scalar.rs:36-40
//!   is_non_zero                 src/scalar.rs:63-67
//!   maybe_equal                 src/scalar.rs:105-109
#![allow(clippy::missing_safety_doc)]
pub mod ffi;

use dusk_plonk::prelude::{BlsScalar, Variable};
use ffi::*;
use plonk_gadgets::Error;

fn to_pg(s: &BlsScalar) -> PgScalar {
    PgScalar { l: *s.internal_repr() } // four Montgomery limbs, the library's own representation
}
fn from_pg(s: &PgScalar) -> BlsScalar {
    BlsScalar::from_raw_unchecked(s.l) // hypothetical constructor name: "these limbs are already Montgomery form"
}
fn status(st: i32, what: &str) -> Result<(), Error> {
    match st {
        PG_OK => Ok(()),
        PG_ERR_NON_EXISTING_INVERSE => Err(Error::NonExistingInverse),
        _ => panic!("{}: plonk_gadgets_hip status {}", what, st), // the reference panics where this returns an error
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

/// Stands where the reference takes `&mut StandardComposer`: columns and variable table live in HBM.
pub struct StandardComposer(*mut PgComposer);
impl StandardComposer {
    /// `StandardComposer::new()`; appends double the capacity when they do not fit, like the reference's Vecs.
    pub fn new(engine: &Engine) -> Self {
        let mut c = core::ptr::null_mut();
        status(unsafe { pg_composer_create(engine.0, 1 << 16, 1 << 16, 1, core::ptr::null_mut(), &mut c) }, "pg_composer_create").unwrap();
        unsafe { pg_composer_auto_grow(c, 1) };
        StandardComposer(c)
    }
    pub fn circuit_size(&self) -> usize {
        unsafe { pg_composer_circuit_size(self.0) as usize }
    }
    pub fn add_input(&mut self, s: BlsScalar) -> Variable {
        let mut v = 0u64;
        status(unsafe { pg_composer_add_input(self.0, &to_pg(&s), &mut v) }, "add_input").unwrap();
        Variable::new(v as usize)
    }
    pub fn constrain_to_constant(&mut self, a: Variable, constant: BlsScalar, pi: Option<BlsScalar>) {
        let p = pi.map(|x| to_pg(&x));
        let pp = p.as_ref().map_or(core::ptr::null(), |x| x as *const PgScalar);
        status(unsafe { pg_composer_constrain_to_constant(self.0, a.index() as u64, &to_pg(&constant), pp) }, "constrain_to_constant").unwrap();
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
impl Drop for StandardComposer {
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
    pub fn allocate(composer: &mut StandardComposer, scalar: BlsScalar) -> AllocatedScalar {
        let mut out = PgAllocatedScalar { var: 0, scalar: PgScalar::default() };
        status(unsafe { pg_allocated_scalar_allocate(composer.0, &to_pg(&scalar), &mut out) }, "allocate").unwrap();
        AllocatedScalar { var: Variable::new(out.var as usize), scalar: from_pg(&out.scalar) }
    }
    fn c(&self) -> PgAllocatedScalar {
        PgAllocatedScalar { var: self.var.index() as u64, scalar: to_pg(&self.scalar) }
    }
}

pub fn range_check(composer: &mut StandardComposer, min_range: BlsScalar, max_range: BlsScalar, witness: AllocatedScalar) -> Variable {
    let mut out = 0u64;
    status(unsafe { pg_range_check(composer.0, &to_pg(&min_range), &to_pg(&max_range), &witness.c(), &mut out) }, "range_check").unwrap();
    Variable::new(out as usize)
}
pub fn max_bound(composer: &mut StandardComposer, max_range: BlsScalar, witness: AllocatedScalar) -> (Variable, u64) {
    let (mut out, mut num_bits) = (0u64, 0u64);
    status(unsafe { pg_max_bound(composer.0, &to_pg(&max_range), &witness.c(), &mut out, &mut num_bits) }, "max_bound").unwrap();
    (Variable::new(out as usize), num_bits)
}
pub fn conditionally_select_zero(composer: &mut StandardComposer, x: Variable, select: Variable) -> Variable {
    let mut out = 0u64;
    status(unsafe { pg_conditionally_select_zero(composer.0, x.index() as u64, select.index() as u64, &mut out) }, "conditionally_select_zero").unwrap();
    Variable::new(out as usize)
}
pub fn conditionally_select_one(composer: &mut StandardComposer, y: Variable, selector: Variable) -> Variable {
    let mut out = 0u64;
    status(unsafe { pg_conditionally_select_one(composer.0, y.index() as u64, selector.index() as u64, &mut out) }, "conditionally_select_one").unwrap();
    Variable::new(out as usize)
}
pub fn is_non_zero(composer: &mut StandardComposer, var: Variable, value_assigned: BlsScalar) -> Result<(), Error> {
    status(unsafe { pg_is_non_zero(composer.0, var.index() as u64, &to_pg(&value_assigned)) }, "is_non_zero")
}
pub fn maybe_equal(composer: &mut StandardComposer, a: AllocatedScalar, b: AllocatedScalar) -> Variable {
    let mut out = 0u64;
    status(unsafe { pg_maybe_equal(composer.0, &a.c(), &b.c(), &mut out) }, "maybe_equal").unwrap();
    Variable::new(out as usize)
}
