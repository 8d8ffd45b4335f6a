//! Raw bindings of `liblocityper_hip.so` (`include/locityper_hip.h`) — the subset the solver shim calls.
//!
//! Goes to `src/hip/sys.rs` of the reference crate. Written by hand instead of `bindgen` (the reference uses bindgen for WFA2 only,
//! `build.rs:9-24`): the structs below mirror the header field for field; `tests/test_host_api.py` of the library checks that
//! every symbol the header declares is exported.
//! NOT COMPILED in the image this library is built in (no rustc / cargo there): see `shim/README.md`.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const LCTY_OK: i32 = 0;
pub const LCTY_ERR_INVALID_INPUT: i32 = 1;
pub const LCTY_ERR_INVALID_DATA: i32 = 2;
pub const LCTY_ERR_RUNTIME: i32 = 3;
pub const LCTY_ERR_SOLVER: i32 = 4;
pub const LCTY_ERR_UNSUPPORTED: i32 = 5;

pub const LCTY_SOLVER_GREEDY: i32 = 0;
pub const LCTY_SOLVER_ANNEAL: i32 = 1;
pub const LCTY_SOLVER_EXACT: i32 = 2;

/// Opaque handles.
#[repr(C)] pub struct lcty_ctx { _private: [u8; 0] }
#[repr(C)] pub struct lcty_locus { _private: [u8; 0] }

/// `lcty_solver`: the parameters of `Greedy` (stoch.rs:36-43), `SimAnneal` (stoch.rs:151-160) and of the exact solver.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct lcty_solver {
    pub kind: i32,
    pub best_start: i32,
    pub sample_size: u32,
    pub plato_size: u32,
    pub anneal_steps: u32,
    pub node_limit: u32,
    pub init_prob: f64,
}

/// `lcty_gt_alns_view`: a `GenotypeAlignments` (model/assgn.rs:16-36) after `apply_tweak` as plain arrays.
#[repr(C)]
pub struct lcty_gt_alns_view {
    pub n_reads: u64,
    pub read_ixs: *const u64,
    pub ln_prob: *const f64,
    pub windows: *const u32,
    pub n_windows: u32,
    pub n_contigs: u32,
    pub window_gc: *const u8,
    pub window_weight: *const f64,
    pub wshifts: *const u32,
    pub depth_contrib: f64,
    pub aln_contrib: f64,
}

/// `lcty_depth_tables`: `values[n_rows][width]` = `ln_pmf(depth)` of the caller's own window distributions.
#[repr(C)]
pub struct lcty_depth_tables {
    pub n_rows: u32,
    pub width: u32,
    pub values: *const f64,
    pub id: u64,
}

extern "C" {
    pub fn lcty_last_error() -> *const c_char;
    pub fn lcty_device_count() -> i32;
    pub fn lcty_ctx_create(device_id: i32, out: *mut *mut lcty_ctx) -> i32;
    pub fn lcty_ctx_destroy(ctx: *mut lcty_ctx);
    pub fn lcty_solver_default(out: *mut lcty_solver, kind: i32) -> i32;
    pub fn lcty_gt_alns_deepest(gt_alns: *const lcty_gt_alns_view, deepest: *mut u32) -> i32;
    pub fn lcty_solve_given_tables(ctx: *mut lcty_ctx, gt_alns: *const lcty_gt_alns_view, tables: *const lcty_depth_tables,
        solver: *const lcty_solver, rng_state: *mut u64, read_assgn: *mut u16, lik_parts: *mut f64, likelihood: *mut f64) -> i32;
    pub fn lcty_solve_given(locus: *mut lcty_locus, gt_alns: *const lcty_gt_alns_view, solver: *const lcty_solver,
        rng_state: *mut u64, read_assgn: *mut u16, lik_parts: *mut f64, likelihood: *mut f64) -> i32;
    pub fn lcty_rng_seed_from_u64(seed: u64, state: *mut u64) -> i32;
    pub fn lcty_rng_next_u64(state: *mut u64, out: *mut u64) -> i32;
    pub fn lcty_host_alloc(ctx: *mut lcty_ctx, bytes: u64, out: *mut *mut c_void) -> i32;
    pub fn lcty_host_free(p: *mut c_void);
}
