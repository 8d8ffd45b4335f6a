//! `impl Solver for HipSolver` — the device solvers of `liblocityper_hip.so` behind the crate's own solver trait
//! (src/solvers/mod.rs:22-92). Goes to `src/solvers/hip.rs`; registered in `Stage::parse` (shim/patches/solve_rs.patch).
//!
//! `solve_nontrivial` honours what it is handed: the `GenotypeAlignments` the caller built and tweaked with ITS generator
//! (`gt_alns.apply_tweak(rng, ..); stage.solver.solve(&gt_alns, rng)`, solve.rs:824-826) goes to the device as it is — locations,
//! windows after the tweak, window distributions, contributions — through `lcty_solve_given_tables`; one chain runs there; the
//! assignment comes back and the `ReadAssignment` is built from it by the crate's own constructor (`ReadAssignment::new`,
//! assgn.rs:191-226), so `likelihood()`, `update_counts()`, `write_depth()` and `summarize()` of the returned object are
//! computed on the caller's `gt_alns` by the crate's own code. No file of `src/model` changes but for one visibility
//! (`CachedDistr` becomes `pub(crate)`, model/distr_cache.rs:17).
//!
//! NOT COMPILED in the image this library is built in (no rustc / cargo; the crates of `Cargo.toml` are not vendored): the C side
//! of every call below is exercised by `tests/test_gpu_solve_given.py` of the library through the same entry points.
use std::{
    fmt,
    sync::{Arc, Mutex},
};
use rand::Rng;
use crate::{
    ext::{
        rand::XoshiroRng,
        fmt::PrettyUsize,
    },
    hip::{self, sys, HipCtx},
    math::distr::DiscretePmf,
    model::{
        assgn::{GenotypeAlignments, ReadAssignment},
        distr_cache::CachedDistr,
    },
};
use super::{Solver, SetParams, ParamErr};

/// Rows of `ln_pmf(depth)` of the window distributions seen so far (one `CachedDistr` per GC bin and locus, distr_cache.rs:61-75).
/// The `Arc`s are held, so an address cannot come back as another distribution while its row is here.
#[derive(Default)]
struct RowCache {
    rows: Vec<(CachedDistr, Vec<f64>)>,
}

impl RowCache {
    /// Values `0..width` of `distr`, evaluated once (LinearCache::ln_pmf, lincache.rs:41-48: the crate's own numbers).
    fn row(&mut self, distr: &CachedDistr, width: usize) -> &[f64] {
        let i = match self.rows.iter().position(|(d, _)| Arc::ptr_eq(d, distr)) {
            Some(i) => i,
            None => {
                if self.rows.len() >= 512 {
                    self.rows.clear();      // several loci later: start over
                }
                self.rows.push((Arc::clone(distr), Vec::new()));
                self.rows.len() - 1
            }
        };
        let values = &mut self.rows[i].1;
        for k in values.len()..width {
            values.push(distr.ln_pmf(k as u32));
        }
        &values[..width]
    }
}

#[derive(Clone)]
pub struct HipSolver {
    raw: sys::lcty_solver,
    ctx: Arc<HipCtx>,
    rows: Arc<Mutex<RowCache>>,
}

impl HipSolver {
    /// `Greedy::default()` (stoch.rs:45-53) on the device.
    pub fn greedy() -> crate::Result<Self> { Self::new(sys::LCTY_SOLVER_GREEDY) }
    /// `SimAnneal::default()` (stoch.rs:161-169) on the device.
    pub fn anneal() -> crate::Result<Self> { Self::new(sys::LCTY_SOLVER_ANNEAL) }
    /// In the place of `HighsSolver` / `GurobiSolver` (highs.rs:22-28): the same integer programme, proven or refused.
    pub fn exact() -> crate::Result<Self> { Self::new(sys::LCTY_SOLVER_EXACT) }

    fn new(kind: i32) -> crate::Result<Self> {
        let ctx = HipCtx::global()?;
        let mut raw = sys::lcty_solver { kind, best_start: 1, sample_size: 0, plato_size: 0, anneal_steps: 0, node_limit: 0, init_prob: 0.0 };
        hip::check(unsafe { sys::lcty_solver_default(&mut raw, kind) }, "Hip")?;
        Ok(Self { raw, ctx, rows: Arc::new(Mutex::new(RowCache::default())) })
    }

    fn name(&self) -> &'static str {
        match self.raw.kind {
            sys::LCTY_SOLVER_GREEDY => "HipGreedy",
            sys::LCTY_SOLVER_ANNEAL => "HipAnneal",
            _ => "HipExact",
        }
    }
}

impl SetParams for HipSolver {
    /// The keys, ranges and messages of `Greedy::set_param` / `SimAnneal::set_param` (stoch.rs:128-139, 249-260).
    fn set_param(&mut self, key: &str, val: &str) -> Result<(), ParamErr> {
        match (self.raw.kind, &key.to_lowercase() as &str) {
            (sys::LCTY_SOLVER_GREEDY, "x0" | "start") => self.raw.best_start = match val {
                "b" | "best" => 1,
                "r" | "rand" | "random" => 0,
                _ => return Err(ParamErr::Invalid(format!("Invalid start value {}", val))),
            },
            (sys::LCTY_SOLVER_GREEDY, "s" | "sample") => {
                let n = val.parse::<PrettyUsize>()?.0;
                if n == 0 {
                    return Err(ParamErr::Invalid("Sample size must be positive".to_string()));
                }
                self.raw.sample_size = n as u32;
            }
            (sys::LCTY_SOLVER_GREEDY | sys::LCTY_SOLVER_ANNEAL, "p" | "plato") => self.raw.plato_size = val.parse::<PrettyUsize>()?.0 as u32,
            (sys::LCTY_SOLVER_ANNEAL, "n" | "steps") => {
                let n = val.parse::<PrettyUsize>()?.0;
                if n == 0 {
                    return Err(ParamErr::Invalid(format!("Number of annealing steps ({}) must be positive", n)));
                }
                self.raw.anneal_steps = n as u32;
            }
            // ("P" is unreachable upstream too: the key is lower-cased first, stoch.rs:251-255)
            (sys::LCTY_SOLVER_ANNEAL, "prob" | "init-prob") => {
                let p: f64 = val.parse()?;
                if !(p > 0.0 && p <= 1.0) {
                    return Err(ParamErr::Invalid(format!("Initial probability ({}) must be within (0, 1]", p)));
                }
                self.raw.init_prob = p;
            }
            // `-S hip-exact:nodes=50m`: branch-and-bound nodes per attempt before `Error::Solver` (HighsSolver has "mode", highs.rs:30-36)
            (sys::LCTY_SOLVER_EXACT, "nodes" | "node-limit") => {
                let n = val.parse::<PrettyUsize>()?.0;
                if n == 0 || n > u32::MAX as usize {
                    return Err(ParamErr::Invalid(format!("Node limit ({}) must be within [1, 2^32)", n)));
                }
                self.raw.node_limit = n as u32;
            }
            // `-S hip-exact:gap=0`: HiGHS' mip_rel_gap (default 1e-4: highs.rs:103-110 leaves the option alone); 0 = a proof
            (sys::LCTY_SOLVER_EXACT, "gap" | "rel-gap") => {
                let g: f64 = val.parse()?;
                if !(g >= 0.0 && g < 1.0) {
                    return Err(ParamErr::Invalid(format!("Relative gap ({}) must be within [0, 1)", g)));
                }
                self.raw.init_prob = g;
            }
            _ => return Err(ParamErr::Unknown),
        }
        Ok(())
    }
}

impl Solver for HipSolver {
    fn solve_nontrivial<'a>(
        &self,
        gt_alns: &'a GenotypeAlignments,
        rng: &mut XoshiroRng,
    ) -> crate::Result<ReadAssignment<'a>>
    {
        // The object as arrays (assgn.rs:16-36): locations of every read pair with the windows apply_tweak left them ...
        let n_reads = gt_alns.total_reads();
        let n_alns = gt_alns.total_possible_alns();
        let mut read_ixs = Vec::with_capacity(n_reads + 1);
        let mut ln_prob = Vec::with_capacity(n_alns);
        let mut windows = Vec::with_capacity(2 * n_alns);
        read_ixs.push(0_u64);
        for rp in 0..n_reads {
            for aln in gt_alns.possible_read_alns(rp) {
                ln_prob.push(aln.ln_prob());
                windows.extend_from_slice(&aln.windows());
            }
            read_ixs.push(ln_prob.len() as u64);
        }
        // ... and the distribution of every window: its weight, and which of the (few) cached distributions it points to.
        let n_windows = gt_alns.total_windows();
        let mut distrs: Vec<&CachedDistr> = Vec::new();
        let mut window_row = vec![0_u8; n_windows];
        let mut window_weight = vec![0.0_f64; n_windows];      // 0 = WindowDistr::TRIVIAL
        for w in 0..n_windows {
            let distr = gt_alns.depth_distr(w);
            if let Some(inner) = distr.inner() {
                let row = match distrs.iter().position(|d| Arc::ptr_eq(d, inner)) {
                    Some(i) => i,
                    None => { distrs.push(inner); distrs.len() - 1 }
                };
                window_row[w] = row as u8;      // at most GC_BINS = 101 of them (bg/depth.rs:42)
                window_weight[w] = distr.weight();
            }
        }
        let gt_windows = gt_alns.gt_windows();
        let ploidy = gt_windows.genotype().ploidy();
        let wshifts: Vec<u32> = (0..=ploidy).map(|i| gt_windows.get_wshift(i)).collect();
        let (depth_contrib, aln_contrib) = gt_alns.contributions();
        let view = sys::lcty_gt_alns_view {
            n_reads: n_reads as u64,
            read_ixs: read_ixs.as_ptr(), ln_prob: ln_prob.as_ptr(), windows: windows.as_ptr(),
            n_windows: n_windows as u32, n_contigs: ploidy.min(16) as u32,
            window_gc: window_row.as_ptr(), window_weight: window_weight.as_ptr(),
            wshifts: wshifts.as_ptr(),
            depth_contrib, aln_contrib,
        };

        // Rows of ln_pmf as deep as a window of this object can get, from the crate's own LinearCache / BayesCalc.
        let mut deepest = 0_u32;
        hip::check(unsafe { sys::lcty_gt_alns_deepest(&view, &mut deepest) }, self.name())?;
        let width = ((deepest as usize + 1 + 255) / 256) * 256;
        let n_rows = distrs.len().max(1);
        let mut values = vec![0.0_f64; n_rows * width];
        {
            let mut cache = self.rows.lock().unwrap();
            for (i, distr) in distrs.iter().enumerate() {
                values[i * width..(i + 1) * width].copy_from_slice(cache.row(distr, width));
            }
        }
        let tables = sys::lcty_depth_tables { n_rows: n_rows as u32, width: width as u32, values: values.as_ptr(), id: 0 };

        // rand_xoshiro keeps the four words of its generator private: the call gets a stream seeded by ONE draw of the caller's
        // generator (which is also all the library would take from the words themselves: one next_u64, the chain's seed).
        let mut state = [0_u64; 4];
        hip::check(unsafe { sys::lcty_rng_seed_from_u64(rng.next_u64(), state.as_mut_ptr()) }, self.name())?;
        let mut read_assgn = vec![0_u16; n_reads];
        hip::check(unsafe {
            sys::lcty_solve_given_tables(self.ctx.ptr(), &view, &tables, &self.raw, state.as_mut_ptr(),
                read_assgn.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut())
        }, self.name())?;

        // The crate's own constructor: `select_init` is asked once per non-trivial read pair, in read order (assgn.rs:204-219).
        let mut non_trivial = gt_alns.non_trivial_reads().iter();
        Ok(ReadAssignment::new(gt_alns, |_| usize::from(read_assgn[*non_trivial.next().expect("one call per non-trivial read")])))
    }

    fn describe_params(&self) -> String {
        match self.raw.kind {
            sys::LCTY_SOLVER_GREEDY => format!("x0={},s={},p={}", if self.raw.best_start != 0 { "best" } else { "random" },
                PrettyUsize(self.raw.sample_size as usize), PrettyUsize(self.raw.plato_size as usize)),
            sys::LCTY_SOLVER_ANNEAL => format!("n={},p={},P={:.3}", PrettyUsize(self.raw.anneal_steps as usize),
                PrettyUsize(self.raw.plato_size as usize), self.raw.init_prob),
            _ => format!("nodes={},gap={:e}", PrettyUsize(if self.raw.node_limit == 0 { 20_000_000 } else { self.raw.node_limit as usize }),
                self.raw.init_prob),
        }
    }
}

impl fmt::Display for HipSolver {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        match self.raw.kind {
            sys::LCTY_SOLVER_GREEDY => write!(f, "Stochastic greedy (HIP)"),
            sys::LCTY_SOLVER_ANNEAL => write!(f, "Simulated annealing (HIP)"),
            _ => write!(f, "Exact (HIP)"),
        }
    }
}
