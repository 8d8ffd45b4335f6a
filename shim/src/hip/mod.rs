//! The device side of `locityper genotype`: `liblocityper_hip.so` behind safe wrappers. Goes to `src/hip/mod.rs`
//! (`mod hip;` in `src/main.rs`, behind `#[cfg(feature = "hip")]`).
//! NOT COMPILED in the image this library is built in: see `shim/README.md`.
pub mod sys;

use std::{
    ffi::CStr,
    ptr,
    sync::{Arc, OnceLock},
};
use crate::err::{Error, error};

/// Turns a status of the library into the crate's error (src/err.rs:11-30): same categories, the library's message.
pub(crate) fn check(status: i32, solver_name: &'static str) -> crate::Result<()> {
    if status == sys::LCTY_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(sys::lcty_last_error()) }.to_string_lossy().into_owned();
    Err(match status {
        sys::LCTY_ERR_INVALID_INPUT => error!(InvalidInput, "{}", msg),
        sys::LCTY_ERR_INVALID_DATA => error!(InvalidData, "{}", msg),
        sys::LCTY_ERR_SOLVER => Error::solver(solver_name, msg),
        // UNSUPPORTED: a shape the device kernels do not take (more than 255 locations of a read pair, ...) — a run-time failure here
        _ => error!(RuntimeError, "{}", msg),
    })
}

/// One GPU (`lcty_ctx`): streams and the per-call chain workspaces. Shared by every `HipSolver` of the process.
pub struct HipCtx(*mut sys::lcty_ctx);

// The library's contract: distinct calls on one context may run on distinct threads at once (include/locityper_hip.h, lcty_solve_given).
unsafe impl Send for HipCtx {}
unsafe impl Sync for HipCtx {}

impl HipCtx {
    /// The context of this process: created at the first `-S hip-*` stage, on the device `LOCITYPER_HIP_DEVICE` names (default 0).
    pub fn global() -> crate::Result<Arc<HipCtx>> {
        static CTX: OnceLock<Result<Arc<HipCtx>, String>> = OnceLock::new();
        CTX.get_or_init(|| {
            let device = std::env::var("LOCITYPER_HIP_DEVICE").ok().and_then(|s| s.parse::<i32>().ok()).unwrap_or(0);
            let mut raw = ptr::null_mut();
            match check(unsafe { sys::lcty_ctx_create(device, &mut raw) }, "Hip") {
                Ok(()) => Ok(Arc::new(HipCtx(raw))),
                Err(e) => Err(e.display()),
            }
        }).clone().map_err(|msg| error!(RuntimeError, "{}", msg))
    }

    pub(crate) fn ptr(&self) -> *mut sys::lcty_ctx {
        self.0
    }
}

impl Drop for HipCtx {
    fn drop(&mut self) {
        unsafe { sys::lcty_ctx_destroy(self.0) }
    }
}
