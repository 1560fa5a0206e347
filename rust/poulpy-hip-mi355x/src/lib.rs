//! `FFT64Hip`: a poulpy-hal backend whose FFT64 family runs on an AMD MI355X (gfx950) through libpoulpy_hip.so.
//!
//! Shape of the crate — the same as poulpy-cpu-avx (poulpy-cpu-avx/src/lib.rs, src/hal_impl.rs):
//!   * [`FFT64Hip`]: the backend marker; `impl Backend` with `ScalarPrep = f64`, `ScalarBig = i64`, pinned host buffers, and a handle
//!     that owns the C module (device twiddle tables, stream, workspaces);
//!   * `znx.rs`: the `Znx*` single-polynomial traits, delegated to poulpy-cpu-ref's reference kernels — they are what the portable
//!     defaults of the pure-i64 families are written against (poulpy-cpu-ref/src/fft64/znx.rs does the same for `FFT64Ref`);
//!   * `hal_impl.rs`: `unsafe impl HalImpl<FFT64Hip>` — all 105 required methods: DFT-domain families forward to the C ABI;
//!   * `core_impl.rs` (feature `core-fused`, default): `unsafe impl CoreImpl<FFT64Hip>` with the GLWE-level key switch / external
//!     product / automorphism family on the fused three-kernel device pipeline;
//!   * `batched.rs` (same feature): `HipBatched`, an ADDITIVE extension trait on `Module<FFT64Hip>` over device buffers - every
//!     `pz_*_batched` entry point of the C ABI without `unsafe` in the caller (external products, key switches, the automorphism family,
//!     trace, tensoring + relinearization, blind rotation, circuit bootstrapping, the LWE glue, the per-op batched HAL);
//!   * `tests.rs`: the reference's own `cross_backend_test_suite!` against `FFT64Ref`, as poulpy-cpu-avx/src/fft64/tests.rs.
//!
//! Bytes of `VecZnxDft` / `SvpPPol` / `VmpPMat` / `CnvPVec*` are backend-private ("device order": natural-frequency, interleaved
//! re/im); callers treat them as opaque, which poulpy-core does.  `i64` containers have the reference's layout.
mod ffi;
mod hal_impl;
mod znx;

#[cfg(feature = "core-fused")]
mod core_impl;

/// The additive batched, device-resident API (`HipBatched` on `Module<FFT64Hip>`): the route to this backend's throughput and to the
/// batched blind rotation / circuit bootstrapping, which poulpy-bin-fhe's blanket impls do not let a backend override (batched.rs).
#[cfg(feature = "core-fused")]
pub mod batched;
#[cfg(feature = "core-fused")]
pub use batched::{AutomorphismMode, DeviceBuf, DeviceVecZnx, DeviceVecZnxDft, DeviceVmpPMat, HipBatched, HipBatchedCore, TensorMode};

#[cfg(test)]
mod tests;

use std::ptr::NonNull;

use poulpy_hal::layouts::Backend;

pub use ffi::{pz_blind_rotation_params, pz_circuit_bootstrapping_params, pz_glwe_op_params, pz_glwe_tensor_params};

/// Backend marker (`Module<FFT64Hip>`).
#[derive(Debug, Clone, Copy)]
pub struct FFT64Hip;

/// ABI generation of include/poulpy_hip.h this crate was generated against (`pz_abi_version()`); checked BEFORE the first call into the
/// library (`check_abi`, from `Module::new`): a stale or variant libpoulpy_hip.so would otherwise be handed structs of another layout.
pub const PZ_ABI_VERSION: u32 = 4;

/// Panics unless the loaded libpoulpy_hip.so is the generation this crate was generated for.  `pz_abi_version` takes no argument and
/// allocates nothing, so a mismatch leaves nothing behind (the check used to run after `pz_module_new`, leaking the module).
pub(crate) fn check_abi() {
    static ONCE: std::sync::Once = std::sync::Once::new();
    ONCE.call_once(|| {
        let v = unsafe { ffi::pz_abi_version() };
        assert_eq!(v, PZ_ABI_VERSION, "libpoulpy_hip.so has ABI version {v}, this crate was generated for {PZ_ABI_VERSION}");
    });
}

/// The sibling modules (`pz_module_clone`: own stream, workspaces and lock; shared immutable tables) that threads other than the
/// creating one run on.  A thread LEASES one on its first call and returns it when it exits (thread-local destructor), so the pool
/// is bounded by the largest number of threads that were alive at the same time — poulpy-bin-fhe spawns scoped threads on every
/// evaluation (bdd_arithmetic/eval.rs:210-221) and thread ids are never reused: a map keyed by thread id would grow, each sibling
/// holding a stream and grow-only workspaces, until the device runs out of memory.
struct SiblingPool {
    free: std::sync::Mutex<Vec<usize>>, // siblings no live thread holds
    all: std::sync::Mutex<Vec<usize>>,  // every sibling ever cloned: freed with the handle
    closed: std::sync::atomic::AtomicBool,
}

/// One thread's hold on a sibling of one handle; dropped by the thread-local destructor when the thread exits.
struct Lease {
    pool: std::sync::Arc<SiblingPool>,
    handle_id: u64,
    module: usize,
}
impl Drop for Lease {
    fn drop(&mut self) {
        if !self.pool.closed.load(std::sync::atomic::Ordering::Acquire) {
            self.pool.free.lock().unwrap().push(self.module);
        }
    }
}
thread_local! {
    static LEASES: std::cell::RefCell<Vec<Lease>> = const { std::cell::RefCell::new(Vec::new()) };
}
static NEXT_HANDLE_ID: std::sync::atomic::AtomicU64 = std::sync::atomic::AtomicU64::new(1);

/// `Backend::Handle`: owns the C module and the siblings other threads use.  Calls on one C module are serialized inside the library (one
/// HIP stream per module); `&Module<FFT64Hip>` is `Sync` like the CPU backends (`unsafe impl Sync for Module`, poulpy-hal/src/layouts/
/// module.rs:103-104) and threads other than the creating one run on a sibling module leased from a bounded pool.
pub struct FFT64HipHandle {
    pub(crate) raw: *mut ffi::pz_module,
    /// the thread that created the module uses `raw`
    owner: std::thread::ThreadId,
    /// unique per handle (a freed handle's address may be reused: leases are matched on this, never on the pointer)
    id: u64,
    pool: std::sync::Arc<SiblingPool>,
}
unsafe impl Send for FFT64HipHandle {}
unsafe impl Sync for FFT64HipHandle {}

impl FFT64HipHandle {
    pub(crate) fn new(raw: *mut ffi::pz_module) -> Self {
        check_abi();   // (already done by `Module::new` before `pz_module_new`; kept for other constructors of a handle)
        Self {
            raw,
            owner: std::thread::current().id(),
            id: NEXT_HANDLE_ID.fetch_add(1, std::sync::atomic::Ordering::Relaxed),
            pool: std::sync::Arc::new(SiblingPool {
                free: std::sync::Mutex::new(Vec::new()),
                all: std::sync::Mutex::new(Vec::new()),
                closed: std::sync::atomic::AtomicBool::new(false),
            }),
        }
    }

    /// The C module for the calling thread (see `hal_impl::raw`).
    pub(crate) fn for_this_thread(&self) -> *mut ffi::pz_module {
        if std::thread::current().id() == self.owner {
            return self.raw;
        }
        LEASES.with(|cell| {
            let mut leases = cell.borrow_mut();
            // leases of handles that have been destroyed meanwhile are dropped here (their modules are gone with the handle)
            leases.retain(|l| !l.pool.closed.load(std::sync::atomic::Ordering::Acquire));
            if let Some(l) = leases.iter().find(|l| l.handle_id == self.id) {
                return l.module as *mut ffi::pz_module;
            }
            let reused = self.pool.free.lock().unwrap().pop();
            let module = match reused {
                Some(m) => m,
                None => {
                    let mut sib: *mut ffi::pz_module = std::ptr::null_mut();
                    ffi::check(unsafe { ffi::pz_module_clone(self.raw, &mut sib) }, "pz_module_clone");
                    self.pool.all.lock().unwrap().push(sib as usize);
                    sib as usize
                }
            };
            leases.push(Lease { pool: self.pool.clone(), handle_id: self.id, module });
            module as *mut ffi::pz_module
        })
    }

    /// Number of sibling modules created so far (bounded by the peak number of concurrently live threads; for tests).
    pub fn sibling_count(&self) -> usize {
        self.pool.all.lock().unwrap().len()
    }

    fn free_all(&self) {
        self.pool.closed.store(true, std::sync::atomic::Ordering::Release);
        self.pool.free.lock().unwrap().clear();
        for p in self.pool.all.lock().unwrap().drain(..) {
            unsafe { ffi::pz_module_free(p as *mut ffi::pz_module) }
        }
        unsafe { ffi::pz_module_free(self.raw) }
    }

    /// The raw `pz_module*`, for callers that drive the batched device-resident entry points of include/poulpy_hip.h directly
    /// (`pz_glwe_external_product_batched`, `pz_blind_rotation_execute_batched`, ... on `pz_device_alloc` buffers).
    pub fn as_raw(&self) -> *mut std::ffi::c_void {
        self.raw as *mut std::ffi::c_void
    }
}

/// `Backend::OwnedBuf`: pinned, 64-byte aligned host memory from `pz_alloc_bytes` (hipHostMalloc) — CPU-dereferenceable as
/// `DataMut` requires (poulpy-hal/src/layouts/mod.rs:63), DMA-able for the staged H2D / D2H copies of the per-op path.
pub struct PinnedBuf {
    ptr: NonNull<u8>,
    len: usize,
}
unsafe impl Send for PinnedBuf {}
unsafe impl Sync for PinnedBuf {}

impl PinnedBuf {
    pub fn new(len: usize) -> Self {
        let p = unsafe { ffi::pz_alloc_bytes(len) } as *mut u8;
        Self { ptr: NonNull::new(p).expect("pz_alloc_bytes failed (no HIP device / out of pinned memory)"), len }
    }
}
impl Default for PinnedBuf {
    fn default() -> Self {
        Self { ptr: NonNull::dangling(), len: 0 }
    }
}
impl Drop for PinnedBuf {
    fn drop(&mut self) {
        if self.len != 0 {
            unsafe { ffi::pz_free_bytes(self.ptr.as_ptr() as *mut std::ffi::c_void) }
        }
    }
}
impl AsRef<[u8]> for PinnedBuf {
    fn as_ref(&self) -> &[u8] {
        unsafe { std::slice::from_raw_parts(self.ptr.as_ptr(), self.len) }
    }
}
impl AsMut<[u8]> for PinnedBuf {
    fn as_mut(&mut self) -> &mut [u8] {
        unsafe { std::slice::from_raw_parts_mut(self.ptr.as_ptr(), self.len) }
    }
}
impl PartialEq for PinnedBuf {
    fn eq(&self, other: &Self) -> bool {
        self.as_ref() == other.as_ref()
    }
}
impl Eq for PinnedBuf {}

impl Backend for FFT64Hip {
    type ScalarPrep = f64; // opaque device order, the reference's byte sizes (poulpy-hal/src/layouts/module.rs:51-73)
    type ScalarBig = i64;
    type OwnedBuf = PinnedBuf;
    type Handle = FFT64HipHandle;
    fn alloc_bytes(len: usize) -> Self::OwnedBuf {
        PinnedBuf::new(len.max(1))
    }
    fn from_bytes(bytes: Vec<u8>) -> Self::OwnedBuf {
        let mut b = PinnedBuf::new(bytes.len().max(1));
        b.as_mut()[..bytes.len()].copy_from_slice(&bytes);
        b
    }
    unsafe fn destroy(handle: NonNull<Self::Handle>) {
        unsafe {
            let h: Box<FFT64HipHandle> = Box::from_raw(handle.as_ptr());
            h.free_all();
        }
    }
}
