//! `FFT64Hip`: poulpy-hal backend whose FFT64 hot path runs on an MI355X through libpoulpy_hip.so.
//!
//! Shape of the crate (same as poulpy-cpu-avx, poulpy-cpu-avx/src/hal_impl.rs:51-61):
//!   * `impl Backend for FFT64Hip` — `ScalarPrep = f64`, `ScalarBig = i64`, pinned host buffers,
//!     handle = the C module (device twiddle tables, stream, workspace);
//!   * `unsafe impl HalImpl<FFT64Hip>` — every method that touches `ScalarPrep` bytes
//!     (VecZnxDft / SvpPPol / VmpPMat) forwards to the C ABI, because those bytes are in the
//!     backend's private "device order"; pure-`i64` `VecZnx` ops reuse poulpy-cpu-ref's defaults
//!     through the `hal_impl_*!` macros exactly as FFT64Avx does.
//!   * (feature `core-fused`) `CoreImpl` overrides for `glwe_external_product` / `glwe_keyswitch`
//!     that keep ciphertexts on the device.
//!
//! The convolution family (`cnv_*`, hal_impl.rs:670-754) also produces `VecZnxDft`; it is not part
//! of this path and panics with `unimplemented!` in this backend.
mod ffi;
mod hal_impl;

use std::ptr::NonNull;

use poulpy_hal::layouts::Backend;

pub struct FFT64Hip;

/// `Backend::Handle`: owns the C module.  Immutable after construction except for the module's
/// internal, mutex-protected workspace, so `&Module<FFT64Hip>` is `Sync` like the CPU backends.
#[repr(C)]
pub struct FFT64HipHandle {
    pub(crate) raw: *mut ffi::pz_module,
}
unsafe impl Send for FFT64HipHandle {}
unsafe impl Sync for FFT64HipHandle {}

/// Pinned, 64-byte aligned host memory from `pz_alloc_bytes` (hipHostMalloc): CPU-dereferenceable
/// as `Backend::OwnedBuf: DataMut` requires (poulpy-hal/src/layouts/mod.rs:63), DMA-able for the
/// staged H2D/D2H copies.
pub struct PinnedBuf {
    ptr: NonNull<u8>,
    len: usize,
}
unsafe impl Send for PinnedBuf {}
unsafe impl Sync for PinnedBuf {}
impl PinnedBuf {
    pub fn new(len: usize) -> Self {
        let p = unsafe { ffi::pz_alloc_bytes(len) } as *mut u8;
        Self { ptr: NonNull::new(p).expect("pz_alloc_bytes failed"), len }
    }
}
impl Drop for PinnedBuf {
    fn drop(&mut self) {
        unsafe { ffi::pz_free_bytes(self.ptr.as_ptr() as *mut _) }
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

impl Backend for FFT64Hip {
    type ScalarPrep = f64; // opaque device order, same byte size as the reference (module.rs:51-65)
    type ScalarBig = i64;
    type OwnedBuf = PinnedBuf;
    type Handle = FFT64HipHandle;
    fn alloc_bytes(len: usize) -> Self::OwnedBuf {
        PinnedBuf::new(len)
    }
    fn from_bytes(bytes: Vec<u8>) -> Self::OwnedBuf {
        let mut b = PinnedBuf::new(bytes.len());
        b.as_mut().copy_from_slice(&bytes);
        b
    }
    unsafe fn destroy(handle: NonNull<Self::Handle>) {
        unsafe {
            let h = Box::from_raw(handle.as_ptr());
            ffi::pz_module_free(h.raw);
        }
    }
}
