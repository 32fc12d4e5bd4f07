// hip_backend.rs -- safe Rust wrappers over ffi.rs: what the reference-side host code looks like once the C-ABI is bound.
//
// Where it goes: `tfhe/src/hip_backend/mod.rs` (feature `hip`), next to the generated `ffi.rs`.  The types mirror the
// reference's own: `HipServerKey` stands beside `shortint::ServerKey` (tfhe/src/shortint/server_key/mod.rs:176-230) and
// takes its keys from it; `HipStringServerKey` carries the method names of `StringServerKey`
// (tfhe/examples/fhe_strings/server_key/*.rs) over flat block arrays.  Errors: the C-ABI returns 0 / 1 with a thread-local
// message (the reference's `catch_panic` convention, c_api/utils.rs:3-12); here they become `Result<_, HipError>`.
// NOT COMPILED in this repository (no rustc in the build image); tests/test_rust_ffi.py checks that every `ffi::` function
// named below exists in ffi.rs with that number of arguments.
#![allow(dead_code)]
use super::ffi;
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::ptr;

#[derive(Debug, Clone)]
pub struct HipError(pub String);

fn check(rc: c_int) -> Result<(), HipError> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::tfhe_hip_last_error()) }.to_string_lossy().into_owned();
    Err(HipError(msg))
}

/// One or several GPUs of this process (`device_ids[0]` is the primary device).
pub struct HipContext {
    raw: *mut ffi::TfheHipContext,
}
unsafe impl Send for HipContext {}
unsafe impl Sync for HipContext {}

impl HipContext {
    pub fn new(device_ids: &[i32]) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::tfhe_hip_context_create(&mut raw, device_ids.as_ptr(), device_ids.len() as c_int) })?;
        Ok(Self { raw })
    }
    pub fn device_count(&self) -> Result<i32, HipError> {
        let mut n = 0;
        check(unsafe { ffi::tfhe_hip_context_device_count(self.raw, &mut n) })?;
        Ok(n)
    }
}
impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { ffi::tfhe_hip_context_destroy(self.raw) };
    }
}

/// The device-resident keyswitch + bootstrap key: the server half of `shortint::ServerKey`.
pub struct HipServerKey {
    raw: *mut ffi::TfheHipServerKey,
    pub params: ffi::TfheHipParams,
    _ctx: std::sync::Arc<HipContext>,
}
unsafe impl Send for HipServerKey {}
unsafe impl Sync for HipServerKey {} // the batch entry points are re-entrant (shortint/engine/mod.rs:23-25: thread-local scratch)

impl HipServerKey {
    /// From the standard-domain keys the reference holds while it builds its own `ServerKey`
    /// (`LweBootstrapKeyOwned<u64>` / `LweKeyswitchKeyOwned<u64>` as flat slices, shortint/engine/server_side.rs:60-90).
    pub fn new(ctx: std::sync::Arc<HipContext>, params: ffi::TfheHipParams, bsk_std: &[u64], ksk: &[u64]) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::tfhe_hip_server_key_create(ctx.raw, &mut raw, &params, bsk_std.as_ptr(), bsk_std.len(), ksk.as_ptr(), ksk.len()) })?;
        Ok(Self { raw, params, _ctx: ctx })
    }
    /// `LweMultiBitBootstrapKey` (grouping factor 2 or 3).
    pub fn new_multi_bit(ctx: std::sync::Arc<HipContext>, params: ffi::TfheHipParams, grouping_factor: u32, bsk_std: &[u64],
                         ksk: &[u64]) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        check(unsafe {
            ffi::tfhe_hip_server_key_create_multi_bit(ctx.raw, &mut raw, &params, grouping_factor, bsk_std.as_ptr(), bsk_std.len(),
                                                      ksk.as_ptr(), ksk.len())
        })?;
        Ok(Self { raw, params, _ctx: ctx })
    }
    /// `bincode::serialize(&CompressedServerKey)` as a client ships it: the masks are regenerated on the device.
    pub fn from_compressed_bytes(ctx: std::sync::Arc<HipContext>, bytes: &[u8]) -> Result<(Self, u32), HipError> {
        let mut raw = ptr::null_mut();
        let mut params = ffi::TfheHipParams::default();
        let mut grouping = 0u32;
        check(unsafe { ffi::tfhe_hip_server_key_create_from_compressed(ctx.raw, &mut raw, bytes.as_ptr(), bytes.len(), &mut params, &mut grouping) })?;
        Ok((Self { raw, params, _ctx: ctx }, grouping))
    }
    /// The two standard-domain keys in their bincode form.
    pub fn from_standard_key_bytes(ctx: std::sync::Arc<HipContext>, ksk_bytes: &[u8], bsk_bytes: &[u8], message_modulus: u32,
                                   carry_modulus: u32, pbs_order: u32) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        let mut params = ffi::TfheHipParams::default();
        check(unsafe {
            ffi::tfhe_hip_server_key_create_from_standard_keys(ctx.raw, &mut raw, ksk_bytes.as_ptr(), ksk_bytes.len(), bsk_bytes.as_ptr(),
                                                               bsk_bytes.len(), message_modulus, carry_modulus, pbs_order, &mut params)
        })?;
        Ok(Self { raw, params, _ctx: ctx })
    }
    pub fn big_lwe_size(&self) -> usize {
        (self.params.glwe_dim * self.params.poly_size) as usize + 1
    }
    pub fn small_lwe_size(&self) -> usize {
        self.params.lwe_dim as usize + 1
    }
    /// `ServerKey::generate_lookup_table` (shortint/server_key/mod.rs:383-399): the accumulator and its degree.
    pub fn generate_lookup_table(&self, f: impl Fn(u64) -> u64) -> Result<(Vec<u64>, u64), HipError> {
        let m = (self.params.message_modulus * self.params.carry_modulus) as u64;
        let table: Vec<u64> = (0..m).map(f).collect();
        let mut acc = vec![0u64; ((self.params.glwe_dim + 1) * self.params.poly_size) as usize];
        let mut degree = 0u64;
        check(unsafe { ffi::tfhe_hip_lut_generate(&self.params, table.as_ptr(), acc.as_mut_ptr(), &mut degree) })?;
        Ok((acc, degree))
    }
    /// `keyswitch_lwe_ciphertext` on B ciphertexts (core_crypto/algorithms/lwe_keyswitch.rs:96-170).
    pub fn keyswitch_batch(&self, lwe_in: &[u64]) -> Result<Vec<u64>, HipError> {
        let b = lwe_in.len() / self.big_lwe_size();
        let mut out = vec![0u64; b * self.small_lwe_size()];
        check(unsafe { ffi::tfhe_hip_keyswitch_batch(self.raw, lwe_in.as_ptr(), out.as_mut_ptr(), b) })?;
        Ok(out)
    }
    /// `programmable_bootstrap_lwe_ciphertext_mem_optimized` on B small ciphertexts.
    pub fn pbs_batch(&self, lwe_in: &[u64], luts: &[u64], lut_idx: &[u32]) -> Result<Vec<u64>, HipError> {
        let b = lut_idx.len();
        let lut_len = ((self.params.glwe_dim + 1) * self.params.poly_size) as usize;
        let mut out = vec![0u64; b * self.big_lwe_size()];
        check(unsafe { ffi::tfhe_hip_pbs_batch(self.raw, lwe_in.as_ptr(), luts.as_ptr(), luts.len() / lut_len, lut_idx.as_ptr(), out.as_mut_ptr(), b) })?;
        Ok(out)
    }
    /// The crypto part of `keyswitch_programmable_bootstrap_assign` (shortint/server_key/mod.rs:783-857) on B ciphertexts: ONE launch.
    pub fn ks_pbs_batch(&self, lwe_in: &[u64], luts: &[u64], lut_idx: &[u32]) -> Result<Vec<u64>, HipError> {
        let b = lut_idx.len();
        let lut_len = ((self.params.glwe_dim + 1) * self.params.poly_size) as usize;
        let mut out = vec![0u64; b * self.big_lwe_size()];
        check(unsafe { ffi::tfhe_hip_ks_pbs_batch(self.raw, lwe_in.as_ptr(), luts.as_ptr(), luts.len() / lut_len, lut_idx.as_ptr(), out.as_mut_ptr(), b) })?;
        Ok(out)
    }
    /// Dispatch on `pbs_order` like `apply_lookup_table_assign` (shortint/server_key/mod.rs:465-476).
    pub fn apply_lookup_table_batch(&self, lwe_in: &[u64], luts: &[u64], lut_idx: &[u32]) -> Result<Vec<u64>, HipError> {
        let b = lut_idx.len();
        let lut_len = ((self.params.glwe_dim + 1) * self.params.poly_size) as usize;
        let in_size = if self.params.pbs_order == 0 { self.big_lwe_size() } else { self.small_lwe_size() };
        let mut out = vec![0u64; b * in_size];
        check(unsafe {
            ffi::tfhe_hip_apply_lookup_table_batch(self.raw, lwe_in.as_ptr(), luts.as_ptr(), luts.len() / lut_len, lut_idx.as_ptr(), out.as_mut_ptr(), b)
        })?;
        Ok(out)
    }
    /// Device-buffer form on the caller's HIP stream (asynchronous; check `device_status` after synchronising).
    pub unsafe fn ks_pbs_batch_device(&self, d_in: *const u64, d_luts: *const u64, n_luts: usize, d_idx: *const u32, d_out: *mut u64, b: usize,
                                      stream: *mut c_void) -> Result<(), HipError> {
        check(ffi::tfhe_hip_ks_pbs_batch_device(self.raw, d_in, d_luts, n_luts, d_idx, d_out, b, stream))
    }
    /// The key's device error word (0 = every launch so far produced valid results); sticky until cleared.
    pub fn device_status(&self, clear: bool) -> Result<u32, HipError> {
        let mut code = 0u32;
        check(unsafe { ffi::tfhe_hip_server_key_device_status(self.raw, &mut code, clear as c_int) })?;
        Ok(code)
    }
    /// Hands the per-stream workspace of a stream the host is about to destroy back.
    pub unsafe fn release_stream(&self, stream: *mut c_void) -> Result<usize, HipError> {
        let mut left = 0usize;
        check(ffi::tfhe_hip_server_key_release_stream(self.raw, stream, &mut left))?;
        Ok(left)
    }
    /// Ciphertexts one round of the latency / throughput kernel takes (what a level scheduler batches against).
    pub fn batch_rounds(&self) -> Result<(usize, usize), HipError> {
        let (mut a, mut b) = (0usize, 0usize);
        check(unsafe { ffi::tfhe_hip_server_key_batch_rounds(self.raw, &mut a, &mut b) })?;
        Ok((a, b))
    }
}
impl Drop for HipServerKey {
    fn drop(&mut self) {
        unsafe { ffi::tfhe_hip_server_key_destroy(self.raw) };
    }
}

/// `Padding` of the reference (tfhe/examples/fhe_strings/ciphertext.rs:20-44), as the C-ABI numbers it.
#[repr(i32)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Padding {
    None = 0,
    Final = 1,
    Initial = 2,
    InitialAndFinal = 3,
    Anywhere = 4,
}

/// A pattern argument (`FhePattern`, pattern.rs:37-129): encrypted blocks or clear bytes.
pub enum Pattern<'a> {
    Encrypted(&'a [u64], Padding),
    Clear(&'a str),
}
/// An integer argument (`FheIntegerArg`, integer_arg.rs:21-88): clear or a `num_blocks`-block radix ciphertext.
pub enum IntegerArg<'a> {
    Clear(u64),
    Encrypted(&'a [u64]),
}

/// The level-batching executor over one device key: `integer::ServerKey` + `StringServerKey` of the reference in one handle.
/// Strings are flat arrays: `len` chars x 4 blocks x `big` words, least significant block first (client_key.rs:27-60).
pub struct HipStringServerKey {
    raw: *mut ffi::TfheHipHostKey,
    big: usize,
    num_blocks: usize,
    _sk: std::sync::Arc<HipServerKey>,
}
unsafe impl Send for HipStringServerKey {} // one host key per host thread over a shared HipServerKey

impl HipStringServerKey {
    pub fn new(sk: std::sync::Arc<HipServerKey>) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::tfhe_hip_host_key_create(sk.raw, &sk.params, &mut raw) })?;
        Ok(Self { raw, big: sk.big_lwe_size(), num_blocks: 4, _sk: sk })
    }
    /// `gen_keys_number_blocks(num_blocks)` (ciphertext.rs:113-126): width of every length / index / count.
    pub fn set_num_blocks(&mut self, num_blocks: usize) -> Result<(), HipError> {
        check(unsafe { ffi::tfhe_hip_host_key_set_num_blocks(self.raw, num_blocks) })?;
        self.num_blocks = num_blocks;
        Ok(())
    }
    fn chars(&self, s: &[u64]) -> usize {
        s.len() / (4 * self.big)
    }
    fn block(&self) -> Vec<u64> {
        vec![0u64; self.big]
    }
    fn radix(&self) -> Vec<u64> {
        vec![0u64; self.num_blocks * self.big]
    }
    /// Independent operations between `batch_begin` and `batch_end` share their launches (rayon's role in the reference).
    pub fn batch<T>(&self, f: impl FnOnce(&Self) -> Result<T, HipError>) -> Result<T, HipError> {
        check(unsafe { ffi::tfhe_hip_host_key_batch_begin(self.raw) })?;
        let r = f(self);
        check(unsafe { ffi::tfhe_hip_host_key_batch_end(self.raw) })?;
        r
    }

    // ---- comparisons (server_key/comparisons.rs) -------------------------------------------------------------------
    pub fn eq(&self, s1: &[u64], s2: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        check(unsafe { ffi::tfhe_hip_string_eq(self.raw, s1.as_ptr(), self.chars(s1), s2.as_ptr(), self.chars(s2), out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn ne(&self, s1: &[u64], s2: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        check(unsafe { ffi::tfhe_hip_string_ne(self.raw, s1.as_ptr(), self.chars(s1), s2.as_ptr(), self.chars(s2), out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn le(&self, s1: &[u64], s2: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        check(unsafe { ffi::tfhe_hip_string_le(self.raw, s1.as_ptr(), self.chars(s1), s2.as_ptr(), self.chars(s2), out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn ge(&self, s1: &[u64], s2: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        check(unsafe { ffi::tfhe_hip_string_ge(self.raw, s1.as_ptr(), self.chars(s1), s2.as_ptr(), self.chars(s2), out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn eq_ignore_case(&self, s1: &[u64], s2: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        check(unsafe { ffi::tfhe_hip_string_eq_ignore_case(self.raw, s1.as_ptr(), self.chars(s1), s2.as_ptr(), self.chars(s2), out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn eq_padded(&self, s1: &[u64], p1: Padding, s2: &[u64], p2: Padding) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        check(unsafe {
            ffi::tfhe_hip_string_eq_padded(self.raw, s1.as_ptr(), self.chars(s1), p1 as c_int, s2.as_ptr(), self.chars(s2), p2 as c_int, out.as_mut_ptr())
        })?;
        Ok(out)
    }

    // ---- contains / starts_with / ends_with (contains.rs, starts_with.rs, ends_with.rs) ------------------------------
    pub fn contains(&self, s: &[u64], pattern: Pattern) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        match pattern {
            Pattern::Clear(c) => check(unsafe { ffi::tfhe_hip_string_contains_clear(self.raw, s.as_ptr(), self.chars(s), c.as_ptr(), c.len(), out.as_mut_ptr()) })?,
            Pattern::Encrypted(p, Padding::None) => {
                check(unsafe { ffi::tfhe_hip_string_contains(self.raw, s.as_ptr(), self.chars(s), p.as_ptr(), self.chars(p), out.as_mut_ptr()) })?
            }
            Pattern::Encrypted(p, pad) => check(unsafe {
                ffi::tfhe_hip_string_contains_padded(self.raw, s.as_ptr(), self.chars(s), Padding::None as c_int, p.as_ptr(), self.chars(p), pad as c_int,
                                                     out.as_mut_ptr())
            })?,
        }
        Ok(out)
    }
    pub fn starts_with(&self, s: &[u64], pattern: Pattern) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        match pattern {
            Pattern::Clear(c) => check(unsafe { ffi::tfhe_hip_string_starts_with_clear(self.raw, s.as_ptr(), self.chars(s), c.as_ptr(), c.len(), out.as_mut_ptr()) })?,
            Pattern::Encrypted(p, Padding::None) => {
                check(unsafe { ffi::tfhe_hip_string_starts_with(self.raw, s.as_ptr(), self.chars(s), p.as_ptr(), self.chars(p), out.as_mut_ptr()) })?
            }
            Pattern::Encrypted(p, pad) => check(unsafe {
                ffi::tfhe_hip_string_starts_with_padded(self.raw, s.as_ptr(), self.chars(s), Padding::None as c_int, p.as_ptr(), self.chars(p), pad as c_int,
                                                        out.as_mut_ptr())
            })?,
        }
        Ok(out)
    }
    pub fn ends_with(&self, s: &[u64], pattern: Pattern) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        match pattern {
            Pattern::Clear(c) => check(unsafe { ffi::tfhe_hip_string_ends_with_clear(self.raw, s.as_ptr(), self.chars(s), c.as_ptr(), c.len(), out.as_mut_ptr()) })?,
            Pattern::Encrypted(p, _) => check(unsafe { ffi::tfhe_hip_string_ends_with(self.raw, s.as_ptr(), self.chars(s), p.as_ptr(), self.chars(p), out.as_mut_ptr()) })?,
        }
        Ok(out)
    }

    // ---- find / rfind (find.rs): FheOptionInt = (found, index) -----------------------------------------------------------
    fn find_impl(&self, s: &[u64], pattern: Pattern, reverse: bool) -> Result<(Vec<u64>, Vec<u64>), HipError> {
        let (mut found, mut index) = (self.block(), self.radix());
        match pattern {
            Pattern::Clear(c) => check(unsafe {
                ffi::tfhe_hip_string_find_clear(self.raw, s.as_ptr(), self.chars(s), c.as_ptr(), c.len(), reverse as c_int, found.as_mut_ptr(), index.as_mut_ptr())
            })?,
            Pattern::Encrypted(p, Padding::None) => check(unsafe {
                ffi::tfhe_hip_string_find(self.raw, s.as_ptr(), self.chars(s), p.as_ptr(), self.chars(p), reverse as c_int, found.as_mut_ptr(), index.as_mut_ptr())
            })?,
            Pattern::Encrypted(p, pad) => check(unsafe {
                ffi::tfhe_hip_string_find_padded(self.raw, s.as_ptr(), self.chars(s), Padding::None as c_int, p.as_ptr(), self.chars(p), pad as c_int,
                                                 reverse as c_int, found.as_mut_ptr(), index.as_mut_ptr())
            })?,
        }
        Ok((found, index))
    }
    pub fn find(&self, s: &[u64], pattern: Pattern) -> Result<(Vec<u64>, Vec<u64>), HipError> {
        self.find_impl(s, pattern, false)
    }
    pub fn rfind(&self, s: &[u64], pattern: Pattern) -> Result<(Vec<u64>, Vec<u64>), HipError> {
        self.find_impl(s, pattern, true)
    }

    // ---- case, length, trimming (change_case.rs, is_empty.rs, trim.rs, strip.rs) ----------------------------------------------
    pub fn to_uppercase(&self, s: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = vec![0u64; s.len()];
        check(unsafe { ffi::tfhe_hip_string_to_uppercase(self.raw, s.as_ptr(), self.chars(s), out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn to_lowercase(&self, s: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = vec![0u64; s.len()];
        check(unsafe { ffi::tfhe_hip_string_to_lowercase(self.raw, s.as_ptr(), self.chars(s), out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn len(&self, s: &[u64], padding: Padding) -> Result<Vec<u64>, HipError> {
        let mut out = self.radix();
        check(unsafe { ffi::tfhe_hip_string_len(self.raw, s.as_ptr(), self.chars(s), padding as c_int, out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn is_empty(&self, s: &[u64], padding: Padding) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        check(unsafe { ffi::tfhe_hip_string_is_empty(self.raw, s.as_ptr(), self.chars(s), padding as c_int, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// mode 0 = trim_start, 1 = trim_end, 2 = trim; returns (content, encrypted length)
    pub fn trim(&self, s: &[u64], mode: i32) -> Result<(Vec<u64>, Vec<u64>), HipError> {
        let (mut content, mut length) = (vec![0u64; s.len()], self.radix());
        check(unsafe { ffi::tfhe_hip_string_trim(self.raw, s.as_ptr(), self.chars(s), mode, content.as_mut_ptr(), length.as_mut_ptr()) })?;
        Ok((content, length))
    }
    /// strip_prefix / strip_suffix: FheOptionString = (found, content, length)
    pub fn strip(&self, s: &[u64], pattern: Pattern, suffix: bool) -> Result<(Vec<u64>, Vec<u64>, Vec<u64>), HipError> {
        let (mut found, mut content, mut length) = (self.block(), vec![0u64; s.len()], self.radix());
        match pattern {
            Pattern::Clear(c) => check(unsafe {
                ffi::tfhe_hip_string_strip_clear(self.raw, s.as_ptr(), self.chars(s), c.as_ptr(), c.len(), suffix as c_int, found.as_mut_ptr(),
                                                 content.as_mut_ptr(), length.as_mut_ptr())
            })?,
            Pattern::Encrypted(p, _) => check(unsafe {
                ffi::tfhe_hip_string_strip(self.raw, s.as_ptr(), self.chars(s), p.as_ptr(), self.chars(p), suffix as c_int, found.as_mut_ptr(),
                                           content.as_mut_ptr(), length.as_mut_ptr())
            })?,
        }
        Ok((found, content, length))
    }
    pub fn push_padding_to_end(&self, s: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = vec![0u64; s.len()];
        check(unsafe { ffi::tfhe_hip_string_push_padding_to_end(self.raw, s.as_ptr(), self.chars(s), out.as_mut_ptr()) })?;
        Ok(out)
    }

    // ---- the split family (split.rs, rsplit.rs, rsplitn.rs, split_terminator.rs, split_inclusive.rs): FheSplit -----------------
    /// `kind`: ffi::TFHE_HIP_SPLIT .. TFHE_HIP_SPLIT_ASCII_WHITESPACE; returns (parts, each `len` chars long; number_parts radix)
    pub fn split(&self, s: &[u64], padding: Padding, pattern: Pattern, kind: c_int, n: IntegerArg) -> Result<(Vec<Vec<u64>>, Vec<u64>), HipError> {
        let len = self.chars(s);
        let (p_enc, p_clear, p_len, p_pad) = match pattern {
            Pattern::Encrypted(p, pad) => (p.as_ptr(), ptr::null(), self.chars(p), pad as c_int),
            Pattern::Clear(c) => (ptr::null(), c.as_ptr(), c.len(), 0),
        };
        let (n_clear, n_enc, n_is_enc) = match n {
            IntegerArg::Clear(v) => (v, ptr::null(), 0),
            IntegerArg::Encrypted(e) => (0, e.as_ptr(), 1),
        };
        let mut cap = 0usize;
        check(unsafe { ffi::tfhe_hip_string_split_max_parts(kind, len, p_len, p_pad, n_is_enc, n_clear, &mut cap) })?;
        let part_words = len * 4 * self.big;
        let mut flat = vec![0u64; cap * part_words];
        let (mut number, mut got) = (self.radix(), 0usize);
        check(unsafe {
            ffi::tfhe_hip_string_split(self.raw, s.as_ptr(), len, padding as c_int, p_enc, p_clear, p_len, p_pad, kind, n_clear, n_enc, flat.as_mut_ptr(), cap,
                                       number.as_mut_ptr(), &mut got)
        })?;
        Ok((flat.chunks(part_words.max(1)).take(got).map(|c| c.to_vec()).collect(), number))
    }
    /// `replace` / `replacen` (replace.rs:20-52): returns (content, its padding)
    pub fn replace(&self, s: &[u64], padding: Padding, from: Pattern, to: Pattern, n: Option<IntegerArg>) -> Result<(Vec<u64>, i32), HipError> {
        let len = self.chars(s);
        let (f_enc, f_clear, f_len, f_pad) = match from {
            Pattern::Encrypted(p, pad) => (p.as_ptr(), ptr::null(), self.chars(p), pad as c_int),
            Pattern::Clear(c) => (ptr::null(), c.as_ptr(), c.len(), 0),
        };
        let (t_enc, t_clear, t_len) = match to {
            Pattern::Encrypted(p, _) => (p.as_ptr(), ptr::null(), self.chars(p)),
            Pattern::Clear(c) => (ptr::null(), c.as_ptr(), c.len()),
        };
        let (limited, n_clear, n_enc) = match n {
            None => (0, 0, ptr::null()),
            Some(IntegerArg::Clear(v)) => (1, v, ptr::null()),
            Some(IntegerArg::Encrypted(e)) => (1, 0, e.as_ptr()),
        };
        let mut cap = 0usize;
        check(unsafe { ffi::tfhe_hip_string_replace_max_len(len, f_len, f_pad, t_len, &mut cap) })?;
        let mut out = vec![0u64; cap * 4 * self.big];
        let (mut out_len, mut out_pad) = (0usize, 0 as c_int);
        check(unsafe {
            ffi::tfhe_hip_string_replace(self.raw, s.as_ptr(), len, padding as c_int, f_enc, f_clear, f_len, f_pad, t_enc, t_clear, t_len, limited, n_clear,
                                         n_enc, out.as_mut_ptr(), cap, &mut out_len, &mut out_pad)
        })?;
        out.truncate(out_len * 4 * self.big);
        Ok((out, out_pad))
    }

    // ---- radix integers (integer/server_key/radix_parallel/*.rs) --------------------------------------------------------------
    pub fn integer_eq(&self, lhs: &[u64], rhs: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        check(unsafe { ffi::tfhe_hip_integer_eq(self.raw, lhs.as_ptr(), rhs.as_ptr(), lhs.len() / self.big, out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn integer_mul(&self, lhs: &[u64], rhs: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = vec![0u64; lhs.len()];
        check(unsafe { ffi::tfhe_hip_integer_mul(self.raw, lhs.as_ptr(), rhs.as_ptr(), lhs.len() / self.big, out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn integer_add(&self, lhs: &[u64], rhs: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = vec![0u64; lhs.len()];
        check(unsafe { ffi::tfhe_hip_integer_add(self.raw, lhs.as_ptr(), rhs.as_ptr(), lhs.len() / self.big, out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn integer_sub(&self, lhs: &[u64], rhs: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = vec![0u64; lhs.len()];
        check(unsafe { ffi::tfhe_hip_integer_sub(self.raw, lhs.as_ptr(), rhs.as_ptr(), lhs.len() / self.big, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// op: 0 = gt, 1 = ge, 2 = lt, 3 = le (comparator.rs:1079-1330)
    pub fn integer_compare(&self, lhs: &[u64], rhs: &[u64], op: c_int) -> Result<Vec<u64>, HipError> {
        let mut out = self.block();
        check(unsafe { ffi::tfhe_hip_integer_compare(self.raw, lhs.as_ptr(), rhs.as_ptr(), lhs.len() / self.big, op, out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn integer_if_then_else(&self, condition: &[u64], t: &[u64], f: &[u64]) -> Result<Vec<u64>, HipError> {
        let mut out = vec![0u64; t.len()];
        check(unsafe { ffi::tfhe_hip_integer_if_then_else(self.raw, condition.as_ptr(), t.as_ptr(), f.as_ptr(), t.len() / self.big, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// PBS and launch counters of this host key (the numbers the benchmarks print).
    pub fn stats(&self, reset: bool) -> Result<(u64, u64), HipError> {
        let (mut pbs, mut launches) = (0u64, 0u64);
        check(unsafe { ffi::tfhe_hip_host_key_stats(self.raw, &mut pbs, &mut launches, reset as c_int) })?;
        Ok((pbs, launches))
    }
}
impl Drop for HipStringServerKey {
    fn drop(&mut self) {
        unsafe { ffi::tfhe_hip_host_key_destroy(self.raw) };
    }
}
