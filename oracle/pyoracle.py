"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (tfhe_rs_string_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class Params(C.Structure):
    """Mirror of orc_params (shortint/parameters/mod.rs:703-717)."""

    _fields_ = [
        ("lwe_dim", C.c_uint32),
        ("glwe_dim", C.c_uint32),
        ("poly_size", C.c_uint32),
        ("pbs_base_log", C.c_uint32),
        ("pbs_level", C.c_uint32),
        ("ks_base_log", C.c_uint32),
        ("ks_level", C.c_uint32),
        ("message_modulus", C.c_uint32),
        ("carry_modulus", C.c_uint32),
        ("lwe_std", C.c_double),
        ("glwe_std", C.c_double),
    ]

    @property
    def big_size(self):
        return self.glwe_dim * self.poly_size + 1

    @property
    def small_size(self):
        return self.lwe_dim + 1

    @property
    def lut_len(self):
        return (self.glwe_dim + 1) * self.poly_size


def param_message_2_carry_2(lwe_dim=742):
    """PARAM_MESSAGE_2_CARRY_2_KS_PBS; `lwe_dim` can be reduced for fast tests (same kernels)."""
    return Params(lwe_dim, 1, 2048, 23, 1, 3, 5, 4, 4, 7.069849454709433e-6, 2.9403601535432533e-16)


def param_message_2_carry_2_test():
    """PARAM_MESSAGE_2_CARRY_2_TEST, the reference string layer's own non-secure fast-test set
    (tfhe/examples/fhe_strings/ciphertext.rs:75-90: lwe_dimension 1, polynomial_size 256)."""
    return Params(1, 1, 256, 23, 1, 3, 5, 4, 4, 7.069849454709433e-6, 2.9403601535432533e-16)


def param_custom(lwe_dim, poly_size, pbs_base_log=23, pbs_level=1, ks_base_log=3, ks_level=5, glwe_dim=1, message_modulus=4,
                 carry_modulus=4, lwe_std=7.069849454709433e-6, glwe_std=2.9403601535432533e-16):
    """another GLWE dimension / polynomial size / decomposition / message space: the generic kernel's cases (the noise defaults
    are those of PARAM_MESSAGE_2_CARRY_2_KS_PBS)"""
    return Params(lwe_dim, glwe_dim, poly_size, pbs_base_log, pbs_level, ks_base_log, ks_level, message_modulus, carry_modulus,
                  lwe_std, glwe_std)


def param_message_2_carry_2_pbs_ks(lwe_dim=870):
    """PARAM_MESSAGE_2_CARRY_2_PBS_KS (shortint/parameters/mod.rs:1155-1169)."""
    return Params(lwe_dim, 1, 2048, 23, 1, 4, 4, 4, 4, 6.791658447437413e-7, 2.9403601535432533e-16)


def param_multi_bit_message_2_carry_2_group_2(lwe_dim=818):
    """PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS (shortint/parameters/multi_bit.rs:115-132), grouping factor 2."""
    return Params(lwe_dim, 1, 2048, 22, 1, 5, 3, 4, 4, 0.000002226459789930014, 0.0000000000000003152931493498455)


def param_multi_bit_message_2_carry_2_group_3(lwe_dim=888):
    """PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS (shortint/parameters/multi_bit.rs:173-190), grouping factor 3."""
    return Params(lwe_dim, 1, 2048, 21, 1, 7, 2, 4, 4, 0.0000006125031601933181, 0.0000000000000003152931493498455)


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("tfhe_oracle.c", "tfhe_oracle_fast.c", "tfhe_oracle_ntt.c", "tfhe_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


_lib = None
_BENCH_LIB_PATH = os.path.join(_HERE, "liboracle_bench.so")
_use_bench = False


def use_bench_build():
    """bench.py's cpu_baseline leg: (re)build oracle/liboracle_bench.so FOR THIS HOST (-march=native; the file is
    never shipped between machines) and bind to it instead of the strict build.  Same source, same results."""
    global _use_bench, _lib
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle_bench.so"])
    _use_bench = True
    _lib = None


def bench_build_flags():
    return subprocess.check_output(["make", "-C", _HERE, "-s", "bench-flags"], text=True).strip()


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_BENCH_LIB_PATH if _use_bench else _LIB_PATH)
    PP = C.POINTER(Params)
    RP = C.POINTER(Rng)
    sig = {
        "orc_closest_representable": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint32]),
        "orc_decompose": (None, [C.c_uint64, C.c_uint32, C.c_uint32, i64p]),
        "orc_modswitch": (C.c_uint64, [C.c_uint64, C.c_uint32]),
        "orc_monomial_div": (None, [u64p, u64p, C.c_uint64, C.c_uint32]),
        "orc_monomial_mul": (None, [u64p, u64p, C.c_uint64, C.c_uint32]),
        "orc_monomial_mul_and_subtract": (None, [u64p, u64p, C.c_uint64, C.c_uint32]),
        "orc_sample_extract": (None, [u64p, C.c_uint32, C.c_uint32, u64p]),
        "orc_keyswitch": (None, [PP, u64p, u64p, u64p]),
        "orc_lut_generate": (C.c_uint64, [PP, u64p, u64p]),
        "orc_trivial_pbs": (C.c_uint64, [PP, u64p, C.c_uint64]),
        "orc_lwe_add_assign": (None, [u64p, u64p, C.c_size_t]),
        "orc_lwe_scalar_mul_assign": (None, [u64p, C.c_uint64, C.c_size_t]),
        "orc_torus_from_f64": (C.c_uint64, [C.c_double]),
        "orc_f64_to_u64_wrap": (C.c_uint64, [C.c_double]),
        "orc_slice_sub_scalar_mul": (None, [u64p, u64p, C.c_uint64, C.c_size_t]),
        "orc_poly_add_mul": (None, [u64p, u64p, u64p, C.c_uint32]),
        "orc_fft_new": (C.c_void_p, [C.c_uint32]),
        "orc_fft_free": (None, [C.c_void_p]),
        "orc_fft_twist_re": (C.POINTER(C.c_double), [C.c_void_p]),
        "orc_fft_twist_im": (C.POINTER(C.c_double), [C.c_void_p]),
        "orc_fft_w_re": (C.POINTER(C.c_double), [C.c_void_p]),
        "orc_fft_w_im": (C.POINTER(C.c_double), [C.c_void_p]),
        "orc_fft_forward_integer": (None, [C.c_void_p, i64p, f64p, f64p]),
        "orc_fft_forward_torus": (None, [C.c_void_p, u64p, f64p, f64p]),
        "orc_fft_add_backward_torus": (None, [C.c_void_p, u64p, f64p, f64p]),
        "orc_bsk_to_fourier": (None, [PP, u64p, f64p, f64p]),
        "orc_add_external_product": (None, [PP, C.c_void_p, u64p, f64p, f64p, u64p]),
        "orc_blind_rotate": (None, [PP, C.c_void_p, f64p, f64p, u64p, u64p]),
        "orc_pbs": (None, [PP, C.c_void_p, f64p, f64p, u64p, u64p, u64p]),
        "orc_ks_pbs": (None, [PP, C.c_void_p, u64p, f64p, f64p, u64p, u64p, u64p]),
        "orc_ks_pbs_batch": (None, [PP, C.c_void_p, u64p, f64p, f64p, u64p, u64p, u32p, u64p, C.c_size_t, C.c_int]),
        "orcf_ks_pbs_batch": (None, [PP, C.c_void_p, u64p, f64p, f64p, u64p, u64p, u32p, u64p, C.c_size_t, C.c_int]),
        "orc_keyswitch_batch": (None, [PP, u64p, u64p, u64p, C.c_size_t, C.c_int]),
        "orc_ntt_new": (C.c_void_p, [C.c_uint32]),
        "orc_ntt_free": (None, [C.c_void_p]),
        "orc_ntt_ggsw_len": (C.c_size_t, [PP]),
        "orc_ntt_ggsw_prepare": (None, [PP, C.c_void_p, u64p, u64p]),
        "orc_ntt_bsk_prepare": (None, [PP, C.c_void_p, u64p, u64p]),
        "orc_ntt_add_external_product": (None, [PP, C.c_void_p, u64p, u64p, u64p]),
        "orc_schoolbook_add_external_product": (None, [PP, u64p, u64p, u64p]),
        "orc_ntt_pbs_batch": (None, [PP, C.c_void_p, u64p, u64p, u64p, u32p, u64p, C.c_size_t, C.c_int]),
        "orc_pbs_batch": (None, [PP, C.c_void_p, f64p, f64p, u64p, u64p, u32p, u64p, C.c_size_t, C.c_int]),
        "orc_rng_seed": (None, [RP, C.c_uint64]),
        "orc_rng_u64": (C.c_uint64, [RP]),
        "orc_rng_noise": (C.c_uint64, [RP, C.c_double]),
        "orc_gen_binary_key": (None, [RP, u64p, C.c_size_t]),
        "orc_lwe_encrypt": (None, [RP, u64p, C.c_uint32, C.c_double, C.c_uint64, u64p]),
        "orc_lwe_decrypt": (C.c_uint64, [u64p, C.c_uint32, u64p]),
        "orc_gen_bsk": (None, [RP, PP, u64p, u64p, u64p]),
        "orc_gen_ksk": (None, [RP, PP, u64p, u64p, u64p]),
        "orc_shortint_encode": (C.c_uint64, [PP, C.c_uint64]),
        "orc_shortint_decode": (C.c_uint64, [PP, C.c_uint64]),
        "orc_bsk_len": (C.c_size_t, [PP]),
        "orc_ksk_len": (C.c_size_t, [PP]),
        "orc_max_threads": (C.c_int, []),
        "orc_mb_bsk_len": (C.c_size_t, [PP, C.c_uint32]),
        "orc_mb_gen_bsk": (None, [RP, PP, C.c_uint32, u64p, u64p, u64p]),
        "orc_polys_to_fourier": (None, [C.c_uint32, C.c_size_t, u64p, f64p, f64p]),
        "orc_fft_freq_of_pos": (C.c_uint32, [C.c_uint32, C.c_uint32]),
        "orc_mb_pbs_batch": (None, [PP, C.c_uint32, C.c_void_p, f64p, f64p, u64p, u64p, u32p, u64p, C.c_size_t, C.c_int]),
        "orc_mb_ks_pbs_batch": (None, [PP, C.c_uint32, C.c_void_p, u64p, f64p, f64p, u64p, u64p, u32p, u64p, C.c_size_t, C.c_int]),
        "orc_aes_sbox": (C.POINTER(C.c_uint8), []),
        "orc_aes128_key_schedule": (None, [u8p, u8p]),
        "orc_aes128_encrypt": (None, [u8p, u8p, u8p]),
        "orc_csprng_bytes": (None, [u8p, C.c_uint64, u8p, C.c_size_t]),
        "orc_csprng_words": (None, [u8p, C.c_uint64, u64p, C.c_size_t]),
        "orc_seeded_expand": (None, [u8p, u64p, C.c_size_t, C.c_size_t, C.c_size_t, u64p]),
        "orc_seeded_reseed_glwe_rows": (None, [PP, u8p, u64p, u64p, C.c_size_t, u64p]),
        "orc_seeded_reseed_lwe_list": (None, [u8p, u64p, C.c_uint32, u64p, C.c_size_t, u64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class Fft:
    def __init__(self, poly_size):
        self.N = poly_size
        self.h = lib().orc_fft_new(poly_size)
        if not self.h:
            raise ValueError("poly_size must be a power of two >= 8")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_fft_free(self.h)
            self.h = None

    def _arr(self, fn):
        return np.ctypeslib.as_array(fn(self.h), shape=(self.N // 2,)).copy()

    def twist(self):
        return self._arr(lib().orc_fft_twist_re), self._arr(lib().orc_fft_twist_im)

    def twiddles(self):
        return self._arr(lib().orc_fft_w_re), self._arr(lib().orc_fft_w_im)

    def forward_integer(self, poly_i64):
        re = np.empty(self.N // 2)
        im = np.empty(self.N // 2)
        lib().orc_fft_forward_integer(self.h, np.ascontiguousarray(poly_i64, dtype=np.int64), re, im)
        return re, im

    def forward_torus(self, poly_u64):
        re = np.empty(self.N // 2)
        im = np.empty(self.N // 2)
        lib().orc_fft_forward_torus(self.h, np.ascontiguousarray(poly_u64, dtype=np.uint64), re, im)
        return re, im

    def add_backward_torus(self, poly_u64, re, im):
        re = np.array(re, dtype=np.float64)
        im = np.array(im, dtype=np.float64)
        lib().orc_fft_add_backward_torus(self.h, poly_u64, re, im)


class KeySet:
    """Client + server keys for one parameter set, generated by the oracle's seeded RNG.

    Mirrors which secret key feeds what (shortint/engine/server_side.rs:54-...):
    BSK = small LWE key bits under the GLWE key; KSK = big (flattened GLWE) key -> small key;
    messages are encrypted under the big key (EncryptionKeyChoice::Big).
    """

    def __init__(self, params, seed=0x5EED0001, fourier=True):
        L = lib()
        self.p = params
        self.rng = Rng()
        L.orc_rng_seed(C.byref(self.rng), seed)
        p = params
        self.glwe_sk = np.zeros(p.glwe_dim * p.poly_size, dtype=np.uint64)
        self.lwe_sk = np.zeros(p.lwe_dim, dtype=np.uint64)
        L.orc_gen_binary_key(C.byref(self.rng), self.glwe_sk, self.glwe_sk.size)
        L.orc_gen_binary_key(C.byref(self.rng), self.lwe_sk, self.lwe_sk.size)
        self.bsk = np.zeros(L.orc_bsk_len(C.byref(p)), dtype=np.uint64)
        self.ksk = np.zeros(L.orc_ksk_len(C.byref(p)), dtype=np.uint64)
        L.orc_gen_bsk(C.byref(self.rng), C.byref(p), self.lwe_sk, self.glwe_sk, self.bsk)
        L.orc_gen_ksk(C.byref(self.rng), C.byref(p), self.glwe_sk, self.lwe_sk, self.ksk)
        self.fft = Fft(p.poly_size)
        self.bsk_re = self.bsk_im = None
        if fourier:
            self.to_fourier()

    def to_fourier(self):
        if self.bsk_re is None:
            self.bsk_re = np.empty(self.bsk.size // 2)
            self.bsk_im = np.empty(self.bsk.size // 2)
            lib().orc_bsk_to_fourier(C.byref(self.p), self.bsk, self.bsk_re, self.bsk_im)

    # -- client side
    def encrypt(self, msgs):
        p = self.p
        msgs = np.atleast_1d(np.asarray(msgs, dtype=np.uint64))
        out = np.zeros((msgs.size, p.big_size), dtype=np.uint64)
        for i, m in enumerate(msgs):
            pt = lib().orc_shortint_encode(C.byref(p), int(m))
            lib().orc_lwe_encrypt(C.byref(self.rng), self.glwe_sk, p.big_size - 1, p.glwe_std, pt, out[i])
        return out

    def decrypt(self, cts):
        p = self.p
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, p.big_size)
        return np.array(
            [
                lib().orc_shortint_decode(C.byref(p), lib().orc_lwe_decrypt(self.glwe_sk, p.big_size - 1, ct))
                for ct in cts
            ],
            dtype=np.uint64,
        )

    def encrypt_small(self, msgs):
        """EncryptionKeyChoice::Small: messages live under the small LWE key (PBS_KS order)."""
        p = self.p
        msgs = np.atleast_1d(np.asarray(msgs, dtype=np.uint64))
        out = np.zeros((msgs.size, p.small_size), dtype=np.uint64)
        for i, m in enumerate(msgs):
            pt = lib().orc_shortint_encode(C.byref(p), int(m))
            lib().orc_lwe_encrypt(C.byref(self.rng), self.lwe_sk, p.lwe_dim, p.lwe_std, pt, out[i])
        return out

    def decrypt_small(self, cts):
        p = self.p
        return np.array([lib().orc_shortint_decode(C.byref(p), int(v)) for v in self.decrypt_small_raw(cts)], dtype=np.uint64)

    def pbs_ks_batch(self, small_cts, luts, lut_idx, threads=1):
        """programmable_bootstrap_keyswitch_assign (shortint/server_key/mod.rs:859-932)."""
        return self.keyswitch_batch(self.pbs_batch(small_cts, luts, lut_idx, threads), threads)

    def decrypt_small_raw(self, cts):
        p = self.p
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, p.small_size)
        return np.array([lib().orc_lwe_decrypt(self.lwe_sk, p.lwe_dim, ct) for ct in cts], dtype=np.uint64)

    def decrypt_raw(self, cts):
        p = self.p
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, p.big_size)
        return np.array([lib().orc_lwe_decrypt(self.glwe_sk, p.big_size - 1, ct) for ct in cts], dtype=np.uint64)

    # -- server side (oracle)
    def lut(self, table):
        out = np.zeros(self.p.lut_len, dtype=np.uint64)
        deg = lib().orc_lut_generate(C.byref(self.p), np.asarray(table, dtype=np.uint64), out)
        return out, deg

    def keyswitch_batch(self, cts, threads=1):
        p = self.p
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, p.big_size)
        out = np.zeros((cts.shape[0], p.small_size), dtype=np.uint64)
        lib().orc_keyswitch_batch(C.byref(p), self.ksk, cts, out, cts.shape[0], threads)
        return out

    def pbs_batch(self, small_cts, luts, lut_idx, threads=1):
        p = self.p
        self.to_fourier()
        small_cts = np.ascontiguousarray(small_cts, dtype=np.uint64).reshape(-1, p.small_size)
        luts = np.ascontiguousarray(luts, dtype=np.uint64).reshape(-1, p.lut_len)
        lut_idx = np.ascontiguousarray(lut_idx, dtype=np.uint32)
        out = np.zeros((small_cts.shape[0], p.big_size), dtype=np.uint64)
        lib().orc_pbs_batch(C.byref(p), self.fft.h, self.bsk_re, self.bsk_im, small_cts, luts, lut_idx, out,
                            small_cts.shape[0], threads)
        return out

    def ks_pbs_batch(self, cts, luts, lut_idx, threads=1, fast=False):
        """fast=True: the tuned port (tfhe_oracle_fast.c, bench.py's CPU baseline); same bits as the plain restatement"""
        p = self.p
        self.to_fourier()
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, p.big_size)
        luts = np.ascontiguousarray(luts, dtype=np.uint64).reshape(-1, p.lut_len)
        lut_idx = np.ascontiguousarray(lut_idx, dtype=np.uint32)
        out = np.zeros((cts.shape[0], p.big_size), dtype=np.uint64)
        fn = lib().orcf_ks_pbs_batch if fast else lib().orc_ks_pbs_batch
        fn(C.byref(p), self.fft.h, self.ksk, self.bsk_re, self.bsk_im, cts, luts, lut_idx, out, cts.shape[0], threads)
        return out


class MultiBitKeySet(KeySet):
    """Keys of a multi-bit parameter set (grouping factor g): the bootstrap key holds 2^g GGSWs per group of g
    small-key bits (lwe_multi_bit_bootstrap_key_generation.rs:87-170)."""

    def __init__(self, params, grouping_factor, seed=0x5EED0002):
        L = lib()
        self.p = params
        self.g = grouping_factor
        assert params.lwe_dim % grouping_factor == 0
        self.rng = Rng()
        L.orc_rng_seed(C.byref(self.rng), seed)
        p = params
        self.glwe_sk = np.zeros(p.glwe_dim * p.poly_size, dtype=np.uint64)
        self.lwe_sk = np.zeros(p.lwe_dim, dtype=np.uint64)
        L.orc_gen_binary_key(C.byref(self.rng), self.glwe_sk, self.glwe_sk.size)
        L.orc_gen_binary_key(C.byref(self.rng), self.lwe_sk, self.lwe_sk.size)
        self.bsk = np.zeros(L.orc_mb_bsk_len(C.byref(p), self.g), dtype=np.uint64)
        self.ksk = np.zeros(L.orc_ksk_len(C.byref(p)), dtype=np.uint64)
        L.orc_mb_gen_bsk(C.byref(self.rng), C.byref(p), self.g, self.lwe_sk, self.glwe_sk, self.bsk)
        L.orc_gen_ksk(C.byref(self.rng), C.byref(p), self.glwe_sk, self.lwe_sk, self.ksk)
        self.fft = Fft(p.poly_size)
        self.bsk_re = self.bsk_im = None
        self.to_fourier()

    def to_fourier(self):
        if self.bsk_re is None:
            self.bsk_re = np.empty(self.bsk.size // 2)
            self.bsk_im = np.empty(self.bsk.size // 2)
            lib().orc_polys_to_fourier(self.p.poly_size, self.bsk.size // self.p.poly_size, self.bsk, self.bsk_re, self.bsk_im)

    def pbs_batch(self, small_cts, luts, lut_idx, threads=1):
        p = self.p
        small_cts = np.ascontiguousarray(small_cts, dtype=np.uint64).reshape(-1, p.small_size)
        luts = np.ascontiguousarray(luts, dtype=np.uint64).reshape(-1, p.lut_len)
        lut_idx = np.ascontiguousarray(np.broadcast_to(np.asarray(lut_idx, dtype=np.uint32), (small_cts.shape[0],)))
        out = np.zeros((small_cts.shape[0], p.big_size), dtype=np.uint64)
        lib().orc_mb_pbs_batch(C.byref(p), self.g, self.fft.h, self.bsk_re, self.bsk_im, small_cts, luts, lut_idx, out,
                               small_cts.shape[0], threads)
        return out

    def ks_pbs_batch(self, cts, luts, lut_idx, threads=1):
        p = self.p
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, p.big_size)
        luts = np.ascontiguousarray(luts, dtype=np.uint64).reshape(-1, p.lut_len)
        lut_idx = np.ascontiguousarray(np.broadcast_to(np.asarray(lut_idx, dtype=np.uint32), (cts.shape[0],)))
        out = np.zeros((cts.shape[0], p.big_size), dtype=np.uint64)
        lib().orc_mb_ks_pbs_batch(C.byref(p), self.g, self.fft.h, self.ksk, self.bsk_re, self.bsk_im, cts, luts, lut_idx, out,
                                  cts.shape[0], threads)
        return out


# ---- seeded (compressed) keys -------------------------------------------------------------------------
def seed_bytes(seed):
    """Seed(u128) -> its 16 native (little-endian) bytes, the AES-128 key of the CSPRNG"""
    return np.frombuffer(int(seed).to_bytes(16, "little"), dtype=np.uint8).copy()


def csprng_words(seed, first, count):
    out = np.zeros(count, dtype=np.uint64)
    lib().orc_csprng_words(seed_bytes(seed), first, out, count)
    return out


class SeededKeys:
    """The seeded form of a KeySet's server keys (shortint CompressedServerKey, server_key/compressed.rs:44-55): the
    bodies and one compression seed per key.  TEST DATA: the KeySet's keys are re-seeded in place (same secret, noise
    and plaintexts, masks replaced by the CSPRNG's output), so `keyset.bsk` / `keyset.ksk` afterwards ARE the
    decompressed keys and the keyset's Fourier key is recomputed."""

    def __init__(self, keyset, bsk_seed=0x0123456789ABCDEF0FEDCBA987654321, ksk_seed=0x00112233445566778899AABBCCDDEEFF, fourier=True):
        L = lib()
        p = keyset.p
        self.p, self.bsk_seed, self.ksk_seed = p, bsk_seed, ksk_seed
        rows = keyset.bsk.size // ((p.glwe_dim + 1) * p.poly_size)
        self.bsk_bodies = np.zeros(rows * p.poly_size, dtype=np.uint64)
        L.orc_seeded_reseed_glwe_rows(C.byref(p), seed_bytes(bsk_seed), keyset.glwe_sk, keyset.bsk, rows, self.bsk_bodies)
        count = keyset.ksk.size // (p.lwe_dim + 1)
        self.ksk_bodies = np.zeros(count, dtype=np.uint64)
        L.orc_seeded_reseed_lwe_list(seed_bytes(ksk_seed), keyset.lwe_sk, p.lwe_dim, keyset.ksk, count, self.ksk_bodies)
        keyset.bsk_re = keyset.bsk_im = None
        if fourier:
            keyset.to_fourier()

    def decompress(self):
        """seeded_lwe_bootstrap_key_decompression.rs / seeded_lwe_keyswitch_key_decompression.rs -> (bsk, ksk)"""
        L = lib()
        p = self.p
        n_poly = p.poly_size
        rows = self.bsk_bodies.size // n_poly
        bsk = np.zeros(rows * (p.glwe_dim + 1) * n_poly, dtype=np.uint64)
        L.orc_seeded_expand(seed_bytes(self.bsk_seed), self.bsk_bodies, rows, p.glwe_dim * n_poly, n_poly, bsk)
        ksk = np.zeros(self.ksk_bodies.size * (p.lwe_dim + 1), dtype=np.uint64)
        L.orc_seeded_expand(seed_bytes(self.ksk_seed), self.ksk_bodies, self.ksk_bodies.size, p.lwe_dim, 1, ksk)
        return bsk, ksk
