"""ctypes loader of the in-tree HIP extension (tfhe_rs_string_amd/libtfhe_hip.so).

There is no Python/CPU fallback for compute: if the extension is missing or no HIP device is
visible, the calls raise.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFHE_HIP_LIB") or os.path.join(HERE, "libtfhe_hip.so")  # TFHE_HIP_LIB: diagnostic builds (tools/ only)


class Params(C.Structure):
    """tfhe_hip_params (include/tfhe_hip.h) == ClassicPBSParameters server-side fields
    (reference tfhe/src/shortint/parameters/mod.rs:703-717)."""

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
        ("pbs_order", C.c_uint32),
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

    @property
    def bsk_len(self):
        g = self.glwe_dim + 1
        return self.lwe_dim * self.pbs_level * g * g * self.poly_size

    @property
    def ksk_len(self):
        return self.glwe_dim * self.poly_size * self.ks_level * (self.lwe_dim + 1)


def PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=742):
    """reference tfhe/src/shortint/parameters/mod.rs:703-717; `lwe_dim` may be cut for tests."""
    return Params(lwe_dim, 1, 2048, 23, 1, 3, 5, 4, 4, 0)


def PARAM_MESSAGE_2_CARRY_2_TEST():
    """the reference string layer's non-secure fast-test set (tfhe/examples/fhe_strings/ciphertext.rs:75-90): n = 1, N = 256;
    runs on the generic kernel (variant 0)"""
    return Params(1, 1, 256, 23, 1, 3, 5, 4, 4, 0)


# The classic KS_PBS sets of the reference this library takes (tfhe/src/shortint/parameters/mod.rs:599-820; values only):
# name -> (lwe_dim, glwe_dim, poly_size, pbs_base_log, pbs_level, ks_base_log, ks_level, message_modulus, carry_modulus).
# glwe_dim = 1, N = 2048, 23 x 1 (the four 4-bit sets) run on the wave / workgroup kernels, the others on the generic kernel;
# the sets of 6 bits and more need N >= 8192 and are refused.
CLASSIC_KS_PBS = {
    "PARAM_MESSAGE_1_CARRY_0_KS_PBS": (678, 5, 256, 15, 1, 5, 2, 2, 1),
    "PARAM_MESSAGE_1_CARRY_1_KS_PBS": (684, 3, 512, 18, 1, 4, 3, 2, 2),
    "PARAM_MESSAGE_2_CARRY_0_KS_PBS": (656, 2, 512, 8, 2, 3, 4, 4, 1),
    "PARAM_MESSAGE_1_CARRY_2_KS_PBS": (742, 2, 1024, 23, 1, 4, 3, 2, 4),
    "PARAM_MESSAGE_2_CARRY_1_KS_PBS": (742, 2, 1024, 23, 1, 4, 3, 4, 2),
    "PARAM_MESSAGE_3_CARRY_0_KS_PBS": (742, 2, 1024, 23, 1, 4, 3, 8, 1),
    "PARAM_MESSAGE_1_CARRY_3_KS_PBS": (745, 1, 2048, 23, 1, 3, 5, 2, 8),
    "PARAM_MESSAGE_2_CARRY_2_KS_PBS": (742, 1, 2048, 23, 1, 3, 5, 4, 4),
    "PARAM_MESSAGE_3_CARRY_1_KS_PBS": (742, 1, 2048, 23, 1, 3, 5, 8, 2),
    "PARAM_MESSAGE_4_CARRY_0_KS_PBS": (742, 1, 2048, 23, 1, 3, 5, 16, 1),
    "PARAM_MESSAGE_1_CARRY_4_KS_PBS": (807, 1, 4096, 15, 2, 3, 5, 2, 16),
    "PARAM_MESSAGE_2_CARRY_3_KS_PBS": (856, 1, 4096, 22, 1, 3, 6, 4, 8),
    "PARAM_MESSAGE_3_CARRY_2_KS_PBS": (812, 1, 4096, 22, 1, 3, 5, 8, 4),
    "PARAM_MESSAGE_4_CARRY_1_KS_PBS": (808, 1, 4096, 22, 1, 3, 5, 16, 2),
    "PARAM_MESSAGE_5_CARRY_0_KS_PBS": (807, 1, 4096, 22, 1, 3, 5, 32, 1),
}


def classic_params(name, lwe_dim=None):
    """Params of one of CLASSIC_KS_PBS by the reference's name (lwe_dim can be cut down for tests)"""
    v = CLASSIC_KS_PBS[name]
    return Params(v[0] if lwe_dim is None else lwe_dim, *v[1:], 0)


def PARAM_MESSAGE_2_CARRY_2_PBS_KS(lwe_dim=870):
    """reference tfhe/src/shortint/parameters/mod.rs:1155-1169 (EncryptionKeyChoice::Small:
    ciphertexts live under the small key, PBS first, then keyswitch)."""
    return Params(lwe_dim, 1, 2048, 23, 1, 4, 4, 4, 4, 1)


def PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS(lwe_dim=818):
    """reference tfhe/src/shortint/parameters/multi_bit.rs:115-132; grouping factor 2"""
    return Params(lwe_dim, 1, 2048, 22, 1, 5, 3, 4, 4, 0)


def PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS(lwe_dim=888):
    """reference tfhe/src/shortint/parameters/multi_bit.rs:173-190; grouping factor 3"""
    return Params(lwe_dim, 1, 2048, 21, 1, 7, 2, 4, 4, 0)


_vp = C.c_void_p
_SIG = {
    "tfhe_hip_last_error": (C.c_char_p, []),
    "tfhe_hip_host_alloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "tfhe_hip_host_free": (C.c_int, [_vp]),
    "tfhe_hip_context_create": (C.c_int, [C.POINTER(_vp), C.POINTER(C.c_int), C.c_int]),
    "tfhe_hip_context_device_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "tfhe_hip_context_destroy": (C.c_int, [_vp]),
    "tfhe_hip_context_info": (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "tfhe_hip_server_key_create": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(Params), _vp, C.c_size_t, _vp, C.c_size_t]),
    "tfhe_hip_server_key_create_multi_bit": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(Params), C.c_uint32, _vp, C.c_size_t, _vp, C.c_size_t]),
    "tfhe_hip_server_key_create_seeded": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(Params), C.c_uint32, _vp, C.c_size_t, _vp, _vp,
                                                    C.c_size_t, _vp]),
    "tfhe_hip_csprng_fill": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_size_t]),
    "tfhe_hip_server_key_batch_rounds": (C.c_int, [_vp, _vp, _vp]),
    "tfhe_hip_server_key_get_params": (C.c_int, [_vp, _vp]),
    "tfhe_hip_device_alloc": (C.c_int, [_vp, _vp, C.c_size_t]),
    "tfhe_hip_device_free": (C.c_int, [_vp]),
    "tfhe_hip_device_upload": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "tfhe_hip_device_upload_overlapped": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int]),
    "tfhe_hip_device_download": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "tfhe_hip_device_synchronize": (C.c_int, [_vp]),
    "tfhe_hip_device_lincomb_rows": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp, _vp]),
    "tfhe_hip_lwe_decompress_seeded_batch": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, C.c_size_t]),
    "tfhe_hip_shortint_compressed_ciphertext_serialize": (C.c_int, [C.c_uint64, C.c_uint64, _vp, C.c_uint64, C.c_uint64, C.c_uint32,
                                                                    C.c_uint32, C.c_uint32, _vp, C.c_size_t, _vp]),
    "tfhe_hip_shortint_compressed_ciphertext_deserialize": (C.c_int, [_vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tfhe_hip_compressed_server_key_info": (C.c_int, [_vp, C.c_size_t, C.POINTER(Params), _vp, _vp, _vp]),
    "tfhe_hip_server_key_create_from_compressed": (C.c_int, [_vp, C.POINTER(_vp), _vp, C.c_size_t, C.POINTER(Params), _vp]),
    "tfhe_hip_compressed_server_key_serialize": (C.c_int, [C.POINTER(Params), C.c_uint32, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp,
                                                           _vp, C.c_size_t, _vp]),
    "tfhe_hip_shortint_ciphertext_serialize": (C.c_int, [_vp, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                                         _vp, C.c_size_t, _vp]),
    "tfhe_hip_shortint_ciphertext_deserialize": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tfhe_hip_lwe_keyswitch_key_deserialize": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tfhe_hip_lwe_bootstrap_key_deserialize": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tfhe_hip_lwe_keyswitch_key_serialize": (C.c_int, [_vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_size_t, _vp]),
    "tfhe_hip_lwe_bootstrap_key_serialize": (C.c_int, [_vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_size_t, _vp]),
    "tfhe_hip_server_key_create_from_standard_keys": (C.c_int, [_vp, C.POINTER(_vp), _vp, C.c_size_t, _vp, C.c_size_t, C.c_uint32,
                                                                C.c_uint32, C.c_uint32, C.POINTER(Params)]),
    "tfhe_hip_server_key_destroy": (C.c_int, [_vp]),
    "tfhe_hip_lut_generate": (C.c_int, [C.POINTER(Params), _vp, _vp, C.POINTER(C.c_uint64)]),
    "tfhe_hip_keyswitch_batch": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "tfhe_hip_pbs_batch": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t]),
    "tfhe_hip_ks_pbs_batch": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t]),
    "tfhe_hip_pbs_ks_batch": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t]),
    "tfhe_hip_apply_lookup_table_batch": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t]),
    "tfhe_hip_pbs_ks_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_apply_lookup_table_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_keyswitch_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_pbs_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_ks_pbs_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_profiling_enable": (C.c_int, [_vp, C.c_int]),
    "tfhe_hip_profiling_read": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "tfhe_hip_server_key_fourier_bsk": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "tfhe_hip_set_variant": (C.c_int, [_vp, C.c_int]),
    "tfhe_hip_exact_external_product_batch": (C.c_int, [C.c_int, _vp, C.c_uint32, _vp, _vp, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "tfhe_hip_server_key_device_status": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.c_int]),
    "tfhe_hip_server_key_release_stream": (C.c_int, [_vp, _vp, C.POINTER(C.c_size_t)]),
    "tfhe_hip_debug_skip_pair_signal": (C.c_int, [_vp, C.c_int]),
}

_lib = None


class TfheHipError(RuntimeError):
    pass


def lib():
    """Loads libtfhe_hip.so (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TfheHipError(
                f"{LIB_PATH} is missing: build it with `python -m tfhe_rs_string_amd.build` "
                "(the engine has no CPU fallback)"
            )
        try:
            # PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process cannot both
            # own the GPU.  Loading torch first makes this library bind to the runtime torch uses,
            # so torch tensors / streams and the engine share one HIP context.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIG.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return list(_SIG.keys())


def check(rc):
    if rc != 0:
        raise TfheHipError(lib().tfhe_hip_last_error().decode())
