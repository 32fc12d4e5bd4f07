"""Host-side handles over the C-ABI (include/tfhe_hip.h).

`Context` / `ServerKey` own the opaque handles; `ServerKey.ks_pbs_batch` & co. are the batched
replacements of `shortint::ServerKey::apply_lookup_table` (reference
tfhe/src/shortint/server_key/mod.rs:457-476, 783-857).  numpy arrays go through the host-buffer
entry points; torch CUDA tensors go through the device-buffer entry points on torch's current
stream (PyTorch-ROCm provides device memory and streams only).
"""
import ctypes as C

import numpy as np

from . import _native
from ._native import Params, TfheHipError, check


def _np_u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(C.c_void_p)


class Context:
    """tfhe_hip_context: one GPU (`Context(0)`) or several of this process (`Context([0, 1, 2, 3])`: keys are replicated,
    every batch is split over the devices; the first one is the primary, whose memory the device-buffer calls use)."""

    def __init__(self, device_ids=0):
        ids = [int(device_ids)] if np.isscalar(device_ids) else [int(d) for d in device_ids]
        self._h = C.c_void_p()
        arr = (C.c_int * len(ids))(*ids)
        check(_native.lib().tfhe_hip_context_create(C.byref(self._h), arr, len(ids)))
        self.device_ids = ids
        self.device_id = ids[0]

    def device_count(self):
        n = C.c_int()
        check(_native.lib().tfhe_hip_context_device_count(self._h, C.byref(n)))
        return n.value

    def info(self):
        buf = C.create_string_buffer(128)
        cu = C.c_int()
        check(_native.lib().tfhe_hip_context_info(self._h, buf, 128, C.byref(cu)))
        return buf.value.decode(), cu.value

    def close(self):
        if self._h:
            _native.lib().tfhe_hip_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def generate_lookup_table(params, f):
    """ServerKey::generate_lookup_table (reference shortint/server_key/mod.rs:383-399).
    `f` is a callable or a table of message_modulus*carry_modulus values.
    Returns (glwe accumulator as np.uint64[(k+1)N], degree)."""
    m = params.message_modulus * params.carry_modulus
    table = np.array([f(i) for i in range(m)] if callable(f) else list(f), dtype=np.uint64)
    if table.size != m:
        raise ValueError(f"table must have {m} entries")
    out = np.zeros(params.lut_len, dtype=np.uint64)
    mx = C.c_uint64()
    check(_native.lib().tfhe_hip_lut_generate(C.byref(params), table.ctypes.data_as(C.c_void_p),
                                              out.ctypes.data_as(C.c_void_p), C.byref(mx)))
    return out, mx.value


def _seed16(seed):
    return int(seed).to_bytes(16, "little") if isinstance(seed, int) else bytes(seed)


def csprng_words(ctx, seed, first_word, count):
    """u64 outputs of the reference's CSPRNG (concrete-csprng AES-128-CTR), computed by the device kernel"""
    out = np.zeros(count, dtype=np.uint64)
    check(_native.lib().tfhe_hip_csprng_fill(ctx._h, _seed16(seed), first_word, out.ctypes.data_as(C.c_void_p), count))
    return out


def compressed_server_key_info(data):
    """parameters found in `bincode::serialize(&CompressedServerKey)` bytes -> (Params, grouping_factor, max_degree)"""
    buf = bytes(data)
    params, g, md, used = _native.Params(), C.c_uint32(), C.c_uint64(), C.c_size_t()
    check(_native.lib().tfhe_hip_compressed_server_key_info(buf, len(buf), C.byref(params), C.byref(g), C.byref(md), C.byref(used)))
    return params, g.value, md.value, used.value


def serialize_compressed_server_key(params, bsk_bodies, bsk_seed, ksk_bodies, ksk_seed, grouping_factor=0):
    bb, pb = _np_u64(bsk_bodies)
    kb, pk = _np_u64(ksk_bodies)
    n = C.c_size_t()
    args = (C.byref(params), grouping_factor, pb, bb.size, _seed16(bsk_seed), pk, kb.size, _seed16(ksk_seed))
    check(_native.lib().tfhe_hip_compressed_server_key_serialize(*args, None, 0, C.byref(n)))
    out = (C.c_uint8 * n.value)()
    check(_native.lib().tfhe_hip_compressed_server_key_serialize(*args, out, n.value, C.byref(n)))
    return bytes(out)


def _two_call_bytes(fn, *args):
    n = C.c_size_t()
    check(fn(*args, None, 0, C.byref(n)))
    out = np.empty(n.value, dtype=np.uint8)
    check(fn(*args, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
    return out


def serialize_keyswitch_key(params, ksk):
    """`bincode::serialize(&LweKeyswitchKey<Vec<u64>>)` -> np.uint8 array"""
    a, pa = _np_u64(ksk)
    return _two_call_bytes(_native.lib().tfhe_hip_lwe_keyswitch_key_serialize, pa, a.size, params.ks_base_log, params.ks_level, params.lwe_dim)


def serialize_bootstrap_key(params, bsk_std):
    """`bincode::serialize(&LweBootstrapKey<Vec<u64>>)` (standard domain) -> np.uint8 array"""
    a, pa = _np_u64(bsk_std)
    return _two_call_bytes(_native.lib().tfhe_hip_lwe_bootstrap_key_serialize, pa, a.size, params.glwe_dim, params.poly_size,
                           params.pbs_base_log, params.pbs_level)


def _bytes_arg(data):
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else data, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def deserialize_keyswitch_key(data):
    """-> (ksk, ks_base_log, ks_level, input_lwe_dim, output_lwe_dim, bytes consumed)"""
    a, pa = _bytes_arg(data)
    n, used = C.c_size_t(), C.c_size_t()
    bl, lv, din, dout = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    L = _native.lib()
    check(L.tfhe_hip_lwe_keyswitch_key_deserialize(pa, a.size, None, 0, C.byref(n), None, None, None, None, None))
    out = np.empty(n.value, dtype=np.uint64)
    check(L.tfhe_hip_lwe_keyswitch_key_deserialize(pa, a.size, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n), C.byref(bl), C.byref(lv),
                                                   C.byref(din), C.byref(dout), C.byref(used)))
    return out, bl.value, lv.value, din.value, dout.value, used.value


def deserialize_bootstrap_key(data):
    """-> (bsk_std, glwe_dim, poly_size, pbs_base_log, pbs_level, input_lwe_dim, bytes consumed)"""
    a, pa = _bytes_arg(data)
    n, used = C.c_size_t(), C.c_size_t()
    k, N, bl, lv, din = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    L = _native.lib()
    check(L.tfhe_hip_lwe_bootstrap_key_deserialize(pa, a.size, None, 0, C.byref(n), None, None, None, None, None, None))
    out = np.empty(n.value, dtype=np.uint64)
    check(L.tfhe_hip_lwe_bootstrap_key_deserialize(pa, a.size, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n), C.byref(k), C.byref(N),
                                                   C.byref(bl), C.byref(lv), C.byref(din), C.byref(used)))
    return out, k.value, N.value, bl.value, lv.value, din.value, used.value


def serialize_ciphertext(params, lwe, degree, noise_level=1):
    """`bincode::serialize(&shortint::Ciphertext)`"""
    a, pa = _np_u64(lwe)
    n = C.c_size_t()
    args = (pa, a.size, int(degree), int(noise_level), params.message_modulus, params.carry_modulus, params.pbs_order)
    check(_native.lib().tfhe_hip_shortint_ciphertext_serialize(*args, None, 0, C.byref(n)))
    out = (C.c_uint8 * n.value)()
    check(_native.lib().tfhe_hip_shortint_ciphertext_serialize(*args, out, n.value, C.byref(n)))
    return bytes(out)


def deserialize_ciphertext(data):
    """-> (lwe, degree, noise_level, message_modulus, carry_modulus, pbs_order, bytes consumed)"""
    buf = bytes(data)
    n, deg, noise, used = C.c_size_t(), C.c_uint64(), C.c_uint64(), C.c_size_t()
    mm, cm, order = C.c_uint32(), C.c_uint32(), C.c_uint32()
    L = _native.lib()
    check(L.tfhe_hip_shortint_ciphertext_deserialize(buf, len(buf), None, 0, C.byref(n), None, None, None, None, None, None))
    lwe = np.zeros(n.value, dtype=np.uint64)
    check(L.tfhe_hip_shortint_ciphertext_deserialize(buf, len(buf), lwe.ctypes.data_as(C.c_void_p), lwe.size, C.byref(n), C.byref(deg),
                                                     C.byref(noise), C.byref(mm), C.byref(cm), C.byref(order), C.byref(used)))
    return lwe, deg.value, noise.value, mm.value, cm.value, order.value, used.value


def decompress_ciphertexts(ctx, bodies, seeds, lwe_dim):
    """shortint CompressedCiphertext -> LWE ciphertexts on the device: bodies [B], seeds (ints or 16-byte strings) [B]"""
    bodies, pb = _np_u64(bodies)
    raw = b"".join(_seed16(s) for s in seeds)
    out = np.zeros((bodies.size, lwe_dim + 1), dtype=np.uint64)
    check(_native.lib().tfhe_hip_lwe_decompress_seeded_batch(ctx._h, pb, raw, lwe_dim, out.ctypes.data_as(C.c_void_p), bodies.size))
    return out


def serialize_compressed_ciphertext(params, body, seed, degree, noise_level=1):
    """`bincode::serialize(&shortint::CompressedCiphertext)`"""
    size = params.big_size if params.pbs_order == 0 else params.small_size
    n = C.c_size_t()
    args = (int(body), size, _seed16(seed), int(degree), int(noise_level), params.message_modulus, params.carry_modulus, params.pbs_order)
    out = (C.c_uint8 * 128)()
    check(_native.lib().tfhe_hip_shortint_compressed_ciphertext_serialize(*args, out, 128, C.byref(n)))
    return bytes(out[:n.value])


def deserialize_compressed_ciphertext(data):
    """-> (body, lwe_size, seed bytes, degree, noise_level, message_modulus, carry_modulus, pbs_order, bytes consumed)"""
    buf = bytes(data)
    body, size, deg, noise, used = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_size_t()
    mm, cm, order = C.c_uint32(), C.c_uint32(), C.c_uint32()
    seed = (C.c_uint8 * 16)()
    check(_native.lib().tfhe_hip_shortint_compressed_ciphertext_deserialize(buf, len(buf), C.byref(body), C.byref(size), seed, C.byref(deg),
                                                                            C.byref(noise), C.byref(mm), C.byref(cm), C.byref(order),
                                                                            C.byref(used)))
    return body.value, size.value, bytes(seed), deg.value, noise.value, mm.value, cm.value, order.value, used.value


def exact_external_product_batch(ggsw_std, pbs_base_log, glwe_in, glwe_acc, device_id=0, timing_reps=0):
    """EXPERIMENT (include/tfhe_hip_experimental.h): glwe_acc += ggsw (x) glwe_in with exact integer arithmetic (NTT modulo
    2^64 - 2^32 + 1) on the GPU; returns (glwe_acc, kernel ms or None)"""
    g, pg = _np_u64(ggsw_std)
    a, pa = _np_u64(glwe_in)
    out = np.array(glwe_acc, dtype=np.uint64, order="C")
    B = a.size // 4096
    ms = C.c_double()
    check(_native.lib().tfhe_hip_exact_external_product_batch(device_id, pg, pbs_base_log, pa, out.ctypes.data_as(C.c_void_p), B,
                                                              timing_reps, C.byref(ms)))
    return out, (ms.value if timing_reps > 0 else None)


class ServerKey:
    """Device-resident keyswitch key + Fourier bootstrap key (reference `shortint::ServerKey`)."""

    def __init__(self, ctx, params, bsk_std, ksk, grouping_factor=0):
        """grouping_factor 2 / 3: a multi-bit PBS key (`bsk_std` in LweMultiBitBootstrapKey order)"""
        self.ctx = ctx
        self.params = params
        self.grouping_factor = grouping_factor
        bsk_std, pb = _np_u64(bsk_std)
        ksk, pk = _np_u64(ksk)
        self._h = C.c_void_p()
        if grouping_factor:
            check(_native.lib().tfhe_hip_server_key_create_multi_bit(ctx._h, C.byref(self._h), C.byref(params), grouping_factor,
                                                                     pb, bsk_std.size, pk, ksk.size))
        else:
            check(_native.lib().tfhe_hip_server_key_create(ctx._h, C.byref(self._h), C.byref(params), pb, bsk_std.size,
                                                           pk, ksk.size))

    @classmethod
    def from_seeded(cls, ctx, params, bsk_bodies, bsk_seed, ksk_bodies, ksk_seed, grouping_factor=0):
        """A seeded (compressed) key: the bodies and the two compression seeds (ints or 16 bytes); the masks are
        generated on the device (shortint CompressedServerKey -> ServerKey)."""
        self = cls.__new__(cls)
        self.ctx, self.params, self.grouping_factor = ctx, params, grouping_factor
        bb, pb = _np_u64(bsk_bodies)
        kb, pk = _np_u64(ksk_bodies)
        s1, s2 = _seed16(bsk_seed), _seed16(ksk_seed)
        self._h = C.c_void_p()
        check(_native.lib().tfhe_hip_server_key_create_seeded(ctx._h, C.byref(self._h), C.byref(params), grouping_factor, pb, bb.size,
                                                              s1, pk, kb.size, s2))
        return self

    @classmethod
    def from_compressed_bytes(cls, ctx, data):
        """`bincode::serialize(&CompressedServerKey)` bytes -> device key; the parameters come from the key"""
        self = cls.__new__(cls)
        buf = bytes(data)
        params = _native.Params()
        g = C.c_uint32()
        self._h = C.c_void_p()
        check(_native.lib().tfhe_hip_server_key_create_from_compressed(ctx._h, C.byref(self._h), buf, len(buf), C.byref(params),
                                                                       C.byref(g)))
        self.ctx, self.params, self.grouping_factor = ctx, params, g.value
        return self

    @classmethod
    def from_standard_key_bytes(cls, ctx, ksk_bytes, bsk_bytes, message_modulus=4, carry_modulus=4, pbs_order=0):
        """`bincode::serialize` images of the client's LweKeyswitchKey<Vec<u64>> and (standard-domain)
        LweBootstrapKey<Vec<u64>> -> device key; the dimensions and decompositions come from the keys"""
        self = cls.__new__(cls)
        ka, pk = _bytes_arg(ksk_bytes)
        ba, pb = _bytes_arg(bsk_bytes)
        params = _native.Params()
        self._h = C.c_void_p()
        check(_native.lib().tfhe_hip_server_key_create_from_standard_keys(ctx._h, C.byref(self._h), pk, ka.size, pb, ba.size, message_modulus,
                                                                          carry_modulus, pbs_order, C.byref(params)))
        self.ctx, self.params, self.grouping_factor = ctx, params, 0
        return self

    def close(self):
        if getattr(self, "_h", None):
            _native.lib().tfhe_hip_server_key_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer path ------------------------------------------------------------------
    def keyswitch_batch(self, lwe_in):
        p = self.params
        lwe_in, pi = _np_u64(lwe_in)
        B = lwe_in.size // p.big_size
        if lwe_in.size != B * p.big_size:
            raise ValueError("input is not a whole number of big LWE ciphertexts")
        out = np.zeros((B, p.small_size), dtype=np.uint64)
        check(_native.lib().tfhe_hip_keyswitch_batch(self._h, pi, out.ctypes.data_as(C.c_void_p), B))
        return out

    def _luts(self, luts, lut_idx, B):
        p = self.params
        luts, pl = _np_u64(luts)
        n_luts = luts.size // p.lut_len
        if luts.size != n_luts * p.lut_len or n_luts == 0:
            raise ValueError("luts is not a whole number of GLWE accumulators")
        idx = np.ascontiguousarray(np.broadcast_to(np.asarray(lut_idx, dtype=np.uint32), (B,)))
        return luts, pl, n_luts, idx, idx.ctypes.data_as(C.c_void_p)

    def pbs_batch(self, lwe_in, luts, lut_idx):
        p = self.params
        lwe_in, pi = _np_u64(lwe_in)
        B = lwe_in.size // p.small_size
        if lwe_in.size != B * p.small_size:
            raise ValueError("input is not a whole number of small LWE ciphertexts")
        luts, pl, n_luts, idx, px = self._luts(luts, lut_idx, B)
        out = np.zeros((B, p.big_size), dtype=np.uint64)
        check(_native.lib().tfhe_hip_pbs_batch(self._h, pi, pl, n_luts, px, out.ctypes.data_as(C.c_void_p), B))
        return out

    def ks_pbs_batch(self, lwe_in, luts, lut_idx, out=None):
        """`out`: optional preallocated (B, big_size) uint64 array, e.g. a PinnedArray's"""
        p = self.params
        lwe_in, pi = _np_u64(lwe_in)
        B = lwe_in.size // p.big_size
        if lwe_in.size != B * p.big_size:
            raise ValueError("input is not a whole number of big LWE ciphertexts")
        luts, pl, n_luts, idx, px = self._luts(luts, lut_idx, B)
        if out is None:
            out = np.zeros((B, p.big_size), dtype=np.uint64)
        elif out.dtype != np.uint64 or out.shape != (B, p.big_size) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (B, big_size) uint64 array")
        check(_native.lib().tfhe_hip_ks_pbs_batch(self._h, pi, pl, n_luts, px, out.ctypes.data_as(C.c_void_p), B))
        return out

    def pbs_ks_batch(self, lwe_in, luts, lut_idx):
        """BootstrapKeyswitch order: small LWE in, small LWE out."""
        p = self.params
        lwe_in, pi = _np_u64(lwe_in)
        B = lwe_in.size // p.small_size
        if lwe_in.size != B * p.small_size:
            raise ValueError("input is not a whole number of small LWE ciphertexts")
        luts, pl, n_luts, idx, px = self._luts(luts, lut_idx, B)
        out = np.zeros((B, p.small_size), dtype=np.uint64)
        check(_native.lib().tfhe_hip_pbs_ks_batch(self._h, pi, pl, n_luts, px, out.ctypes.data_as(C.c_void_p), B))
        return out

    def apply_lookup_table_batch(self, lwe_in, luts, lut_idx):
        """`ServerKey::apply_lookup_table` (crypto part) in the key's own PBSOrder."""
        p = self.params
        size = p.big_size if p.pbs_order == 0 else p.small_size
        lwe_in, pi = _np_u64(lwe_in)
        B = lwe_in.size // size
        if lwe_in.size != B * size:
            raise ValueError("input is not a whole number of ciphertexts of the key's encryption size")
        luts, pl, n_luts, idx, px = self._luts(luts, lut_idx, B)
        out = np.zeros((B, size), dtype=np.uint64)
        check(_native.lib().tfhe_hip_apply_lookup_table_batch(self._h, pi, pl, n_luts, px, out.ctypes.data_as(C.c_void_p), B))
        return out

    # ---- device-buffer path (torch tensors: int64 storage of the u64 torus, int32 lut_idx) ----
    def ks_pbs_batch_device(self, d_in, d_luts, d_idx, d_out, stream=None):
        """All arguments are contiguous torch CUDA tensors on this context's GPU:
        d_in/d_out int64 [B, kN+1], d_luts int64 [L, (k+1)N], d_idx int32 [B].  Enqueued on
        `stream` (default: torch's current stream); not synchronised."""
        import torch

        p = self.params
        B = d_in.shape[0]
        assert d_in.is_cuda and d_in.is_contiguous() and d_in.dtype == torch.int64 and d_in.shape[1] == p.big_size
        assert d_out.is_cuda and d_out.is_contiguous() and d_out.dtype == torch.int64 and tuple(d_out.shape) == (B, p.big_size)
        assert d_luts.is_cuda and d_luts.is_contiguous() and d_luts.dtype == torch.int64 and d_luts.shape[1] == p.lut_len
        assert d_idx.is_cuda and d_idx.is_contiguous() and d_idx.dtype == torch.int32 and d_idx.shape[0] == B
        s = torch.cuda.current_stream(d_in.device).cuda_stream if stream is None else stream
        check(_native.lib().tfhe_hip_ks_pbs_batch_device(self._h, d_in.data_ptr(), d_luts.data_ptr(), d_luts.shape[0],
                                                         d_idx.data_ptr(), d_out.data_ptr(), B, s))

    def keyswitch_batch_device(self, d_in, d_out, stream=None):
        import torch

        s = torch.cuda.current_stream(d_in.device).cuda_stream if stream is None else stream
        check(_native.lib().tfhe_hip_keyswitch_batch_device(self._h, d_in.data_ptr(), d_out.data_ptr(), d_in.shape[0], s))

    def pbs_batch_device(self, d_in, d_luts, d_idx, d_out, stream=None):
        import torch

        s = torch.cuda.current_stream(d_in.device).cuda_stream if stream is None else stream
        check(_native.lib().tfhe_hip_pbs_batch_device(self._h, d_in.data_ptr(), d_luts.data_ptr(), d_luts.shape[0],
                                                      d_idx.data_ptr(), d_out.data_ptr(), d_in.shape[0], s))

    # ---- measurement / introspection ----------------------------------------------------------
    def profiling(self, on=True):
        check(_native.lib().tfhe_hip_profiling_enable(self._h, 1 if on else 0))

    def profiling_read(self, kernel, reset=False):
        ms = C.c_double()
        n = C.c_uint64()
        check(_native.lib().tfhe_hip_profiling_read(self._h, kernel, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def fourier_bsk(self):
        p = self.params
        n = p.bsk_len // 2
        re = np.empty(n)
        im = np.empty(n)
        check(_native.lib().tfhe_hip_server_key_fourier_bsk(self._h, re.ctypes.data_as(C.c_void_p),
                                                            im.ctypes.data_as(C.c_void_p), n))
        return re, im

    @property
    def handle(self):
        """opaque tfhe_hip_server_key* (for the host op-graph layer, strings.HostKey)"""
        return self._h

    def set_variant(self, v):
        check(_native.lib().tfhe_hip_set_variant(self._h, v))

    def device_status(self, clear=False):
        """the key's device error word (0 = every launch so far produced valid results); include/tfhe_hip.h"""
        code = C.c_uint32()
        check(_native.lib().tfhe_hip_server_key_device_status(self._h, C.byref(code), 1 if clear else 0))
        return code.value

    def release_stream(self, stream):
        """hands the key's per-stream workspace of `stream` (a raw hipStream_t) back; returns the workspaces still held"""
        left = C.c_size_t()
        check(_native.lib().tfhe_hip_server_key_release_stream(self._h, C.c_void_p(stream), C.byref(left)))
        return left.value

    def debug_skip_pair_signal(self, step):
        check(_native.lib().tfhe_hip_debug_skip_pair_signal(self._h, step))


class PinnedArray:
    """A uint64 numpy array in page-locked host memory (tfhe_hip_host_alloc): host-buffer entry points skip the
    staging copy for such buffers and pipeline batches of several thousand ciphertexts."""

    def __init__(self, shape):
        n = int(np.prod(shape))
        self._p = C.c_void_p()
        check(_native.lib().tfhe_hip_host_alloc(C.byref(self._p), max(n, 1) * 8))
        buf = (C.c_uint64 * n).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=np.uint64).reshape(shape)

    def close(self):
        if getattr(self, "_p", None) and self._p.value:
            self.array = None
            _native.lib().tfhe_hip_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
