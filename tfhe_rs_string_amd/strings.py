"""Host op-graph layer (include/tfhe_hip_strings.h): the reference's callers of the KS-PBS path,
level-batched.  Class and method names follow the reference:

  ShortintServerKey.apply_lookup_table        tfhe/src/shortint/server_key/mod.rs:457
  IntegerServerKey.eq_parallelized / mul_parallelized
                                              tfhe/src/integer/server_key/radix_parallel/comparison.rs:207, mul.rs:553
  StringServerKey.eq / contains               tfhe/examples/fhe_strings/server_key/comparisons.rs:20, contains.rs:18

Ciphertexts are numpy uint64 arrays: a block is one big LWE (kN+1), a radix integer is
[num_blocks, kN+1] (least significant block first), an unpadded FheString is [len, 4, kN+1].
"""
import contextlib
import ctypes as C

import numpy as np

from . import _native

_vp = C.c_void_p
_SIG = {
    "tfhe_hip_host_key_create": (C.c_int, [_vp, C.POINTER(_native.Params), C.POINTER(_vp)]),
    "tfhe_hip_host_key_destroy": (C.c_int, [_vp]),
    "tfhe_hip_host_key_batch_begin": (C.c_int, [_vp]),
    "tfhe_hip_host_key_batch_end": (C.c_int, [_vp]),
    "tfhe_hip_host_key_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
    "tfhe_hip_host_key_set_strict_outputs": (C.c_int, [_vp, C.c_int]),
    "tfhe_hip_host_key_set_num_blocks": (C.c_int, [_vp, C.c_size_t]),
    "tfhe_hip_host_key_num_blocks": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "tfhe_hip_host_key_output_metadata": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp]),
    "tfhe_hip_shortint_apply_lookup_table_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t]),
    "tfhe_hip_integer_eq": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_integer_mul": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_integer_compare": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp]),
    "tfhe_hip_integer_scalar_compare": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint64, C.c_int, _vp]),
    "tfhe_hip_integer_min_max": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp]),
    "tfhe_hip_integer_if_then_else": (C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_integer_add": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_integer_sub": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_integer_neg": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_integer_scalar_add": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint64, _vp]),
    "tfhe_hip_integer_scalar_mul": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint64, _vp]),
    "tfhe_hip_string_find": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, _vp, _vp]),
    "tfhe_hip_string_find_clear": (C.c_int, [_vp, _vp, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, _vp, _vp]),
    "tfhe_hip_string_len": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp]),
    "tfhe_hip_string_is_empty": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp]),
    "tfhe_hip_string_push_padding_to_end": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_eq_padded": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, C.c_int, _vp]),
    "tfhe_hip_string_starts_with_padded": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, C.c_int, _vp]),
    "tfhe_hip_string_contains_padded": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, C.c_int, _vp]),
    "tfhe_hip_string_find_padded": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, C.c_int, C.c_int, _vp, _vp]),
    "tfhe_hip_string_trim": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, _vp]),
    "tfhe_hip_string_trim_char": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint8, C.c_int, C.c_int, _vp, _vp]),
    "tfhe_hip_string_strip": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, _vp, _vp, _vp]),
    "tfhe_hip_string_strip_clear": (C.c_int, [_vp, _vp, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, _vp, _vp, _vp]),
    "tfhe_hip_string_le": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_ge": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_le_clear": (C.c_int, [_vp, _vp, C.c_size_t, C.c_char_p, C.c_size_t, _vp]),
    "tfhe_hip_string_ge_clear": (C.c_int, [_vp, _vp, C.c_size_t, C.c_char_p, C.c_size_t, _vp]),
    "tfhe_hip_string_to_uppercase": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_to_lowercase": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_eq_ignore_case": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_eq_ignore_case_clear": (C.c_int, [_vp, _vp, C.c_size_t, C.c_char_p, C.c_size_t, _vp]),
    "tfhe_hip_string_eq": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_contains": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_ne": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_starts_with": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_ends_with": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_eq_clear": (C.c_int, [_vp, _vp, C.c_size_t, C.c_char_p, C.c_size_t, _vp]),
    "tfhe_hip_string_contains_clear": (C.c_int, [_vp, _vp, C.c_size_t, C.c_char_p, C.c_size_t, _vp]),
    "tfhe_hip_string_starts_with_clear": (C.c_int, [_vp, _vp, C.c_size_t, C.c_char_p, C.c_size_t, _vp]),
    "tfhe_hip_string_ends_with_clear": (C.c_int, [_vp, _vp, C.c_size_t, C.c_char_p, C.c_size_t, _vp]),
    "tfhe_hip_string_contains_windows": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp]),
    "tfhe_hip_boolean_any": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_boolean_all": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_add": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, C.c_int, _vp, _vp]),
    "tfhe_hip_string_repeat": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_uint64, _vp, C.c_size_t, _vp, _vp]),
    "tfhe_hip_string_substring": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp, _vp]),
    "tfhe_hip_string_nth": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_uint64, _vp, _vp]),
    "tfhe_hip_string_cmux_empty": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp, _vp]),
    "tfhe_hip_string_split_max_parts": (C.c_int, [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_uint64, _vp]),
    "tfhe_hip_string_split": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_uint64, _vp,
                                        _vp, C.c_size_t, _vp, _vp]),
    "tfhe_hip_string_split_once": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_char_p, C.c_size_t, C.c_int, C.c_int, _vp, _vp, _vp]),
    "tfhe_hip_string_replace_max_len": (C.c_int, [C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, _vp]),
    "tfhe_hip_string_replace": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_char_p, C.c_size_t, C.c_int, _vp, C.c_char_p, C.c_size_t,
                                          C.c_int, C.c_uint64, _vp, _vp, C.c_size_t, _vp, _vp]),
    "tfhe_hip_string_trim_encrypted_char": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_int, C.c_int, _vp, _vp]),
    "tfhe_hip_string_push_padding_to_start": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "tfhe_hip_string_pop_non_zero_char": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, _vp]),
}
NUMBER_BLOCKS = 4  # fhe_strings/ciphertext.rs:14


def bind(lib):
    for name, (res, args) in _SIG.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib.tfhe_hip_last_error.restype = C.c_char_p
    return lib


def exported_symbols():
    return list(_SIG.keys())


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(_vp)


class HostKey:
    """tfhe_hip_host_key: the level-batching executor over one device server key."""

    def __init__(self, server_key_handle, params, lib=None):
        self._lib = bind(lib if lib is not None else _native.lib())
        self.params = params
        self.num_blocks = 4  # blocks of every length / index / count (set_num_blocks)
        self._h = _vp()
        self._check(self._lib.tfhe_hip_host_key_create(server_key_handle, C.byref(params), C.byref(self._h)))

    def _check(self, rc):
        if rc != 0:
            raise _native.TfheHipError(self._lib.tfhe_hip_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tfhe_hip_host_key_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self, reset=False):
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.tfhe_hip_host_key_stats(self._h, C.byref(a), C.byref(b), 1 if reset else 0))
        return {"pbs": a.value, "launches": b.value}

    def set_num_blocks(self, num_blocks):
        """num_blocks of the reference's gen_keys_number_blocks: lengths, indices and counts become `num_blocks`-block radix
        ciphertexts (strings up to 4^num_blocks - 1 chars); include/tfhe_hip_strings.h"""
        self._check(self._lib.tfhe_hip_host_key_set_num_blocks(self._h, num_blocks))
        self.num_blocks = int(num_blocks)

    def set_strict_outputs(self, on=True):
        """output contract of include/tfhe_hip_strings.h: True (default) = every result at noise level NOMINAL"""
        self._check(self._lib.tfhe_hip_host_key_set_strict_outputs(self._h, 1 if on else 0))

    def output_metadata(self, out):
        """(degrees, noise_levels) of the blocks of `out`, an array a call on this key returned (shortint::Ciphertext fields)"""
        a = np.asarray(out)
        n = a.size // self.params.big_size
        deg, noise = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        self._check(self._lib.tfhe_hip_host_key_output_metadata(self._h, a.ctypes.data_as(_vp), n, deg.ctypes.data_as(_vp),
                                                                noise.ctypes.data_as(_vp)))
        return deg, noise

    def _block(self):
        return np.zeros(self.params.big_size, dtype=np.uint64)

    def _in(self, a):
        a, ptr = _u64(a)
        if getattr(self, "_keepalive", None) is not None:
            self._keepalive.append(a)  # the C side reads the array at batch_end
        return a, ptr

    @contextlib.contextmanager
    def batch(self):
        """Stage many operations and run them together: inside the block every call returns at once with arrays
        that are filled when the block ends; independent operations share their GPU launches."""
        self._check(self._lib.tfhe_hip_host_key_batch_begin(self._h))
        self._keepalive = []
        try:
            yield self
        finally:
            try:
                self._check(self._lib.tfhe_hip_host_key_batch_end(self._h))
            finally:
                self._keepalive = None


class ShortintServerKey(HostKey):
    def apply_lookup_table(self, cts, degrees, noise_levels, luts, lut_degrees, lut_idx):
        """Batched `apply_lookup_table` with the trivial shortcut; returns (cts, degrees, noise)."""
        p = self.params
        cts = np.array(cts, dtype=np.uint64).reshape(-1, p.big_size)
        B = cts.shape[0]
        degrees = np.array(np.broadcast_to(degrees, (B,)), dtype=np.uint64)
        noise = np.array(np.broadcast_to(noise_levels, (B,)), dtype=np.uint64)
        luts, pl = self._in(luts)
        n_luts = luts.size // p.lut_len
        ld, pld = self._in(np.broadcast_to(lut_degrees, (n_luts,)))
        idx = np.ascontiguousarray(np.broadcast_to(np.asarray(lut_idx, dtype=np.uint32), (B,)))
        self._check(self._lib.tfhe_hip_shortint_apply_lookup_table_batch(
            self._h, cts.ctypes.data_as(_vp), degrees.ctypes.data_as(_vp), noise.ctypes.data_as(_vp), pl, pld, n_luts,
            idx.ctypes.data_as(_vp), B))
        return cts, degrees, noise


class IntegerServerKey(HostKey):
    def eq_parallelized(self, lhs, rhs):
        lhs, pa = self._in(lhs)
        rhs, pb = self._in(rhs)
        if rhs.size != lhs.size:
            raise ValueError("radix ciphertexts have different block counts")
        nb = lhs.size // self.params.big_size
        out = self._block()
        self._check(self._lib.tfhe_hip_integer_eq(self._h, pa, pb, nb, out.ctypes.data_as(_vp)))
        return out

    def mul_parallelized(self, lhs, rhs):
        lhs, pa = self._in(lhs)
        rhs, pb = self._in(rhs)
        if rhs.size != lhs.size:
            raise ValueError("radix ciphertexts have different block counts")
        nb = lhs.size // self.params.big_size
        out = np.zeros((nb, self.params.big_size), dtype=np.uint64)
        self._check(self._lib.tfhe_hip_integer_mul(self._h, pa, pb, nb, out.ctypes.data_as(_vp)))
        return out

    _CMP = {"gt": 0, "ge": 1, "lt": 2, "le": 3}

    def _nb(self, a):
        return a.size // self.params.big_size

    def _radix_out(self, nb):
        return np.zeros((nb, self.params.big_size), dtype=np.uint64)

    def compare_parallelized(self, lhs, rhs, op):
        """gt / ge / lt / le _parallelized; rhs may be a clear int (scalar_*_parallelized)"""
        lhs, pa = self._in(lhs)
        out = self._block()
        if isinstance(rhs, (int, np.integer)):
            self._check(self._lib.tfhe_hip_integer_scalar_compare(self._h, pa, self._nb(lhs), int(rhs), self._CMP[op], out.ctypes.data_as(_vp)))
        else:
            rhs, pb = self._in(rhs)
            if rhs.size != lhs.size:
                raise ValueError("radix ciphertexts have different block counts")
            self._check(self._lib.tfhe_hip_integer_compare(self._h, pa, pb, self._nb(lhs), self._CMP[op], out.ctypes.data_as(_vp)))
        return out

    def gt_parallelized(self, lhs, rhs):
        return self.compare_parallelized(lhs, rhs, "gt")

    def ge_parallelized(self, lhs, rhs):
        return self.compare_parallelized(lhs, rhs, "ge")

    def lt_parallelized(self, lhs, rhs):
        return self.compare_parallelized(lhs, rhs, "lt")

    def le_parallelized(self, lhs, rhs):
        return self.compare_parallelized(lhs, rhs, "le")

    def _bin(self, fn, lhs, rhs, *extra):
        lhs, pa = self._in(lhs)
        rhs, pb = self._in(rhs)
        if rhs.size != lhs.size:
            raise ValueError("radix ciphertexts have different block counts")
        out = self._radix_out(self._nb(lhs))
        self._check(fn(self._h, pa, pb, self._nb(lhs), *extra, out.ctypes.data_as(_vp)))
        return out

    def max_parallelized(self, lhs, rhs):
        return self._bin(self._lib.tfhe_hip_integer_min_max, lhs, rhs, 1)

    def min_parallelized(self, lhs, rhs):
        return self._bin(self._lib.tfhe_hip_integer_min_max, lhs, rhs, 0)

    def add_parallelized(self, lhs, rhs):
        return self._bin(self._lib.tfhe_hip_integer_add, lhs, rhs)

    def sub_parallelized(self, lhs, rhs):
        return self._bin(self._lib.tfhe_hip_integer_sub, lhs, rhs)

    def neg_parallelized(self, ct):
        ct, pa = self._in(ct)
        out = self._radix_out(self._nb(ct))
        self._check(self._lib.tfhe_hip_integer_neg(self._h, pa, self._nb(ct), out.ctypes.data_as(_vp)))
        return out

    def scalar_add_parallelized(self, lhs, scalar):
        lhs, pa = self._in(lhs)
        out = self._radix_out(self._nb(lhs))
        self._check(self._lib.tfhe_hip_integer_scalar_add(self._h, pa, self._nb(lhs), int(scalar), out.ctypes.data_as(_vp)))
        return out

    def scalar_mul_parallelized(self, lhs, scalar):
        lhs, pa = self._in(lhs)
        out = self._radix_out(self._nb(lhs))
        self._check(self._lib.tfhe_hip_integer_scalar_mul(self._h, pa, self._nb(lhs), int(scalar), out.ctypes.data_as(_vp)))
        return out

    def if_then_else_parallelized(self, condition, true_ct, false_ct):
        condition, pc = self._in(condition)
        true_ct, pt = self._in(true_ct)
        false_ct, pf = self._in(false_ct)
        out = self._radix_out(self._nb(true_ct))
        self._check(self._lib.tfhe_hip_integer_if_then_else(self._h, pc, pt, pf, self._nb(true_ct), out.ctypes.data_as(_vp)))
        return out

    cmux_parallelized = if_then_else_parallelized

    def boolean_any(self, blocks):
        blocks, pb = self._in(blocks)
        out = self._block()
        self._check(self._lib.tfhe_hip_boolean_any(self._h, pb, blocks.size // self.params.big_size, out.ctypes.data_as(_vp)))
        return out

    def boolean_all(self, blocks):
        blocks, pb = self._in(blocks)
        out = self._block()
        self._check(self._lib.tfhe_hip_boolean_all(self._h, pb, blocks.size // self.params.big_size, out.ctypes.data_as(_vp)))
        return out


class StringServerKey(IntegerServerKey):
    def _len(self, s):
        n, rest = divmod(s.size, NUMBER_BLOCKS * self.params.big_size)
        if rest:
            raise ValueError("an encrypted string is a whole number of 4-block characters of big LWE ciphertexts")
        return n

    def eq(self, s1, s2):
        s1, p1 = self._in(s1)
        s2, p2 = self._in(s2)
        out = self._block()
        self._check(self._lib.tfhe_hip_string_eq(self._h, p1, self._len(s1), p2, self._len(s2), out.ctypes.data_as(_vp)))
        return out

    def contains(self, s, pattern):
        s, p1 = self._in(s)
        pattern, p2 = self._in(pattern)
        out = self._block()
        self._check(self._lib.tfhe_hip_string_contains(self._h, p1, self._len(s), p2, self._len(pattern), out.ctypes.data_as(_vp)))
        return out

    def _enc(self, fn, s, pattern):
        s, p1 = self._in(s)
        pattern, p2 = self._in(pattern)
        out = self._block()
        self._check(fn(self._h, p1, self._len(s), p2, self._len(pattern), out.ctypes.data_as(_vp)))
        return out

    def _clear(self, fn, s, pattern):
        s, p1 = self._in(s)
        raw = pattern.encode("ascii") if isinstance(pattern, str) else bytes(pattern)
        out = self._block()
        self._check(fn(self._h, p1, self._len(s), raw, len(raw), out.ctypes.data_as(_vp)))
        return out

    def ne(self, s1, s2):
        return self._enc(self._lib.tfhe_hip_string_ne, s1, s2)

    def starts_with(self, s, pattern):
        """pattern: encrypted FheString array, or a clear str (the reference's FhePattern for &str)"""
        if isinstance(pattern, (str, bytes)):
            return self._clear(self._lib.tfhe_hip_string_starts_with_clear, s, pattern)
        return self._enc(self._lib.tfhe_hip_string_starts_with, s, pattern)

    def ends_with(self, s, pattern):
        if isinstance(pattern, (str, bytes)):
            return self._clear(self._lib.tfhe_hip_string_ends_with_clear, s, pattern)
        return self._enc(self._lib.tfhe_hip_string_ends_with, s, pattern)

    # ---- strings with padding: padding = (string's, pattern's); 0 = None, 1 = Final, 2 = Initial,
    # 3 = InitialAndFinal, 4 = Anywhere (2-4 are normalised with push_padding_to_end first)
    def _padded(self, fn, s, pattern, padding):
        s, p1 = self._in(s)
        pattern, p2 = self._in(pattern)
        out = self._block()
        self._check(fn(self._h, p1, self._len(s), int(padding[0]), p2, self._len(pattern), int(padding[1]), out.ctypes.data_as(_vp)))
        return out

    def len(self, s, padding=4):
        """encrypted length (4-block radix) of a padded string: the number of non-zero characters"""
        s, p1 = self._in(s)
        out = np.zeros((self.num_blocks, self.params.big_size), dtype=np.uint64)
        self._check(self._lib.tfhe_hip_string_len(self._h, p1, self._len(s), int(padding), out.ctypes.data_as(_vp)))
        return out

    def is_empty(self, s, padding=4):
        s, p1 = self._in(s)
        out = self._block()
        self._check(self._lib.tfhe_hip_string_is_empty(self._h, p1, self._len(s), int(padding), out.ctypes.data_as(_vp)))
        return out

    def push_padding_to_end(self, s):
        return self._case(self._lib.tfhe_hip_string_push_padding_to_end, s)

    def eq_padded(self, s1, s2, padding=(1, 1)):
        return self._padded(self._lib.tfhe_hip_string_eq_padded, s1, s2, padding)

    def starts_with_padded(self, s, prefix, padding=(1, 1)):
        return self._padded(self._lib.tfhe_hip_string_starts_with_padded, s, prefix, padding)

    def contains_padded(self, s, pattern, padding=(1, 1)):
        return self._padded(self._lib.tfhe_hip_string_contains_padded, s, pattern, padding)

    def find_padded(self, s, pattern, padding=(1, 1), reverse=False):
        s, p1 = self._in(s)
        pattern, p2 = self._in(pattern)
        found = self._block()
        index = np.zeros((self.num_blocks, self.params.big_size), dtype=np.uint64)
        self._check(self._lib.tfhe_hip_string_find_padded(self._h, p1, self._len(s), int(padding[0]), p2, self._len(pattern),
                                                          int(padding[1]), int(reverse), found.ctypes.data_as(_vp),
                                                          index.ctypes.data_as(_vp)))
        return found, index

    # ---- trim / strip: (content with padding zeros, encrypted 4-block length)
    def _content_out(self, n):
        return np.zeros((n, 4, self.params.big_size), dtype=np.uint64), np.zeros((self.num_blocks, self.params.big_size), dtype=np.uint64)

    def _trim(self, s, mode):
        s, p1 = self._in(s)
        content, length = self._content_out(self._len(s))
        self._check(self._lib.tfhe_hip_string_trim(self._h, p1, self._len(s), mode, content.ctypes.data_as(_vp), length.ctypes.data_as(_vp)))
        return content, length

    def trim_start(self, s):
        return self._trim(s, 0)

    def trim_end(self, s):
        return self._trim(s, 1)

    def trim(self, s):
        return self._trim(s, 2)

    def trim_char(self, s, ch, left=True, right=True):
        s, p1 = self._in(s)
        content, length = self._content_out(self._len(s))
        self._check(self._lib.tfhe_hip_string_trim_char(self._h, p1, self._len(s), ord(ch) if isinstance(ch, str) else int(ch),
                                                        int(left), int(right), content.ctypes.data_as(_vp), length.ctypes.data_as(_vp)))
        return content, length

    def trim_encrypted_char(self, s, ch, left=True, right=True):
        """trim_start_encrypted / trim_end_encrypted / trim_encrypted: `ch` is an encrypted char (4 blocks)"""
        s, p1 = self._in(s)
        ch, pc = self._in(ch)
        content, length = self._content_out(self._len(s))
        self._check(self._lib.tfhe_hip_string_trim_encrypted_char(self._h, p1, self._len(s), pc, int(left), int(right),
                                                                  content.ctypes.data_as(_vp), length.ctypes.data_as(_vp)))
        return content, length

    def push_padding_to_start(self, s):
        return self._case(self._lib.tfhe_hip_string_push_padding_to_start, s)

    def pop_non_zero_char(self, s, last=False):
        """pop_first_non_zero_char / pop_last_non_zero_char -> (char [4, big], content)"""
        s, p1 = self._in(s)
        ch = np.zeros((NUMBER_BLOCKS, self.params.big_size), dtype=np.uint64)
        content = np.zeros((self._len(s), NUMBER_BLOCKS, self.params.big_size), dtype=np.uint64)
        self._check(self._lib.tfhe_hip_string_pop_non_zero_char(self._h, p1, self._len(s), int(last), ch.ctypes.data_as(_vp),
                                                                content.ctypes.data_as(_vp)))
        return ch, content

    def _strip(self, s, pattern, suffix):
        s, p1 = self._in(s)
        found = self._block()
        content, length = self._content_out(self._len(s))
        outs = (found.ctypes.data_as(_vp), content.ctypes.data_as(_vp), length.ctypes.data_as(_vp))
        if isinstance(pattern, (str, bytes)):
            raw = pattern.encode("ascii") if isinstance(pattern, str) else bytes(pattern)
            self._check(self._lib.tfhe_hip_string_strip_clear(self._h, p1, self._len(s), raw, len(raw), suffix, *outs))
        else:
            pattern, p2 = self._in(pattern)
            self._check(self._lib.tfhe_hip_string_strip(self._h, p1, self._len(s), p2, self._len(pattern), suffix, *outs))
        return found, content, length

    def strip_prefix(self, s, pattern):
        """FheOptionString: (found, content, encrypted length)"""
        return self._strip(s, pattern, 0)

    def strip_suffix(self, s, pattern):
        return self._strip(s, pattern, 1)

    def _find(self, s, pattern, reverse):
        s, p1 = self._in(s)
        found = self._block()
        index = np.zeros((self.num_blocks, self.params.big_size), dtype=np.uint64)
        if isinstance(pattern, (str, bytes)):
            raw = pattern.encode("ascii") if isinstance(pattern, str) else bytes(pattern)
            self._check(self._lib.tfhe_hip_string_find_clear(self._h, p1, self._len(s), raw, len(raw), reverse,
                                                             found.ctypes.data_as(_vp), index.ctypes.data_as(_vp)))
        else:
            pattern, p2 = self._in(pattern)
            self._check(self._lib.tfhe_hip_string_find(self._h, p1, self._len(s), p2, self._len(pattern), reverse,
                                                       found.ctypes.data_as(_vp), index.ctypes.data_as(_vp)))
        return found, index

    def find(self, s, pattern):
        """FheOptionInt: (found BooleanBlock, index as a 4-block radix ciphertext)"""
        return self._find(s, pattern, 0)

    def rfind(self, s, pattern):
        return self._find(s, pattern, 1)

    def le(self, s, pattern):
        if isinstance(pattern, (str, bytes)):
            return self._clear(self._lib.tfhe_hip_string_le_clear, s, pattern)
        return self._enc(self._lib.tfhe_hip_string_le, s, pattern)

    def ge(self, s, pattern):
        if isinstance(pattern, (str, bytes)):
            return self._clear(self._lib.tfhe_hip_string_ge_clear, s, pattern)
        return self._enc(self._lib.tfhe_hip_string_ge, s, pattern)

    def eq_ignore_case(self, s, pattern):
        if isinstance(pattern, (str, bytes)):
            return self._clear(self._lib.tfhe_hip_string_eq_ignore_case_clear, s, pattern)
        return self._enc(self._lib.tfhe_hip_string_eq_ignore_case, s, pattern)

    def _case(self, fn, s):
        s, p1 = self._in(s)
        n = self._len(s)
        out = np.zeros((n, 4, self.params.big_size), dtype=np.uint64)
        self._check(fn(self._h, p1, n, out.ctypes.data_as(_vp)))
        return out

    def to_uppercase(self, s):
        return self._case(self._lib.tfhe_hip_string_to_uppercase, s)

    def to_lowercase(self, s):
        return self._case(self._lib.tfhe_hip_string_to_lowercase, s)

    def eq_clear(self, s, clear):
        return self._clear(self._lib.tfhe_hip_string_eq_clear, s, clear)

    def contains_clear(self, s, clear):
        return self._clear(self._lib.tfhe_hip_string_contains_clear, s, clear)

    def window_count(self, s, pattern):
        a, b = self._len(np.asarray(s)), self._len(np.asarray(pattern))
        return 0 if b > a else a - b + 1

    def contains_windows(self, s, pattern, w_begin, w_end):
        s, p1 = self._in(s)
        pattern, p2 = self._in(pattern)
        out = np.zeros((w_end - w_begin, self.params.big_size), dtype=np.uint64)
        self._check(self._lib.tfhe_hip_string_contains_windows(self._h, p1, self._len(s), p2, self._len(pattern), w_begin, w_end,
                                                               out.ctypes.data_as(_vp)))
        return out


    # ---- concatenation, repeat, substring, nth (add.rs, repeat.rs, substring.rs, accessors.rs) ----------------
    # strings with a padding kind are (array, padding) pairs on return; padding: 0 None, 1 Final, 2 Initial,
    # 3 InitialAndFinal, 4 Anywhere
    def _chars(self, n):
        return np.zeros((n, NUMBER_BLOCKS, self.params.big_size), dtype=np.uint64)

    def _pattern(self, pattern):
        """-> (keepalive, encrypted pointer or None, clear bytes or None, length)"""
        if isinstance(pattern, (str, bytes)):
            raw = pattern.encode("ascii") if isinstance(pattern, str) else bytes(pattern)
            return None, None, raw, len(raw)
        a, ptr = self._in(pattern)
        return a, ptr, None, self._len(a)

    def _int_arg(self, n):
        """a clear int or an encrypted num_blocks-block radix -> (keepalive, clear value, pointer or None)"""
        if isinstance(n, (int, np.integer)):
            return None, int(n), None
        a, ptr = self._in(n)
        if a.size != self.num_blocks * self.params.big_size:
            raise ValueError(f"encrypted integer arguments of the string layer are {self.num_blocks}-block radix ciphertexts (set_num_blocks)")
        return a, 0, ptr

    def add(self, s1, s2, padding=(0, 0)):
        s1, p1 = self._in(s1)
        s2, p2 = self._in(s2)
        out, pad = self._chars(self._len(s1) + self._len(s2)), C.c_int()
        self._check(self._lib.tfhe_hip_string_add(self._h, p1, self._len(s1), int(padding[0]), p2, self._len(s2), int(padding[1]),
                                                  out.ctypes.data_as(_vp), C.byref(pad)))
        return out, pad.value

    def repeat(self, s, n, n_max=None, padding=0):
        """repeat_clear (n: int) / repeat_encrypted (n: 4-block radix, n_max copies of room)"""
        s, p1 = self._in(s)
        keep, n_clear, n_ptr = self._int_arg(n)
        if n_ptr is not None and n_max is None:
            raise ValueError("repeat with an encrypted count needs n_max")
        out, pad = self._chars(self._len(s) * (int(n_max) if n_ptr is not None else n_clear)), C.c_int()
        self._check(self._lib.tfhe_hip_string_repeat(self._h, p1, self._len(s), int(padding), n_clear, n_ptr, int(n_max or 0),
                                                     out.ctypes.data_as(_vp), C.byref(pad)))
        return out, pad.value

    def substring(self, s, start, end, padding=0):
        """-> (in_range block, content, padding).  start / end: both ints or both 4-block radix ciphertexts"""
        s, p1 = self._in(s)
        ka, a, pa = self._int_arg(start)
        kb, b, pb = self._int_arg(end)
        if (pa is None) != (pb is None):
            raise ValueError("substring: start and end are both clear or both encrypted")
        if pa is None and not (0 <= a <= b <= self._len(s)):
            raise ValueError("substring: the range is not inside the content (ConversionError::OutOfRange)")
        out, pad, ok = self._chars(self._len(s) if pa is not None else b - a), C.c_int(), self._block()
        self._check(self._lib.tfhe_hip_string_substring(self._h, p1, self._len(s), int(padding), a, b, pa, pb, ok.ctypes.data_as(_vp),
                                                        out.ctypes.data_as(_vp), C.byref(pad)))
        return ok, out, pad.value

    def nth(self, s, n, padding=0):
        s, p1 = self._in(s)
        keep, n_clear, n_ptr = self._int_arg(n)
        out = np.zeros((NUMBER_BLOCKS, self.params.big_size), dtype=np.uint64)
        self._check(self._lib.tfhe_hip_string_nth(self._h, p1, self._len(s), int(padding), n_clear, n_ptr, out.ctypes.data_as(_vp)))
        return out

    def cmux_empty_string(self, condition, s, padding=0):
        s, p1 = self._in(s)
        c, pc = self._in(condition)
        out, pad = self._chars(self._len(s)), C.c_int()
        self._check(self._lib.tfhe_hip_string_cmux_empty(self._h, pc, p1, self._len(s), int(padding), out.ctypes.data_as(_vp), C.byref(pad)))
        return out, pad.value

    # ---- the split family -> FheSplit (parts [n_parts, len, 4, big], number_parts [4, big]) ---------------------
    _SPLIT_KINDS = {"split": 0, "rsplit": 1, "split_terminator": 2, "rsplit_terminator": 3, "split_inclusive": 4, "splitn": 5,
                    "rsplitn": 6, "split_ascii_whitespace": 7}

    @staticmethod
    def _paddings(padding):
        """padding: the string's kind, or (string's, pattern's)"""
        return (int(padding[0]), int(padding[1])) if isinstance(padding, (tuple, list)) else (int(padding), 0)

    def _split(self, kind, s, pattern, n=0, padding=0):
        s, p1 = self._in(s)
        L = self._len(s)
        pad, ppad = self._paddings(padding)
        kp, pat_ptr, pat_raw, pat_len = self._pattern(pattern) if pattern is not None else (None, None, None, 0)
        kn, n_clear, n_ptr = self._int_arg(n)
        k = self._SPLIT_KINDS[kind]
        cap = C.c_size_t()
        self._check(self._lib.tfhe_hip_string_split_max_parts(k, L, pat_len, ppad if pat_ptr is not None else 0, int(n_ptr is not None),
                                                              n_clear, C.byref(cap)))
        parts = np.zeros((cap.value, L, NUMBER_BLOCKS, self.params.big_size), dtype=np.uint64)
        number = np.zeros((self.num_blocks, self.params.big_size), dtype=np.uint64)
        got = C.c_size_t()
        self._check(self._lib.tfhe_hip_string_split(self._h, p1, L, pad, pat_ptr, pat_raw, pat_len, ppad, k, n_clear, n_ptr,
                                                    parts.ctypes.data_as(_vp), cap.value, number.ctypes.data_as(_vp), C.byref(got)))
        return parts[:got.value], number

    def split(self, s, pattern, padding=0):
        return self._split("split", s, pattern, padding=padding)

    def rsplit(self, s, pattern, padding=0):
        return self._split("rsplit", s, pattern, padding=padding)

    def split_terminator(self, s, pattern, padding=0):
        return self._split("split_terminator", s, pattern, padding=padding)

    def rsplit_terminator(self, s, pattern, padding=0):
        return self._split("rsplit_terminator", s, pattern, padding=padding)

    def split_inclusive(self, s, pattern, padding=0):
        return self._split("split_inclusive", s, pattern, padding=padding)

    def splitn(self, s, n, pattern, padding=0):
        return self._split("splitn", s, pattern, n=n, padding=padding)

    def rsplitn(self, s, n, pattern, padding=0):
        return self._split("rsplitn", s, pattern, n=n, padding=padding)

    def split_ascii_whitespace(self, s, padding=0):
        return self._split("split_ascii_whitespace", s, None, padding=padding)

    def split_once(self, s, pattern, reverse=False, padding=0):
        """-> (found block, before, after)"""
        s, p1 = self._in(s)
        L = self._len(s)
        kp, pat_ptr, pat_raw, pat_len = self._pattern(pattern)
        pad, ppad = self._paddings(padding)
        found, before, after = self._block(), self._chars(L), self._chars(L)
        self._check(self._lib.tfhe_hip_string_split_once(self._h, p1, L, pad, pat_ptr, pat_raw, pat_len, ppad, int(reverse),
                                                         found.ctypes.data_as(_vp), before.ctypes.data_as(_vp), after.ctypes.data_as(_vp)))
        return found, before, after

    def rsplit_once(self, s, pattern, padding=0):
        return self.split_once(s, pattern, reverse=True, padding=padding)

    def replace(self, s, old, new, n=None, padding=0):
        """replace (n None) / replacen (n: int or 4-block radix) -> (content, padding)"""
        s, p1 = self._in(s)
        L = self._len(s)
        kf, f_ptr, f_raw, f_len = self._pattern(old)
        kt, t_ptr, t_raw, t_len = self._pattern(new)
        kn, n_clear, n_ptr = self._int_arg(0 if n is None else n)
        cap = C.c_size_t()
        spad, fpad = self._paddings(padding)
        fpad = fpad if f_ptr is not None else 0
        self._check(self._lib.tfhe_hip_string_replace_max_len(L, f_len, fpad, t_len, C.byref(cap)))
        out, got, pad = self._chars(cap.value), C.c_size_t(), C.c_int()
        self._check(self._lib.tfhe_hip_string_replace(self._h, p1, L, spad, f_ptr, f_raw, f_len, fpad, t_ptr, t_raw, t_len,
                                                      int(n is not None), n_clear, n_ptr, out.ctypes.data_as(_vp), cap.value,
                                                      C.byref(got), C.byref(pad)))
        return out[:got.value], pad.value

    def replacen(self, s, old, new, n, padding=0):
        return self.replace(s, old, new, n=n, padding=padding)


def shard_range(total, rank, world):
    """Contiguous shard [begin, end) of `total` independent units for `rank` of `world`
    (ceil split: the first ranks take one more unit)."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def contains_sharded(key, s, pattern, dist=None, device=None, timings=None):
    """`StringServerKey::contains` with the windows sharded across the ranks of a
    torch.distributed group (one rank per GPU): every rank evaluates its windows, the per-window
    booleans (one big LWE each) are all-gathered -- the path's only exchange -- and every rank
    finishes with the same OR tree.  With dist=None it is the single-GPU op.
    `timings` (a dict, optional) receives this rank's window range and the seconds spent in its windows, in the
    gather and in the final tree."""
    import time

    import torch

    p = key.params
    nw = key.window_count(s, pattern)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1 or nw == 0:
        t0 = time.perf_counter()
        out = key.contains(s, pattern)
        if timings is not None:
            timings.update(windows=(0, nw), windows_s=time.perf_counter() - t0, gather_s=0.0, tree_s=0.0)
        return out
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-nw // world)
    b, e = shard_range(nw, rank, world)
    t0 = time.perf_counter()
    mine = key.contains_windows(s, pattern, b, e)
    t1 = time.perf_counter()
    buf = np.zeros((per, p.big_size), dtype=np.int64)
    buf[: e - b] = mine.view(np.int64)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    blocks = []
    for r in range(world):
        rb, re_ = shard_range(nw, r, world)
        blocks.append(gathered[r][: re_ - rb].cpu().numpy().view(np.uint64))
    t2 = time.perf_counter()
    out = key.boolean_any(np.concatenate(blocks))
    if timings is not None:
        timings.update(windows=(b, e), windows_s=t1 - t0, gather_s=t2 - t1, tree_s=time.perf_counter() - t2)
    return out


def column_sharded(key, op, column, dist=None, device=None):
    """A predicate over a COLUMN of encrypted strings (`op(key, item)` -> one boolean block), the items sharded across
    the ranks of a torch.distributed group (one rank per GPU): every rank evaluates its contiguous shard in ONE batch
    (`key.batch()`: the shard's op-graphs share their launches), the result blocks (one big LWE per item) are
    all-gathered -- the only exchange -- and every rank returns the full list in column order.  With dist=None it is
    the single-GPU batch."""
    import torch

    p = key.params
    n = len(column)
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    b, e = shard_range(n, rank, world)
    with key.batch():
        mine = [op(key, column[i]) for i in range(b, e)]
    if world == 1:
        return mine
    per = -(-n // world)
    buf = np.zeros((per, p.big_size), dtype=np.int64)
    for i, blk in enumerate(mine):
        buf[i] = np.asarray(blk, dtype=np.uint64).reshape(-1).view(np.int64)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    out = []
    for r in range(world):
        rb, re_ = shard_range(n, r, world)
        rows = gathered[r][: re_ - rb].cpu().numpy().view(np.uint64)
        out.extend(rows[i] for i in range(re_ - rb))
    return out


class ResultGather:
    """The path's only exchange: gather every rank's output LWEs (16 392 B each) on rank 0.

    Asynchronous and double-buffered so that the gather of batch i overlaps the bootstrap of batch
    i+1: `submit(out)` returns immediately; `out` must not be rewritten until the gather submitted
    two calls earlier has completed (the caller waits on `pending` before reusing a buffer).  RCCL
    `gather` is point-to-point sends to the root over xGMI links in parallel (no ring).
    """

    def __init__(self, dist, like, depth=2):
        import torch

        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.depth = depth
        self.pending = []
        self.slot = 0
        self.recv = None
        if self.rank == 0:
            self.recv = [[torch.empty_like(like) for _ in range(self.world)] for _ in range(depth)]

    def submit(self, out):
        while len(self.pending) >= self.depth:
            self.pending.pop(0).wait()
        glist = self.recv[self.slot] if self.rank == 0 else None
        self.slot = (self.slot + 1) % self.depth
        self.pending.append(self.dist.gather(out, gather_list=glist, dst=0, async_op=True))
        return glist

    def drain(self):
        for h in self.pending:
            h.wait()
        self.pending = []
