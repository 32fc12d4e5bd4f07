/*
 * tfhe_hip_strings.h -- C-ABI of the host op-graph layer above include/tfhe_hip.h: the callers of
 * the KS-PBS hot path that BASELINE.json's configs name, re-scheduled so that every dependency level
 * is one batched GPU launch.
 *
 * Reference entry points mirrored (tfhe-rs 0.5.0 fork, /root/reference):
 *   shortint  ServerKey::apply_lookup_table               tfhe/src/shortint/server_key/mod.rs:457-476
 *   integer   ServerKey::eq_parallelized                  tfhe/src/integer/server_key/radix_parallel/comparison.rs:10-33, 207
 *             ServerKey::mul_parallelized                 tfhe/src/integer/server_key/radix_parallel/mul.rs:329-414, 553
 *   strings   StringServerKey::eq / contains              tfhe/examples/fhe_strings/server_key/comparisons.rs:20, contains.rs:18
 *
 * Data: a shortint block is one big LWE ciphertext (kN+1 u64); a radix integer is `num_blocks`
 * consecutive blocks, least significant first (integer/ciphertext/mod.rs:18-21); an unpadded
 * FheString of `len` chars is len x 4 blocks (FheAsciiChar = 4-block radix, fhe_strings/ciphertext.rs:14-18).
 * Inputs are fresh / clean ciphertexts (degree message_modulus - 1, nominal noise); boolean results
 * are one block (BooleanBlock, degree 1).
 * OUTPUT CONTRACT = the reference's (shortint/ciphertext/mod.rs:263-270, server_key/mod.rs:855-856): every block handed
 * back has clean carries and noise_level NOMINAL -- it is a direct bootstrap output (or the caller's own input block
 * passed through, or a trivial ciphertext: level 0) -- so a caller may do with it whatever tfhe-rs allows on a
 * NOMINAL ciphertext (15-term sums, 4a + b, ...).  A few op-graphs end in a LEVELED step (the selected side of a cmux /
 * min / max, a case-adjusted block, a clear replacement char): with `strict_outputs` (tfhe_hip_host_key_set_strict_outputs,
 * ON by default) those results go through one more, cleaning bootstrap; switching it off saves that launch and hands them
 * back at noise level 2-5.  Either way tfhe_hip_host_key_output_metadata reports the degree and noise level of every block a
 * call wrote, which is what a Rust wrapper copies into Ciphertext{degree, noise_level}.
 * Same conventions as tfhe_hip.h: 0 = ok, 1 = error with tfhe_hip_last_error().
 */
#ifndef TFHE_HIP_STRINGS_H
#define TFHE_HIP_STRINGS_H

#include "tfhe_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tfhe_hip_host_key tfhe_hip_host_key;

/* wraps a server key with the host-side bookkeeping (LUT cache, counters); does not own `sk`.  Operations on one host
 * key are serialised (an internal lock); for concurrent host threads create one host key per thread over the same
 * server key -- the device key is shared and tfhe_hip_ks_pbs_batch is re-entrant. */
int tfhe_hip_host_key_create(tfhe_hip_server_key *sk, const tfhe_hip_params *params, tfhe_hip_host_key **out);
int tfhe_hip_host_key_destroy(tfhe_hip_host_key *hk);
/* Output contract (top of this file).  strict_outputs: 1 (default) = every result block is handed back at noise level
 * NOMINAL, through a final cleaning bootstrap where the op-graph ends in a leveled step; 0 = such results come back as they
 * are (one launch less).  Cannot change while a batch is open. */
int tfhe_hip_host_key_set_strict_outputs(tfhe_hip_host_key *hk, int on);
/* num_blocks of the reference's gen_keys_number_blocks (tfhe/examples/fhe_strings/ciphertext.rs:113-126: "should be larger than
 * 4 to work with strings of length larger than 8 bits"): every LENGTH, INDEX and COUNT this layer takes or hands back
 * (find / rfind indices, len, the lengths of trim / strip, number_parts and n of the split family, n of replacen / repeat /
 * substring / nth) is a radix ciphertext of `num_blocks` blocks of 2 bits, least significant first; 4 (the default) .. 16.
 * Strings may then hold up to 4^num_blocks - 1 chars.  Chars stay 4-block radix ciphertexts (an ASCII char needs no more:
 * the reference's wider chars carry zero blocks on top).  Arrays that hold such a value must have num_blocks blocks. */
int tfhe_hip_host_key_set_num_blocks(tfhe_hip_host_key *hk, size_t num_blocks);
int tfhe_hip_host_key_num_blocks(tfhe_hip_host_key *hk, size_t *num_blocks);
/* degree and noise level (shortint::Ciphertext fields; NOMINAL = 1, trivial = 0) of `n_blocks` consecutive blocks at
 * `out_blocks` that the LAST call on this host key -- or, in batch mode, any call of the last batch -- wrote; either array may
 * be NULL.  Fails if an address was not written by that call / batch. */
int tfhe_hip_host_key_output_metadata(tfhe_hip_host_key *hk, const uint64_t *out_blocks, size_t n_blocks, uint64_t *degrees,
                                      uint64_t *noise_levels);
/* PBS executed on the GPU and batched launches issued since creation / last reset */
int tfhe_hip_host_key_stats(tfhe_hip_host_key *hk, uint64_t *pbs_count, uint64_t *launches, int reset);

/* Batch mode.  Between batch_begin and batch_end every operation below only stages its bootstraps and registers
 * its output arrays; batch_end runs the union of all the staged op-graphs level by level (independent
 * operations share their launches: the throughput regime of the kernels) and then writes every output.
 * All input and output arrays passed in between must stay valid, and outputs are undefined, until batch_end.
 * (The reference gets the same effect from rayon: independent operations on a thread pool.) */
int tfhe_hip_host_key_batch_begin(tfhe_hip_host_key *hk);
int tfhe_hip_host_key_batch_end(tfhe_hip_host_key *hk);

/* shortint::ServerKey::apply_lookup_table for B ciphertexts in place, with the trivial-ciphertext
 * shortcut and degree / noise bookkeeping: degrees[B], noise_levels[B] are updated; lut_degrees[n_luts]. */
int tfhe_hip_shortint_apply_lookup_table_batch(tfhe_hip_host_key *hk, uint64_t *cts, uint64_t *degrees,
                                               uint64_t *noise_levels, const uint64_t *luts,
                                               const uint64_t *lut_degrees, size_t n_luts, const uint32_t *lut_idx,
                                               size_t B);

/* integer::ServerKey::eq_parallelized -> BooleanBlock (one big LWE) */
int tfhe_hip_integer_eq(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                        uint64_t *out_block);
/* integer::ServerKey::mul_parallelized: out has num_blocks blocks with clean carries */
int tfhe_hip_integer_mul(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                         uint64_t *out_blocks);

/* integer::ServerKey::{gt,ge,lt,le}_parallelized on clean unsigned radix ciphertexts -> BooleanBlock
 * (integer/server_key/comparator.rs:389-470 packed-pair sign blocks, :257-283 sign tree, :957-972 map).
 * op: 0 = gt, 1 = ge, 2 = lt, 3 = le */
int tfhe_hip_integer_compare(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                             int op, uint64_t *out_block);
/* scalar_{gt,ge,lt,le}_parallelized (comparator.rs:677-783) */
int tfhe_hip_integer_scalar_compare(tfhe_hip_host_key *hk, const uint64_t *lhs, size_t num_blocks, uint64_t scalar,
                                    int op, uint64_t *out_block);
/* min_parallelized / max_parallelized (comparator.rs:849-875) */
int tfhe_hip_integer_min_max(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                             int want_max, uint64_t *out_blocks);
/* if_then_else_parallelized / cmux_parallelized (radix_parallel/cmux.rs:72-145): condition is a BooleanBlock */
int tfhe_hip_integer_if_then_else(tfhe_hip_host_key *hk, const uint64_t *condition, const uint64_t *true_ct,
                                  const uint64_t *false_ct, size_t num_blocks, uint64_t *out_blocks);
/* add_parallelized / sub_parallelized / neg_parallelized (radix_parallel/add.rs, sub.rs:162-243, neg.rs),
 * wrapping, clean inputs and outputs */
int tfhe_hip_integer_add(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                         uint64_t *out_blocks);
int tfhe_hip_integer_sub(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                         uint64_t *out_blocks);
int tfhe_hip_integer_neg(tfhe_hip_host_key *hk, const uint64_t *ct, size_t num_blocks, uint64_t *out_blocks);
/* scalar_add_parallelized / scalar_mul_parallelized (radix_parallel/scalar_add.rs, scalar_mul.rs) */
int tfhe_hip_integer_scalar_add(tfhe_hip_host_key *hk, const uint64_t *lhs, size_t num_blocks, uint64_t scalar,
                                uint64_t *out_blocks);
int tfhe_hip_integer_scalar_mul(tfhe_hip_host_key *hk, const uint64_t *lhs, size_t num_blocks, uint64_t scalar,
                                uint64_t *out_blocks);

/* StringServerKey::eq on two unpadded strings (clear lengths) */
int tfhe_hip_string_eq(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, const uint64_t *s2, size_t len2,
                       uint64_t *out_block);
/* StringServerKey::ne = boolean_bitnot(eq) */
int tfhe_hip_string_ne(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, const uint64_t *s2, size_t len2,
                       uint64_t *out_block);
/* StringServerKey::starts_with / ends_with with an encrypted pattern (unpadded strings)
 * (fhe_strings/server_key/starts_with.rs:8, ends_with.rs:18) */
int tfhe_hip_string_starts_with(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *prefix,
                                size_t prefix_len, uint64_t *out_block);
int tfhe_hip_string_ends_with(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *suffix,
                              size_t suffix_len, uint64_t *out_block);
/* the same operations with a CLEAR pattern (&str implements FhePattern: eq_clear, contains_clear_string,
 * starts_with_vec_clear; comparisons.rs:156-168, contains.rs:42-74, 136-149): every char comparison is
 * scalar_eq_parallelized(char, byte) = 2 packed-block PBS instead of 4 bivariate ones */
int tfhe_hip_string_eq_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                             size_t clear_len, uint64_t *out_block);
int tfhe_hip_string_contains_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                   size_t clear_len, uint64_t *out_block);
int tfhe_hip_string_starts_with_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                      size_t clear_len, uint64_t *out_block);
int tfhe_hip_string_ends_with_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                    size_t clear_len, uint64_t *out_block);
/* StringServerKey::le / ge (comparisons.rs:74-99, 299-330): byte-wise lexicographic order of unpadded
 * strings, clear lengths as tie-break; encrypted and clear (&str) right-hand sides */
int tfhe_hip_string_le(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, const uint64_t *s2, size_t len2,
                       uint64_t *out_block);
int tfhe_hip_string_ge(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, const uint64_t *s2, size_t len2,
                       uint64_t *out_block);
int tfhe_hip_string_le_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                             size_t clear_len, uint64_t *out_block);
int tfhe_hip_string_ge_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                             size_t clear_len, uint64_t *out_block);
/* StringServerKey::to_uppercase / to_lowercase (change_case.rs:18-83): out has len * 4 blocks */
int tfhe_hip_string_to_uppercase(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint64_t *out);
int tfhe_hip_string_to_lowercase(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint64_t *out);
/* StringServerKey::eq_ignore_case (comparisons.rs:55-72, 126-180) */
int tfhe_hip_string_eq_ignore_case(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, const uint64_t *s2,
                                   size_t len2, uint64_t *out_block);
int tfhe_hip_string_eq_ignore_case_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                         size_t clear_len, uint64_t *out_block);
/* StringServerKey::find / rfind (find.rs:25-46) -> FheOptionInt = (found BooleanBlock, index as a 4-block
 * radix ciphertext); strings up to 255 chars.  Not found: index = len (find) / 0 (rfind), as the
 * reference's loops leave it.  reverse != 0 selects rfind. */
int tfhe_hip_string_find(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *pattern,
                         size_t pattern_len, int reverse, uint64_t *out_found, uint64_t *out_index);
int tfhe_hip_string_find_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                               size_t clear_len, int reverse, uint64_t *out_found, uint64_t *out_index);
/* The same operations on strings with FINAL padding (ciphertext.rs:21-29 Padding::Final: zero chars after
 * the last character, as `encrypt_str_padding` produces them).  padding arguments: 0 = Padding::None,
 * 1 = Final, 2 = Initial, 3 = InitialAndFinal, 4 = Anywhere (`encrypt_str_random_padding`); strings with
 * padding 2-4 are first normalised with push_padding_to_end, as the reference does.  `len` is the
 * content length (characters + padding zeros).
 * (eq_no_init_padding comparisons.rs:184-212, starts_with_encrypted_no_init_padding starts_with.rs:22-62,
 * contains_unpadded_string contains.rs:77-130, connected_find_unpadded_string find.rs:164-184.)
 * le / ge / to_uppercase / to_lowercase above accept final padding as they are (a padding zero is the
 * smallest char and has no case); eq_ignore_case does for equal content lengths. */
/* StringServerKey::len / is_empty (is_empty.rs:22-62) for a string with padding (any kind): the encrypted number
 * of non-zero characters as a 4-block radix / the BooleanBlock "every character is a padding zero" */
int tfhe_hip_string_len(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, uint64_t *out_length);
int tfhe_hip_string_is_empty(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, uint64_t *out_block);
/* StringServerKey::push_padding_to_end (change_padding.rs:5-21): the characters in order, then the zeros;
 * out has len * 4 blocks (Padding::Final).  Strings up to 255 chars. */
int tfhe_hip_string_push_padding_to_end(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint64_t *out);
int tfhe_hip_string_eq_padded(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, int padding1, const uint64_t *s2,
                              size_t len2, int padding2, uint64_t *out_block);
int tfhe_hip_string_starts_with_padded(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding,
                                       const uint64_t *prefix, size_t prefix_len, int prefix_padding,
                                       uint64_t *out_block);
int tfhe_hip_string_contains_padded(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding,
                                    const uint64_t *pattern, size_t pattern_len, int pattern_padding,
                                    uint64_t *out_block);
int tfhe_hip_string_find_padded(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding,
                                const uint64_t *pattern, size_t pattern_len, int pattern_padding, int reverse,
                                uint64_t *out_found, uint64_t *out_index);
/* StringServerKey::trim_start (mode 0), trim_end (1), trim (2) (trim.rs:129-147, 202-226) on an unpadded
 * string: out_content has len * 4 blocks with padding zeros where characters were removed
 * (Padding::InitialAndFinal), out_length is the encrypted length (4 blocks).  As in the reference,
 * trim_start removes ASCII whitespace (9..13, 32), trim_end and trim remove the space character. */
int tfhe_hip_string_trim(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int mode, uint64_t *out_content,
                         uint64_t *out_length);
/* trim_start_char / trim_end_char / trim_char with a clear character (trim.rs:123-163) */
int tfhe_hip_string_trim_char(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint8_t character, int left,
                              int right, uint64_t *out_content, uint64_t *out_length);
/* StringServerKey::strip_prefix (suffix = 0) / strip_suffix (suffix = 1) (strip.rs:12-28) -> FheOptionString:
 * out_found, and the string with the pattern's characters zeroed if it was found (+ encrypted length) */
int tfhe_hip_string_strip(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *pattern,
                          size_t pattern_len, int suffix, uint64_t *out_found, uint64_t *out_content,
                          uint64_t *out_length);
int tfhe_hip_string_strip_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                size_t clear_len, int suffix, uint64_t *out_found, uint64_t *out_content,
                                uint64_t *out_length);
/* StringServerKey::contains (unpadded haystack and needle) */
int tfhe_hip_string_contains(tfhe_hip_host_key *hk, const uint64_t *hay, size_t hay_len, const uint64_t *needle,
                             size_t needle_len, uint64_t *out_block);
/* the shardable part of `contains`: one boolean block per window w in [w_begin, w_end)
 * ("does the needle start at haystack position w"); windows are independent, so ranks take disjoint
 * ranges and the booleans are gathered before tfhe_hip_boolean_any */
int tfhe_hip_string_contains_windows(tfhe_hip_host_key *hk, const uint64_t *hay, size_t hay_len,
                                     const uint64_t *needle, size_t needle_len, size_t w_begin, size_t w_end,
                                     uint64_t *out_blocks);
/* is_at_least_one_comparisons_block_true / are_all_comparisons_block_true over n boolean blocks */
int tfhe_hip_boolean_any(tfhe_hip_host_key *hk, const uint64_t *blocks, size_t n, uint64_t *out_block);
int tfhe_hip_boolean_all(tfhe_hip_host_key *hk, const uint64_t *blocks, size_t n, uint64_t *out_block);

/* ---- the rest of the FheString interface (tfhe/examples/fhe_strings/server_key/{add,repeat,substring,accessors,
 * split,rsplit,rsplitn,split_terminator,split_inclusive,split_ascii_whitespace,replace}.rs) --------------------
 * Strings are (content, len, padding) as above; any padding kind is accepted (Initial / InitialAndFinal / Anywhere
 * contents are first compacted by push_padding_to_end, as the reference does).  A pattern is passed EITHER
 * encrypted (`pattern`: pattern_len chars and its padding kind `pattern_padding`; with padding its length is encrypted;
 * an all-padding pattern is the EMPTY pattern in every kind, like the reference's *_allow_empty_pattern functions) OR clear (`clear_pattern`: ASCII bytes; pattern_padding ignored); the other pointer is NULL.
 * Encrypted integers (n, start, end) are 4-block radix ciphertexts. */

/* StringServerKey::add / add_encrypted (add.rs:18-43): out_content = s1 then s2 (len1 + len2 chars), *out_padding the
 * padding kind of the result.  No bootstrap. */
int tfhe_hip_string_add(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, int padding1, const uint64_t *s2,
                        size_t len2, int padding2, uint64_t *out_content, int *out_padding);
/* repeat_clear (n_enc == NULL: n_clear copies, n_clear * len chars; repeat.rs:17-28) / repeat_encrypted (n_enc: copy i
 * is kept iff n > i, n_max * len chars; repeat.rs:43-71) */
int tfhe_hip_string_repeat(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, uint64_t n_clear,
                           const uint64_t *n_enc, size_t n_max, uint64_t *out_content, int *out_padding);
/* substring_clear (start_enc == end_enc == NULL; out_content: end - start chars; a range outside the content is the
 * reference's ConversionError::OutOfRange -> non-zero return) / substring_encrypted (out_content: len chars, characters
 * outside [start, end) zeroed) (substring.rs:14-151).  out_in_range: "start <= end <= length of the string". */
int tfhe_hip_string_substring(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, uint64_t start,
                              uint64_t end, const uint64_t *start_enc, const uint64_t *end_enc, uint64_t *out_in_range,
                              uint64_t *out_content, int *out_padding);
/* nth_clear / nth_encrypted (accessors.rs:9-111): the n-th character (4 blocks), zero if there is none */
int tfhe_hip_string_nth(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, uint64_t n_clear,
                        const uint64_t *n_enc, uint64_t *out_char);
/* cmux_empty_string (accessors.rs:113-139): the string if the boolean block `condition` is 1, else the empty string */
int tfhe_hip_string_cmux_empty(tfhe_hip_host_key *hk, const uint64_t *condition, const uint64_t *s, size_t len,
                               int padding, uint64_t *out_content, int *out_padding);

/* The split family -> FheSplit { parts, number_parts } (split.rs:9-13).  out_parts holds *out_n_parts strings of `len`
 * chars each (part q at out_parts + q * len * 4 blocks): the characters of part q at their original positions, zeros
 * elsewhere (Padding::InitialAndFinal, like the reference's substring_encrypted_final_padding parts); parts at and
 * beyond the encrypted out_number_parts (4 blocks) are to be ignored.  parts_cap (in strings) must be at least
 * tfhe_hip_string_split_max_parts().  The r* kinds yield the parts from the end of the string, as str::rsplit does.
 * n (splitn / rsplitn only): n_enc if not NULL, else n_clear.  The empty pattern (pattern_len 0, or hidden under padding) is accepted by every kind, with the semantics of Rust's str
 * methods ("abc".splitn(3, "") = ["", "a", "bc"], split_terminator / split_inclusive = ["", "a", "b", "c"], rsplit_terminator = ["c", "b", "a", ""]). */
enum {
    TFHE_HIP_SPLIT = 0,                  /* split.rs:16-56 */
    TFHE_HIP_RSPLIT = 1,                 /* rsplit.rs */
    TFHE_HIP_SPLIT_TERMINATOR = 2,       /* split_terminator.rs */
    TFHE_HIP_RSPLIT_TERMINATOR = 3,      /* rsplit.rs (rsplit_terminator) */
    TFHE_HIP_SPLIT_INCLUSIVE = 4,        /* split_inclusive.rs */
    TFHE_HIP_SPLITN = 5,                 /* split.rs:58-161 */
    TFHE_HIP_RSPLITN = 6,                /* rsplitn.rs */
    TFHE_HIP_SPLIT_ASCII_WHITESPACE = 7  /* split_ascii_whitespace.rs (no pattern) */
};
int tfhe_hip_string_split_max_parts(int kind, size_t len, size_t pattern_len, int pattern_padding, int n_is_encrypted,
                                    uint64_t n_clear, size_t *out);
int tfhe_hip_string_split(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, const uint64_t *pattern,
                          const uint8_t *clear_pattern, size_t pattern_len, int pattern_padding, int kind, uint64_t n_clear,
                          const uint64_t *n_enc, uint64_t *out_parts, size_t parts_cap, uint64_t *out_number_parts,
                          size_t *out_n_parts);
/* split_once / rsplit_once: Option<(before, after)>: out_found, and the two strings (len chars each) */
int tfhe_hip_string_split_once(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, const uint64_t *pattern,
                               const uint8_t *clear_pattern, size_t pattern_len, int pattern_padding, int reverse,
                               uint64_t *out_found, uint64_t *out_before, uint64_t *out_after);
/* replace (limited == 0) / replacen (the first n occurrences; replace.rs:20-52).  `from` and `to` are patterns as above
 * (`to` may be empty; its padding zeros, if any, are inserted as they are).  The occurrences are overwritten in place:
 * *out_len = len * (1 + max(0, to_len - from_len)) chars; a padded `from` (encrypted length): len * (1 + to_len) chars,
 * `to` behind the last char of each occurrence (tfhe_hip_string_replace_max_len); padding anywhere (*out_padding). */
int tfhe_hip_string_replace_max_len(size_t len, size_t from_len, int from_padding, size_t to_len, size_t *out);
int tfhe_hip_string_replace(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, const uint64_t *from,
                            const uint8_t *clear_from, size_t from_len, int from_padding, const uint64_t *to,
                            const uint8_t *clear_to, size_t to_len, int limited, uint64_t n_clear, const uint64_t *n_enc, uint64_t *out_content,
                            size_t out_cap, size_t *out_len, int *out_padding);

/* trim_start_encrypted / trim_end_encrypted / trim_encrypted (trim.rs:135-200): trim an ENCRYPTED character (4 blocks) */
int tfhe_hip_string_trim_encrypted_char(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *character,
                                        int left, int right, uint64_t *out_content, uint64_t *out_length);
/* push_padding_to_start (change_padding.rs:36-54) */
int tfhe_hip_string_push_padding_to_start(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint64_t *out);
/* pop_first_non_zero_char (last = 0) / pop_last_non_zero_char (last = 1) (change_padding.rs:74-140): out_char = that char
 * (zero if there is none), out_content = the content with it replaced by a padding zero */
int tfhe_hip_string_pop_non_zero_char(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int last, uint64_t *out_char,
                                      uint64_t *out_content);

#ifdef __cplusplus
}
#endif
#endif
