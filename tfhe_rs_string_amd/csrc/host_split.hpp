// host_split.hpp -- the rest of the FheString interface on the op-graph engine of host_ops.hpp:
// concatenation / repeat / substring / nth (add.rs, repeat.rs, substring.rs, accessors.rs), the split family
// (split.rs, rsplit.rs, rsplitn.rs, split_terminator.rs, split_inclusive.rs, split_ascii_whitespace.rs) and
// replace / replacen (replace.rs).  Included at the end of host_ops.hpp.
//
// The reference's split walks the string once per PART: `find` from the current start, `substring`, next start
// -- (L + 1) sequential rounds of O(L) radix comparisons each.  Here a split is a data-parallel scatter:
//   1. occurrence flags of the pattern at every position (one launch + the all-true tree);
//   2. the occurrences the left-to-right (right-to-left for the r* family) search would take: only patterns
//      that can overlap themselves need a scan, and only over the distances at which they can (an encrypted pattern
//      with padding has an encrypted length: the scans of all its possible lengths run side by side and the
//      one-hot length picks); from here on a separator is represented by the flag of its LAST char;
//   3. the exclusive prefix COUNT of separators that end before each position, as base-4 digits (blocked scan:
//      leveled sums inside chunks of 12, one digit-and-carry PBS pair per digit) -- the part a character
//      belongs to;
//      (while the count cannot exceed 15 -- short strings, or the head of a long one -- it is simply the leveled
//      sum of the flags: no bootstrap);
//   4. part q, position j = char j iff count_j == q (and j is outside the separators): "x == v" flags on the
//      two packed nibbles of the count, then ONE bivariate PBS per output block.  A position can only lie in the
//      parts up to its own bound (count_j <= j / |pattern|), everything beyond is a trivial zero.
// The parts keep the reference's shape (content of the full length, characters at their original positions,
// zeros elsewhere: Padding::InitialAndFinal, substring.rs:117-151) and `number_parts` its meaning (split.rs:9-13).
#pragma once

namespace tfhe_hip {
namespace host {

inline void apply_lut(ServerKey &sk, Ciphertext &c, const LookupTable &lut) { sk.apply_lookup_table_batch({&c}, {&lut}); }

inline Ciphertext boolean_not_leveled(ServerKey &sk, const Ciphertext &b)
{
    Ciphertext r = sk.create_trivial(1);
    sk.lwe_sub_assign(r, b);
    r.degree = 1;
    r.noise_level = b.noise_level;
    return r;
}

struct PaddedString {
    FheString content;
    Padding padding = Padding::None;
};

// ---- add (add.rs:24-43): contents are appended, the padding kind follows -----------------------------
inline Padding concat_padding(Padding p1, Padding p2)
{
    if (p1 == Padding::None && p2 == Padding::None) return Padding::None;
    if (p1 == Padding::None && p2 == Padding::Final) return Padding::Final;
    if (p1 == Padding::Initial && p2 == Padding::None) return Padding::Initial;
    if (p1 == Padding::Initial && p2 == Padding::Final) return Padding::InitialAndFinal;
    return Padding::Anywhere;
}
inline PaddedString string_concat(const FheString &s1, Padding p1, const FheString &s2, Padding p2)
{
    PaddedString r;
    if (s2.empty()) return {s1, p1};
    if (s1.empty()) return {s2, p2};
    r.content = s1;
    r.content.insert(r.content.end(), s2.begin(), s2.end());
    r.padding = concat_padding(p1, p2);
    return r;
}
// repeat_clear (repeat.rs:17-28)
inline PaddedString string_repeat_clear(const FheString &s, Padding p, size_t n)
{
    PaddedString r;
    for (size_t i = 0; i < n; i++) r = string_concat(r.content, r.padding, s, p);
    return r;
}

// every block of `ch` kept iff the sum `cond` of boolean flags equals `want` (1..3), else zero
struct KeepLuts {
    LookupTable lut[4];
    bool have[4] = {false, false, false, false};
    const LookupTable &get(ServerKey &sk, uint64_t want)
    {
        if (!have[want]) {
            lut[want] = sk.generate_lookup_table_bivariate([want](uint64_t block, uint64_t c) { return c == want ? block : (uint64_t)0; });
            have[want] = true;
        }
        return lut[want];
    }
};
inline RadixCiphertext keep_char_if(ServerKey &sk, const RadixCiphertext &ch, const Ciphertext &cond, uint64_t want, KeepLuts &luts)
{
    RadixCiphertext out = ch;
    if (cond.is_trivial()) { // known on the host
        if (cond.body != sk.create_trivial(want).body)
            for (auto &b : out) b = sk.create_trivial(0);
        return out;
    }
    for (auto &b : out) {
        if (b.is_trivial() && b.body == 0) continue;
        Ciphertext c = cond;
        c.degree = std::min<uint64_t>(c.degree, 3);
        sk.bivariate_pack_assign(b, c);
        apply_lut(sk, b, luts.get(sk, want));
    }
    return out;
}

// repeat_encrypted (repeat.rs:43-71): copy i is kept iff n > i.  (The reference labels the result Padding::None
// for an unpadded `s` although the dropped copies are zeros; Final is what the content is.)
inline PaddedString string_repeat_encrypted(ServerKey &sk, const FheString &s, Padding p, const RadixCiphertext &n, size_t n_max)
{
    PaddedString r;
    KeepLuts keep;
    for (size_t i = 0; i < n_max; i++) {
        const Ciphertext again = scalar_compare_parallelized(sk, n, i, Cmp::GT);
        for (const auto &c : s) r.content.push_back(keep_char_if(sk, c, again, 1, keep));
    }
    r.padding = p == Padding::None ? Padding::Final : Padding::Anywhere;
    return r;
}

// cmux_empty_string (accessors.rs:113-139)
inline PaddedString string_cmux_empty(ServerKey &sk, const Ciphertext &condition, const FheString &s, Padding p)
{
    PaddedString r;
    KeepLuts keep;
    for (const auto &c : s) r.content.push_back(keep_char_if(sk, c, condition, 1, keep));
    r.padding = p == Padding::None ? Padding::Final : p;
    return r;
}

// ---- substring (substring.rs) -------------------------------------------------------------------------
struct ResultFheString { // split.rs:7: (the range lies inside the string, the slice)
    Ciphertext in_range;
    PaddedString value;
};
// substring_clear (substring.rs:14-57): the slice of the content; OutOfRange (-> std::out_of_range) when the range is
// outside the content.  With final padding "end <= len" is "the character before `end` is not a padding zero".
inline ResultFheString string_substring_clear(ServerKey &sk, const FheString &s_in, Padding p, size_t start, size_t end)
{
    if (end < start || end > s_in.size()) throw std::out_of_range("substring: the range is not inside the content");
    const FheString s = normalize_padding(sk, s_in, p);
    ResultFheString r;
    r.in_range = sk.create_trivial(1);
    if (p == Padding::Final && end > 0) {
        FheString last(1, s[end - 1]);
        r.in_range = boolean_bitnot(sk, char_is_zero_flags(sk, last)[0]);
    }
    r.value.content.assign(s.begin() + start, s.begin() + end);
    r.value.padding = p;
    return r;
}
// substring_encrypted (substring.rs:59-151): character n stays iff start <= n < end; range check start <= end <= len
inline ResultFheString string_substring_encrypted(ServerKey &sk, const FheString &s_in, Padding p, const RadixCiphertext &start,
                                                  const RadixCiphertext &end)
{
    const FheString s = normalize_padding(sk, s_in, p);
    const size_t L = s.size();
    ResultFheString r;
    const Ciphertext end_in = p == Padding::None ? scalar_compare_parallelized(sk, end, L, Cmp::LE)
                                                 : compare_parallelized(sk, end, string_len(sk, s, p), Cmp::LE);
    const Ciphertext ordered = compare_parallelized(sk, start, end, Cmp::LE);
    r.in_range = are_all_comparisons_block_true(sk, {end_in, ordered});
    KeepLuts keep;
    for (size_t n = 0; n < L; n++) {
        Ciphertext c = scalar_compare_parallelized(sk, start, n, Cmp::LE);
        sk.unchecked_add_assign(c, scalar_compare_parallelized(sk, end, n, Cmp::GT));
        r.value.content.push_back(keep_char_if(sk, s[n], c, 2, keep));
    }
    r.value.padding = Padding::InitialAndFinal;
    return r;
}

// sum of (clean) blocks of which at most one is non-zero: no carry forms and the sum cannot overflow; the fan-in is
// bounded by the noise only (one unit per term), an identity LUT per tree level
inline Ciphertext sum_exclusive_blocks(ServerKey &sk, std::vector<Ciphertext> v)
{
    if (v.empty()) return sk.create_trivial(0);
    const uint64_t mm = sk.message_modulus();
    const LookupTable ident = sk.generate_lookup_table([mm](uint64_t x) { return x % mm; });
    while (v.size() > 1) {
        std::vector<Ciphertext> next;
        constexpr size_t FAN_IN = 256;
        for (size_t b = 0; b < v.size(); b += FAN_IN) {
            Ciphertext acc = v[b];
            for (size_t t = b + 1; t < std::min(v.size(), b + FAN_IN); t++) sk.unchecked_add_assign(acc, v[t]);
            acc.degree = mm - 1;
            if (!acc.is_trivial()) apply_lut(sk, acc, ident);
            next.push_back(std::move(acc));
        }
        v = std::move(next);
    }
    return v[0];
}

// ---- nth (accessors.rs:9-111) -------------------------------------------------------------------------
inline RadixCiphertext string_nth_clear(ServerKey &sk, const FheString &s_in, Padding p, size_t n)
{
    if (n >= s_in.size()) return trivial_radix(sk, 0, 4);
    const FheString s = normalize_padding(sk, s_in, p);
    return s[n];
}
// one-hot select: the flags n == i are exclusive, so the selected blocks can be summed without carries
inline RadixCiphertext string_nth_encrypted(ServerKey &sk, const FheString &s_in, Padding p, const RadixCiphertext &n)
{
    const FheString s = normalize_padding(sk, s_in, p);
    RadixCiphertext out = trivial_radix(sk, 0, 4);
    if (s.empty()) return out;
    const LookupTable eq = equality_lut(sk);
    KeepLuts keep;
    std::vector<std::vector<Ciphertext>> sel(4);
    for (size_t i = 0; i < s.size(); i++) {
        if (i >> (2 * n.size())) break; // n cannot reach this position
        std::vector<Ciphertext> cmp;
        const Comparator c(sk);
        c.stage_scalar(n, i, cmp);
        for (auto &x : cmp) apply_lut(sk, x, eq);
        const Ciphertext here = are_all_comparisons_block_true(sk, std::move(cmp));
        const RadixCiphertext ch = keep_char_if(sk, s[i], here, 1, keep);
        for (size_t b = 0; b < 4; b++) sel[b].push_back(ch[b]);
    }
    for (size_t b = 0; b < 4; b++) out[b] = sum_exclusive_blocks(sk, std::move(sel[b]));
    return out;
}

// ---- exclusive prefix counts --------------------------------------------------------------------------
struct PrefixCounts {
    std::vector<RadixCiphertext> k; // k[j], j = 0..n: the number of events before position j (index_blocks() base-4 digits)
    std::vector<uint64_t> bound;    // k[j] <= bound[j] whatever the data
    // while the count cannot exceed 15 it is also available as a plain leveled sum of the events (no bootstrap at all):
    // exactly the packed low nibble the "count == q" flags are looked up on
    std::vector<Ciphertext> direct;
    std::vector<bool> has_direct;
};
// events: boolean blocks; max_before(j): an upper bound of the number of events before j.
// Chunks of 12 positions: the running sum inside a chunk is leveled (<= 12), the digits of the count up to the
// chunk are carried along: digit_0 + run <= 15.  Per position: a (digit, carry) PBS pair for every digit that can
// overflow; digits that cannot are plain sums.
inline PrefixCounts prefix_counts(ServerKey &sk, const std::vector<Ciphertext> &events, const std::function<uint64_t(size_t)> &max_before)
{
    constexpr size_t CH = 12;
    const size_t n = events.size();
    if (n > sk.max_index()) throw std::runtime_error("split: more positions than the count radix can say (" + std::to_string(sk.max_index()) + " with " + std::to_string(sk.index_blocks()) + " blocks: tfhe_hip_host_key_set_num_blocks)");
    const LookupTable lut_msg = sk.generate_lookup_table([](uint64_t x) { return x & 3; });
    const LookupTable lut_carry = sk.generate_lookup_table([](uint64_t x) { return x >> 2; });
    PrefixCounts r;
    r.k.resize(n + 1);
    r.bound.resize(n + 1);
    r.direct.resize(n + 1);
    r.has_direct.assign(n + 1, false);
    Ciphertext all = sk.create_trivial(0); // the sum of all events so far, kept while it is bounded by 15
    bool all_valid = true;
    RadixCiphertext A = trivial_radix(sk, 0, sk.index_blocks());
    uint64_t A_bound = 0;
    auto digits = [&](const Ciphertext &run, uint64_t run_bound, uint64_t bound) {
        RadixCiphertext d(sk.index_blocks());
        Ciphertext carry = run;
        uint64_t carry_bound = std::min(run_bound, bound);
        for (size_t i = 0; i < sk.index_blocks(); i++) {
            const uint64_t a_bound = std::min<uint64_t>(3, A_bound >> (2 * i));
            Ciphertext y = A[i];
            sk.unchecked_add_assign(y, carry);
            const uint64_t y_bound = std::min(a_bound + carry_bound, bound >> (2 * i));
            y.degree = y_bound;
            if (y_bound < 4) {
                d[i] = std::move(y);
                carry = sk.create_trivial(0);
                carry_bound = 0;
            } else {
                Ciphertext c = y;
                apply_lut(sk, y, lut_msg);
                apply_lut(sk, c, lut_carry);
                d[i] = std::move(y);
                carry = std::move(c);
                carry_bound = y_bound >> 2;
            }
        }
        return d;
    };
    for (size_t c0 = 0;; c0 += CH) {
        Ciphertext run = sk.create_trivial(0);
        uint64_t run_bound = 0;
        const size_t c1 = std::min(n, c0 + CH);
        for (size_t w = c0; w < c1; w++) {
            r.bound[w] = std::min<uint64_t>(max_before(w), w);
            r.k[w] = digits(run, run_bound, r.bound[w]);
            if (all_valid && r.bound[w] <= 15) {
                r.direct[w] = all;
                r.direct[w].degree = r.bound[w];
                r.has_direct[w] = true;
            } else {
                all_valid = false;
            }
            sk.unchecked_add_assign(run, events[w]);
            if (all_valid) sk.unchecked_add_assign(all, events[w]);
            run_bound++;
        }
        const uint64_t b1 = std::min<uint64_t>(max_before(c1), c1);
        RadixCiphertext next = digits(run, run_bound, b1);
        if (c1 == n) {
            r.bound[n] = b1;
            r.k[n] = std::move(next);
            if (all_valid && b1 <= 15) {
                r.direct[n] = all;
                r.direct[n].degree = b1;
                r.has_direct[n] = true;
            }
            break;
        }
        A = std::move(next);
        A_bound = b1;
    }
    return r;
}

// ---- scatter: parts[q][j] = s[j] iff index[j] == q + offset and every gate flag of j is set --------------------
// rest: one more part (the last) takes the chars with index >= n_parts, whatever the gate says (the unsplit
// remainder of splitn; offset 0, n_parts <= 16)
inline std::vector<FheString> scatter_by_index(ServerKey &sk, const FheString &s, const std::vector<const RadixCiphertext *> &index,
                                               const std::vector<uint64_t> &bound, const std::vector<Ciphertext> *gate, size_t n_parts,
                                               uint64_t offset = 0, bool rest = false, const std::vector<const Ciphertext *> *direct = nullptr)
{
    const size_t L = s.size();
    if (rest && (offset != 0 || n_parts > 16)) throw std::runtime_error("scatter: the rest part needs offset 0 and at most 16 parts before it");
    ScalarEqLuts eq;
    KeepLuts keep;
    std::vector<FheString> parts(n_parts + (rest ? 1 : 0), FheString(L, trivial_radix(sk, 0, 4)));
    if (n_parts + offset == 0 && !rest) return parts;
    const LookupTable unless2 = sk.generate_lookup_table_bivariate([](uint64_t block, uint64_t c) { return c != 2 ? block : (uint64_t)0; });
    for (size_t j = 0; j < L; j++) {
        const uint64_t b = bound[j];
        const uint64_t vmax = std::min<uint64_t>(b, n_parts + offset - (n_parts + offset > 0 ? 1 : 0)); // largest value a part asks for
        const RadixCiphertext &k = *index[j];
        std::vector<Ciphertext> eqlo, eqhi;
        if (b > 0) {
            // the low nibble of the index: the plain sum of the events where there is one (b <= 15), else the packed digits
            const Ciphertext lo = (direct && (*direct)[j]) ? *(*direct)[j] : pack_block_pair(sk, k, 0);
            for (uint64_t v = 0; v <= std::min<uint64_t>(15, vmax); v++) {
                eqlo.push_back(lo);
                apply_lut(sk, eqlo.back(), eq.get(sk, (unsigned)v));
            }
        }
        if (b > 15) {
            const Ciphertext hi = pack_block_pair(sk, k, 2);
            for (uint64_t v = 0; v <= (vmax >> 4); v++) {
                eqhi.push_back(hi);
                apply_lut(sk, eqhi.back(), eq.get(sk, (unsigned)v));
            }
        }
        for (size_t q = 0; q < n_parts; q++) {
            const uint64_t v = q + offset;
            if (v > b) break; // the count cannot reach this part: trivial zeros
            Ciphertext cond = sk.create_trivial(0);
            uint64_t want = 0;
            if (b > 0) { sk.unchecked_add_assign(cond, eqlo[v & 15]); want++; }
            if (b > 15) { sk.unchecked_add_assign(cond, eqhi[v >> 4]); want++; }
            if (gate) { sk.unchecked_add_assign(cond, (*gate)[j]); want++; }
            if (want == 0) { parts[q][j] = s[j]; continue; }
            cond.degree = want;
            parts[q][j] = keep_char_if(sk, s[j], cond, want, keep);
        }
        if (rest && b >= n_parts) { // index >= n_parts  <=>  not (high nibble == 0 and low nibble one of 0 .. n_parts - 1)
            Ciphertext early = sk.create_trivial(0);
            for (size_t q = 0; q < n_parts; q++) sk.unchecked_add_assign(early, eqlo[q]);
            early.degree = 1;
            if (b > 15) {
                sk.unchecked_add_assign(early, eqhi[0]);
                early.degree = 2;
                RadixCiphertext out = s[j];
                for (auto &blk : out) {
                    sk.bivariate_pack_assign(blk, early);
                    apply_lut(sk, blk, unless2);
                }
                parts[n_parts][j] = std::move(out);
            } else {
                parts[n_parts][j] = keep_char_if(sk, s[j], early, 0, keep);
            }
        }
    }
    return parts;
}

// ---- which occurrences the search takes ---------------------------------------------------------------
// The pattern occurs at i (m[i]); the left-to-right search takes occurrence i iff it took none at i - d for the
// distances d < |pattern| at which two occurrences can overlap.  At most one of those was taken, so
// 2 m[i] + (their sum) == 2 decides it: a scan of one PBS per position, needed only where overlaps exist
// (never for single chars; "--", "aa", "abab" do need it).
// The scan is sequential, so long strings are cut into chunks of C ~ sqrt(n) positions that are scanned
// SPECULATIVELY and at the same time: the only thing a chunk needs from its past is how many of its first
// positions are still covered by the last occurrence taken before it (s < |pattern|), and the scan after s
// blocked positions is the scan that starts fresh at offset s.  So every chunk runs |pattern| fresh scans
// (C levels, all chunks together), the one-hot state is then passed from chunk to chunk (one level per chunk:
// state_{k+1}[s'] = OR_s state_k[s] AND "scan s of chunk k took the occurrence that leaves s' positions
// blocked") and selects each chunk's result (one level): ~2 sqrt(n) levels instead of n.
inline Ciphertext boolean_and(ServerKey &sk, const Ciphertext &a, const Ciphertext &b, const LookupTable &both)
{
    if (a.is_trivial()) return a.body ? b : a;
    if (b.is_trivial()) return b.body ? a : b;
    Ciphertext x = a;
    sk.unchecked_add_assign(x, b);
    x.degree = 2;
    apply_lut(sk, x, both);
    return x;
}
inline std::vector<Ciphertext> take_non_overlapping(ServerKey &sk, const std::vector<Ciphertext> &m, const std::vector<size_t> &distances,
                                                    size_t P)
{
    if (distances.empty()) return m;
    const size_t n = m.size();
    const LookupTable take = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    // the sequential rule on [a, e) with nothing taken before a; result indexed by position - a
    auto chain = [&](size_t a, size_t e) {
        std::vector<Ciphertext> t(e - a);
        for (size_t i = a; i < e; i++) {
            Ciphertext x = m[i];
            sk.unchecked_scalar_mul_assign(x, 2);
            bool any = false;
            for (size_t d : distances)
                if (d <= i - a) { sk.unchecked_add_assign(x, t[i - a - d]); any = true; }
            if (!any) { t[i - a] = m[i]; continue; }
            x.degree = 3;
            apply_lut(sk, x, take);
            t[i - a] = std::move(x);
        }
        return t;
    };
    size_t C = P - 1;
    while (C * C < n) C++;
    if (n <= 12 || C >= n) return chain(0, n);
    const size_t chunks = (n + C - 1) / C;
    std::vector<Ciphertext> t(n);
    std::vector<Ciphertext> state(P, sk.create_trivial(0)); // one-hot: s positions at the start of the chunk are blocked
    state[0] = sk.create_trivial(1);
    for (size_t k = 0; k < chunks; k++) {
        const size_t b = k * C, e = std::min(n, b + C);
        std::vector<std::vector<Ciphertext>> scan(P);
        for (size_t s = 0; s < P && b + s < e; s++)
            if (!(state[s].is_trivial() && state[s].body == 0)) scan[s] = chain(b + s, e);
        for (size_t i = b; i < e; i++) { // the chunk's result: the scan the incoming state selects
            Ciphertext acc = sk.create_trivial(0);
            for (size_t s = 0; s < P && b + s <= i; s++)
                if (!scan[s].empty()) sk.unchecked_add_assign(acc, boolean_and(sk, state[s], scan[s][i - b - s], take));
            acc.degree = 1;
            t[i] = std::move(acc);
        }
        if (k + 1 == chunks) break;
        std::vector<Ciphertext> next(P, sk.create_trivial(0)); // an occurrence taken at e - P + s' blocks s' positions of the next chunk
        Ciphertext some = sk.create_trivial(0);
        for (size_t sp = 1; sp < P; sp++) {
            if (e + sp < P + b) continue; // that position lies before this chunk (C >= P - 1: cannot happen for full chunks)
            next[sp] = t[e - P + sp];
            sk.unchecked_add_assign(some, next[sp]);
        }
        next[0] = boolean_not_leveled(sk, some);
        state = std::move(next);
    }
    return t;
}

// a pattern of the split family: clear bytes, or an encrypted string (any padding kind: `enc_padding`)
struct SplitPattern {
    const std::string *clear = nullptr;
    const FheString *enc = nullptr;
    Padding enc_padding = Padding::None;
    size_t size() const { return clear ? clear->size() : enc->size(); }
};
inline std::vector<size_t> overlap_distances(const SplitPattern &pat)
{
    std::vector<size_t> d;
    const size_t P = pat.size();
    for (size_t k = 1; k < P; k++)
        if (!pat.clear || pat.clear->compare(k, P - k, *pat.clear, 0, P - k) == 0) d.push_back(k);
    return d;
}
// What is known about the length of the pattern.  An encrypted pattern with (final) padding has an ENCRYPTED length
// 1 .. P: real[d] = "char d is a real char" (non-increasing), is_len[l] = "the length is l + 1" = real[l] - real[l + 1].
// (Length 0 -- the empty pattern, known (P == 0) or hidden under padding -- has no occurrence flags of its own: string_split /
// string_replace add its events, "in front of the string and behind every real char", switched on by "the pattern is empty".)
struct PatternShape {
    size_t P = 0;
    bool padded = false;
    std::vector<Ciphertext> real, is_len;
    size_t min_len() const { return padded ? 1 : P; }
};
inline PatternShape pattern_shape(ServerKey &sk, const SplitPattern &pat)
{
    PatternShape sh;
    sh.P = pat.size();
    sh.padded = pat.enc && pat.enc_padding != Padding::None;
    if (!sh.padded) return sh;
    const std::vector<Ciphertext> z = char_is_zero_flags(sk, *pat.enc);
    for (size_t d = 0; d < sh.P; d++) sh.real.push_back(boolean_not_leveled(sk, z[d]));
    for (size_t l = 0; l < sh.P; l++) {
        Ciphertext f = sh.real[l];
        if (l + 1 < sh.P) sk.lwe_sub_assign(f, sh.real[l + 1]);
        f.degree = 1;
        sh.is_len.push_back(std::move(f));
    }
    return sh;
}
// START flags of the separators the forward search takes, size L (positions where the pattern cannot start: trivial 0)
inline std::vector<Ciphertext> taken_separators(ServerKey &sk, const FheString &s, const SplitPattern &pat, const PatternShape &sh)
{
    const size_t L = s.size(), P = pat.size();
    std::vector<Ciphertext> t(L, sk.create_trivial(0));
    if (P == 0 || L == 0) return t;
    if (sh.padded) {
        // occurrence flags with the padding chars as wildcards, the selection scan for every possible length at
        // once (independent chains: same depth as one), then the scan of the actual length: sum_l is_len[l] AND scan_l
        const std::vector<Ciphertext> m = match_padded_at(sk, s, *pat.enc, all_positions(L));
        const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
        std::vector<size_t> dist;
        for (size_t l = 0; l < P; l++) { // length l + 1: two occurrences may overlap at the distances 1 .. l
            if (l > 0) dist.push_back(l);
            const std::vector<Ciphertext> tl = take_non_overlapping(sk, m, dist, l + 1);
            for (size_t i = 0; i < L; i++) sk.unchecked_add_assign(t[i], boolean_and(sk, sh.is_len[l], tl[i], both));
        }
        for (auto &c : t) c.degree = 1;
        return t;
    }
    if (P > L) return t;
    std::vector<size_t> starts(L - P + 1);
    for (size_t w = 0; w < starts.size(); w++) starts[w] = w;
    std::vector<Ciphertext> m = pat.clear ? match_clear_at(sk, s, *pat.clear, starts) : match_encrypted_at(sk, s, *pat.enc, starts);
    m = take_non_overlapping(sk, m, overlap_distances(pat), P);
    for (size_t w = 0; w < m.size(); w++) t[w] = std::move(m[w]);
    return t;
}
// END flags: a taken separator ends at e (its last char)
inline std::vector<Ciphertext> ends_of(ServerKey &sk, const std::vector<Ciphertext> &starts, const PatternShape &sh)
{
    const size_t L = starts.size(), P = sh.P;
    std::vector<Ciphertext> e(L, sk.create_trivial(0));
    if (P == 0) return e; // the (known) empty pattern has no separators of its own: its events are added by the caller
    if (!sh.padded) {
        for (size_t j = P - 1; j < L; j++) e[j] = starts[j + 1 - P];
        return e;
    }
    const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    for (size_t j = 0; j < L; j++) {
        for (size_t l = 0; l < P && l <= j; l++) sk.unchecked_add_assign(e[j], boolean_and(sk, starts[j - l], sh.is_len[l], both));
        e[j].degree = 1;
    }
    return e;
}
// 1 - (j lies inside a separator), from the END flags: the separator ending at j + d covers j iff it is longer than d
inline std::vector<Ciphertext> outside_separators(ServerKey &sk, const std::vector<Ciphertext> &ends, const PatternShape &sh)
{
    const size_t L = ends.size(), P = sh.P;
    const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    std::vector<Ciphertext> g(L);
    for (size_t j = 0; j < L; j++) {
        Ciphertext in = sk.create_trivial(0);
        for (size_t d = 0; d < P && j + d < L; d++)
            sk.unchecked_add_assign(in, sh.padded && d > 0 ? boolean_and(sk, ends[j + d], sh.real[d], both) : ends[j + d]);
        in.degree = 1;
        g[j] = boolean_not_leveled(sk, in);
    }
    return g;
}

struct FheSplit { // split.rs:9-13
    std::vector<FheString> parts;
    RadixCiphertext number_parts;
};
struct SplitOptions {
    bool reverse = false;    // rsplit*: search from the end, parts in reverse order
    bool inclusive = false;  // split_inclusive: the separator stays at the end of its part
    bool terminator = false; // split_terminator / rsplit_terminator: an empty last part is dropped
    bool limited = false;    // splitn / rsplitn
    uint64_t n_clear = 0;
    const RadixCiphertext *n_enc = nullptr; // index_blocks() blocks
};

inline FheString reversed_string(const FheString &s) { return FheString(s.rbegin(), s.rend()); }

// upper bound of the number of parts `string_split` returns for a content of L chars; min_len: the shortest the
// pattern can be (its content length, or 1 for an encrypted pattern with padding)
inline size_t split_max_parts(size_t L, size_t min_len, const SplitOptions &o)
{
    if (min_len == 0) return L + 2;
    size_t m = L / min_len + 1;
    if (o.limited && !o.n_enc) m = std::min<size_t>(m, o.n_clear);
    return m;
}

// only the events whose rank (the count before them) is below `limit` (clear) / `limit_enc` stay (splitn, replacen)
inline std::vector<Ciphertext> limit_events(ServerKey &sk, const std::vector<Ciphertext> &t, const PrefixCounts &pc, uint64_t limit,
                                            const RadixCiphertext *limit_enc)
{
    const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 4 + IS_INFERIOR); });
    const Comparator cmp(sk);
    std::vector<Ciphertext> out(t.size());
    for (size_t i = 0; i < t.size(); i++) {
        if (t[i].is_trivial() || (!limit_enc && pc.bound[i] < limit)) { out[i] = t[i]; continue; }
        const Ciphertext sign = limit_enc ? cmp.compare(pc.k[i], *limit_enc) : cmp.scalar_compare(pc.k[i], limit);
        Ciphertext x = t[i]; // an event and its rank < limit
        x.degree = 1;
        sk.bivariate_pack_assign(x, sign);
        apply_lut(sk, x, both);
        out[i] = std::move(x);
    }
    return out;
}

// the r* family searches the reversed string for the reversed pattern; a padded pattern is compacted again
inline void reverse_pattern(ServerKey &sk, SplitPattern &pat, std::string &clear_rev, FheString &enc_rev)
{
    if (pat.clear) {
        clear_rev.assign(pat.clear->rbegin(), pat.clear->rend());
        pat.clear = &clear_rev;
        return;
    }
    enc_rev = reversed_string(*pat.enc);
    if (pat.enc_padding != Padding::None) enc_rev = push_padding_to_end(sk, enc_rev);
    pat.enc = &enc_rev;
}

// s: no padding or final padding (normalize_padding first); the pattern: clear, or encrypted with no or final padding
inline FheSplit string_split(ServerKey &sk, const FheString &s_in, Padding pad, const SplitPattern &pat_in, const SplitOptions &o)
{
    if (pad != Padding::None && pad != Padding::Final) throw std::runtime_error("split: normalize the padding first");
    if (pat_in.enc && pat_in.enc_padding != Padding::None && pat_in.enc_padding != Padding::Final)
        throw std::runtime_error("split: normalize the padding of the pattern first");
    const size_t L = s_in.size(), P = pat_in.size();
    if (L + 2 > sk.max_index()) throw std::runtime_error("split: more parts than the count radix can say (" + std::to_string(sk.max_index()) + " with " + std::to_string(sk.index_blocks()) + " blocks: tfhe_hip_host_key_set_num_blocks)");
    const bool known_empty = P == 0; // the clear empty pattern, or an encrypted one of content length 0
    FheSplit r;
    if (o.limited && !o.n_enc && o.n_clear == 0) { // splitn(0): nothing
        r.number_parts = trivial_radix(sk, 0, sk.index_blocks());
        return r;
    }
    if (o.limited && !o.n_enc && o.n_clear == 1) { // splitn(1): the string itself
        r.parts.push_back(s_in);
        r.number_parts = trivial_radix(sk, 1, sk.index_blocks());
        return r;
    }
    const FheString s = o.reverse ? reversed_string(s_in) : s_in;
    std::string clear_rev;
    FheString enc_rev;
    SplitPattern pat = pat_in;
    if (o.reverse && !known_empty) reverse_pattern(sk, pat, clear_rev, enc_rev);
    const PatternShape sh = pattern_shape(sk, pat);
    // An EMPTY pattern -- known, or an encrypted one that turns out to be padding only -- splits between all chars and at
    // both ends (split.rs:283-312: "", every char, ""; the *_allow_empty_pattern functions of split.rs:491-720,
    // rsplit.rs, rsplitn.rs:112-290, split_terminator.rs:266-330, split_inclusive.rs:82-140).  In terms of events: one in
    // FRONT of the string (rank 0) and one BEHIND every real char j (rank j + 1), all switched on by "the pattern is
    // empty"; then no occurrence was found, so the two kinds of event exclude each other and every kind of split runs
    // on their union.
    const bool empty_aware = sh.padded || known_empty;
    const Ciphertext is_empty = known_empty ? sk.create_trivial(1) : (sh.padded ? boolean_not_leveled(sk, sh.real[0]) : sk.create_trivial(0));
    const size_t min_len = known_empty ? 1 : sh.min_len();
    const std::vector<Ciphertext> t = taken_separators(sk, s, pat, sh);
    std::vector<Ciphertext> ends = ends_of(sk, t, sh);
    // ends before position j: disjoint separators of at least min_len chars inside [0, j)
    auto max_before = [min_len](size_t j) { return (uint64_t)(j / min_len); };
    const LookupTable both2 = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    size_t rest_after = 0; // splitn with a small clear n: parts 0 .. n - 2 as usual, part n - 1 = everything behind them
    // pseudo events of the empty pattern that survive splitn's limit (rank < n - 1): keep_front, keep_char[j]
    Ciphertext keep_front = sk.create_trivial(1);
    std::vector<Ciphertext> keep_char(empty_aware ? L : 0, sk.create_trivial(1));
    if (o.limited) {
        const uint64_t most = empty_aware ? L + 1 : L / min_len; // the most separators there can be
        if (o.n_enc) {
            // rank < n - 1; n == 0 wraps to the maximum: every separator stays, number_parts = min(.., 0) = 0
            const RadixCiphertext nm1 = scalar_add_parallelized(sk, *o.n_enc, sk.max_index());
            if (!known_empty) ends = limit_events(sk, ends, prefix_counts(sk, ends, max_before), 0, &nm1);
            if (empty_aware) { // the ranks of the pseudo events are known: rank < n - 1  <=>  n - 1 > rank
                const Comparator cmp(sk);
                std::vector<Ciphertext> signs;
                for (size_t rank = 0; rank <= L; rank++) signs.push_back(cmp.scalar_compare(nm1, rank));
                std::vector<std::function<bool(uint64_t)>> gt{[](uint64_t x) { return x == IS_SUPERIOR; }};
                signs = cmp.map_signs(std::move(signs), gt);
                keep_front = signs[0];
                for (size_t j = 0; j < L; j++) keep_char[j] = signs[j + 1];
            }
        } else if (o.n_clear - 1 < most) {
            if (o.n_clear - 1 <= 16) rest_after = (size_t)o.n_clear - 1; // one count pass: the last part is "count >= n - 1"
            else {
                if (!known_empty) ends = limit_events(sk, ends, prefix_counts(sk, ends, max_before), o.n_clear - 1, nullptr);
                for (size_t j = 0; j < keep_char.size(); j++)
                    if (j + 1 >= o.n_clear - 1) keep_char[j] = sk.create_trivial(0);
            }
        }
    }
    // zero flags of the chars (final padding only, where the position of the last char is not known)
    std::vector<Ciphertext> z;
    if ((o.inclusive || o.terminator || empty_aware) && pad == Padding::Final && L > 0) z = char_is_zero_flags(sk, s);
    const LookupTable all3 = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 3); });
    // events that are counted: the separator ends -- minus, for rsplit_terminator, the separator at the very end of
    // the string (the start of the reversed one): the empty part behind it is dropped, every other part moves up
    std::vector<Ciphertext> counted = ends;
    if (o.reverse && o.terminator && L > 0 && !known_empty) {
        std::vector<Ciphertext> starts = t;
        if (pad == Padding::None) {
            starts[0] = sk.create_trivial(0);
        } else {
            for (size_t i = 0; i < L; i++) {
                if (t[i].is_trivial()) continue;
                Ciphertext x = t[i]; // taken, the char is real and the one before it (reversed order) is padding
                sk.unchecked_add_assign(x, boolean_not_leveled(sk, z[i]));
                sk.unchecked_add_assign(x, i > 0 ? z[i - 1] : sk.create_trivial(1));
                x.degree = 3;
                apply_lut(sk, x, all3);
                sk.lwe_sub_assign(starts[i], x);
            }
        }
        counted = ends_of(sk, starts, sh);
    }
    // "the string does not end with a separator and is not empty": the last part exists (terminator / inclusive)
    Ciphertext last_exists = sk.create_trivial(1);
    if (o.terminator || o.inclusive) {
        if (L == 0) {
            last_exists = sk.create_trivial(0);
        } else if (pad == Padding::None) {
            if (!o.reverse) last_exists = boolean_not_leveled(sk, ends[L - 1]);
        } else if (!o.reverse) {
            std::vector<Ciphertext> fin; // the string is empty, or: a separator ends at j, char j is real, char j + 1 is padding
            fin.push_back(z[0]);
            for (size_t j = 0; j < L; j++) {
                if (ends[j].is_trivial()) continue;
                Ciphertext x = ends[j];
                sk.unchecked_add_assign(x, boolean_not_leveled(sk, z[j]));
                sk.unchecked_add_assign(x, j + 1 < L ? z[j + 1] : sk.create_trivial(1));
                x.degree = 3;
                apply_lut(sk, x, all3);
                fin.push_back(std::move(x));
            }
            last_exists = boolean_not_leveled(sk, is_at_least_one_comparisons_block_true(sk, std::move(fin)));
        } else {
            last_exists = boolean_not_leveled(sk, z[L - 1]); // reversed: the string is not empty
        }
        if (empty_aware) {
            // the empty pattern: forward, the string always "ends with a separator" -- the trailing "" is dropped
            // (split_terminator / split_inclusive: "", every char); from the end, rsplit_terminator drops the part that
            // comes FIRST (the front event below) and the "" in front of the string stays: the last part always exists
            if (!o.reverse) last_exists = boolean_and(sk, last_exists, boolean_not_leveled(sk, is_empty), both2);
            else if (known_empty) last_exists = sk.create_trivial(1);
            else { // last_exists OR is_empty
                Ciphertext x = last_exists;
                sk.unchecked_add_assign(x, is_empty);
                x.degree = 2;
                apply_lut(sk, x, sk.generate_lookup_table([](uint64_t v) { return (uint64_t)(v != 0); }));
                last_exists = std::move(x);
            }
        }
    }
    size_t shift = 0; // pseudo events in front of position 0
    if (empty_aware) {
        for (size_t j = 0; j < L; j++) {
            Ciphertext here = pad == Padding::Final ? boolean_and(sk, is_empty, boolean_not_leveled(sk, z[j]), both2) : is_empty;
            here = boolean_and(sk, here, keep_char[j], both2);
            sk.unchecked_add_assign(counted[j], here);
            counted[j].degree = 1;
        }
        // the event in front of the string; rsplit_terminator: that is the separator at the very end of the string, whose
        // (empty) part is the one that is dropped
        Ciphertext front = (o.reverse && o.terminator) ? sk.create_trivial(0) : boolean_and(sk, is_empty, keep_front, both2);
        counted.insert(counted.begin(), std::move(front));
        shift = 1;
    }
    // counts; one more pseudo event carries the "+ 1 part" (or "+ the last part, if it exists")
    counted.push_back(last_exists);
    const PrefixCounts pc = prefix_counts(sk, counted, [&](size_t j) {
        const size_t jj = j > shift ? j - shift : 0; // position in the string
        return (jj > L ? max_before(L) + 1 : max_before(jj)) + shift;
    });
    r.number_parts = pc.k[L + 1 + shift];
    if (o.limited && o.n_enc) r.number_parts = min_or_max_parallelized(sk, r.number_parts, *o.n_enc, false);
    if (rest_after && o.n_clear <= 16) {
        // min(count, n) for a small clear n without a comparison: "count == v" flags for v < n (one-hot), the rest is n
        const size_t at = L + 1 + shift;
        const uint64_t n = o.n_clear, b = pc.bound[at];
        ScalarEqLuts eq;
        const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
        const Ciphertext lo = pc.has_direct[at] ? pc.direct[at] : pack_block_pair(sk, pc.k[at], 0);
        Ciphertext hi0 = sk.create_trivial(1);
        if (b > 15) {
            hi0 = pack_block_pair(sk, pc.k[at], 2);
            apply_lut(sk, hi0, eq.get(sk, 0));
        }
        std::vector<Ciphertext> flags;
        std::vector<uint64_t> values;
        Ciphertext rest = sk.create_trivial(1);
        for (uint64_t v = 0; v < n && v <= b; v++) {
            Ciphertext f = lo;
            apply_lut(sk, f, eq.get(sk, (unsigned)v));
            if (b > 15) f = boolean_and(sk, f, hi0, both);
            sk.lwe_sub_assign(rest, f);
            flags.push_back(std::move(f));
            values.push_back(v);
        }
        rest.degree = 1;
        rest.noise_level = flags.size();
        flags.push_back(std::move(rest));
        values.push_back(n);
        std::vector<const Ciphertext *> ptrs;
        for (const auto &f : flags) ptrs.push_back(&f);
        r.number_parts = one_hot_index(sk, ptrs, values, sk.index_blocks());
    } else if (rest_after) {
        r.number_parts = min_or_max_parallelized(sk, r.number_parts, trivial_radix(sk, o.n_clear, sk.index_blocks()), false);
    }
    // the part of a char = the number of separators that END before it: outside the separators that is the number
    // before it, inside one it is the separator's own rank (what split_inclusive and the remainder of splitn ask for)
    std::vector<const RadixCiphertext *> index(L);
    std::vector<uint64_t> bound(L);
    std::vector<const Ciphertext *> direct(L, nullptr);
    for (size_t j = 0; j < L; j++) {
        index[j] = &pc.k[j + shift];
        bound[j] = pc.bound[j + shift];
        if (pc.has_direct[j + shift]) direct[j] = &pc.direct[j + shift];
    }
    std::vector<Ciphertext> gate;
    if (!o.inclusive) gate = outside_separators(sk, ends, sh);
    const size_t n_parts = split_max_parts(L, min_len, o) + shift;
    if (rest_after) r.parts = scatter_by_index(sk, s, index, bound, &gate, rest_after, 0, true, &direct);
    else r.parts = scatter_by_index(sk, s, index, bound, o.inclusive ? nullptr : &gate, n_parts, 0, false, &direct);
    if (o.reverse)
        for (auto &p : r.parts) std::reverse(p.begin(), p.end());
    return r;
}

// split / rsplit by the empty pattern (split.rs:283-312): "", every char, ""
inline FheSplit string_split_empty_pattern(ServerKey &sk, const FheString &s_in, Padding pad, bool reverse)
{
    if (pad != Padding::None && pad != Padding::Final) throw std::runtime_error("split: normalize the padding first");
    const size_t L = s_in.size();
    if (L + 2 > sk.max_index()) throw std::runtime_error("split: more parts than the count radix can say (" + std::to_string(sk.max_index()) + " with " + std::to_string(sk.index_blocks()) + " blocks: tfhe_hip_host_key_set_num_blocks)");
    FheSplit r;
    r.parts.assign(L + 2, FheString(L, trivial_radix(sk, 0, 4)));
    if (pad == Padding::None) {
        for (size_t j = 0; j < L; j++) r.parts[reverse ? L - j : j + 1][j] = s_in[j];
        r.number_parts = trivial_radix(sk, L + 2, sk.index_blocks());
        return r;
    }
    r.number_parts = scalar_add_parallelized(sk, string_len(sk, s_in, pad), 2);
    if (!reverse) {
        for (size_t j = 0; j < L; j++) r.parts[j + 1][j] = s_in[j];
        return r;
    }
    // reversed with final padding: char j is part (number of real chars after j) + 1
    const FheString s = reversed_string(s_in);
    std::vector<Ciphertext> real = char_is_zero_flags(sk, s);
    for (auto &f : real) f = boolean_not_leveled(sk, f);
    const PrefixCounts pc = prefix_counts(sk, real, [](size_t j) { return (uint64_t)j; });
    std::vector<const RadixCiphertext *> index(L);
    std::vector<uint64_t> bound(L);
    for (size_t j = 0; j < L; j++) { index[j] = &pc.k[j]; bound[j] = pc.bound[j]; }
    std::vector<FheString> parts = scatter_by_index(sk, s, index, bound, nullptr, L, 0);
    for (size_t q = 0; q < L; q++) {
        std::reverse(parts[q].begin(), parts[q].end());
        r.parts[q + 1] = std::move(parts[q]);
    }
    return r;
}

// split_ascii_whitespace (split_ascii_whitespace.rs): the maximal runs of non-whitespace characters
inline FheSplit string_split_ascii_whitespace(ServerKey &sk, const FheString &s, Padding pad)
{
    if (pad != Padding::None && pad != Padding::Final) throw std::runtime_error("split: normalize the padding first");
    const size_t L = s.size();
    if (L > sk.max_index()) throw std::runtime_error("split: more positions than the count radix can say (" + std::to_string(sk.max_index()) + " with " + std::to_string(sk.index_blocks()) + " blocks: tfhe_hip_host_key_set_num_blocks)");
    FheSplit r;
    if (L == 0) {
        r.number_parts = trivial_radix(sk, 0, sk.index_blocks());
        return r;
    }
    std::vector<Ciphertext> word = ascii_whitespace_flags(sk, s); // -> 1 - whitespace - padding zero
    std::vector<Ciphertext> z;
    if (pad == Padding::Final) z = char_is_zero_flags(sk, s);
    for (size_t j = 0; j < L; j++) {
        word[j] = boolean_not_leveled(sk, word[j]);
        if (pad == Padding::Final) { sk.lwe_sub_assign(word[j], z[j]); word[j].noise_level += z[j].noise_level; }
    }
    const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    std::vector<Ciphertext> starts(L);
    for (size_t j = 0; j < L; j++) {
        starts[j] = word[j];
        if (j == 0) continue;
        sk.unchecked_add_assign(starts[j], boolean_not_leveled(sk, word[j - 1]));
        starts[j].degree = 2;
        apply_lut(sk, starts[j], both);
    }
    auto max_before = [](size_t j) { return (uint64_t)((j + 1) / 2); };
    const PrefixCounts pc = prefix_counts(sk, starts, max_before);
    r.number_parts = pc.k[L];
    std::vector<const RadixCiphertext *> index(L);
    std::vector<uint64_t> bound(L);
    std::vector<const Ciphertext *> direct(L, nullptr);
    for (size_t j = 0; j < L; j++) {
        index[j] = &pc.k[j + 1];
        bound[j] = pc.bound[j + 1];
        if (pc.has_direct[j + 1]) direct[j] = &pc.direct[j + 1];
    }
    r.parts = scatter_by_index(sk, s, index, bound, &word, (L + 1) / 2, 1, false, &direct);
    return r;
}

// split_once / rsplit_once: Option<(before, after)>.  Only the FIRST occurrence matters, so neither the selection scan
// nor the counts are needed: with seen[j] = "an occurrence starts at or before j" (a 15-ary blocked prefix OR),
// `before` keeps the chars with seen[j] == 0, `after` those with seen[j - |pattern|] == 1, found = seen[last].
// (A pattern with padding has an encrypted length: it goes through splitn(2).)
struct FheSplitOnce {
    Ciphertext found;
    FheString before, after;
};
inline FheSplitOnce string_split_once(ServerKey &sk, const FheString &s_in, Padding pad, const SplitPattern &pat_in, bool reverse)
{
    if (pad != Padding::None && pad != Padding::Final) throw std::runtime_error("split: normalize the padding first");
    FheSplitOnce r;
    const size_t L = s_in.size(), P = pat_in.size();
    if (P == 0) { // "abc".split_once("") == Some(("", "abc")), rsplit_once("") == Some(("abc", ""))
        r.found = sk.create_trivial(1);
        const FheString empty(L, trivial_radix(sk, 0, 4));
        r.before = reverse ? s_in : empty;
        r.after = reverse ? empty : s_in;
        return r;
    }
    const bool padded = pat_in.enc && pat_in.enc_padding != Padding::None;
    if (padded) {
        SplitOptions o;
        o.reverse = reverse;
        o.limited = true;
        o.n_clear = 2;
        FheSplit sp = string_split(sk, s_in, pad, pat_in, o);
        // number_parts is 1 or 2: its low block minus 1 is the flag
        r.found = sp.number_parts[0];
        sk.lwe_plaintext_sub_assign(r.found, 1);
        r.found.degree = 1;
        const FheString empty(L, trivial_radix(sk, 0, 4));
        FheString first = sp.parts.size() > 0 ? sp.parts[0] : empty, second = sp.parts.size() > 1 ? sp.parts[1] : empty;
        r.before = reverse ? second : first;
        r.after = reverse ? first : second;
        return r;
    }
    const FheString empty(L, trivial_radix(sk, 0, 4));
    r.before = r.after = empty;
    r.found = sk.create_trivial(0);
    if (P > L) return r;
    const FheString s = reverse ? reversed_string(s_in) : s_in;
    std::string clear_rev;
    FheString enc_rev;
    SplitPattern pat = pat_in;
    if (reverse) reverse_pattern(sk, pat, clear_rev, enc_rev);
    std::vector<size_t> starts(L - P + 1);
    for (size_t w = 0; w < starts.size(); w++) starts[w] = w;
    const std::vector<Ciphertext> m = pat.clear ? match_clear_at(sk, s, *pat.clear, starts) : match_encrypted_at(sk, s, *pat.enc, starts);
    const std::vector<Ciphertext> seen = prefix_or(sk, m); // size L - P + 1; beyond it seen stays seen.back()
    r.found = seen.back();
    KeepLuts keep;
    FheString first(L), second(L); // in search order: before / after the first occurrence
    for (size_t j = 0; j < L; j++) {
        const Ciphertext &sj = seen[std::min(j, seen.size() - 1)];
        first[j] = keep_char_if(sk, s[j], sj, 0, keep);
        second[j] = j >= P ? keep_char_if(sk, s[j], seen[std::min(j - P, seen.size() - 1)], 1, keep) : trivial_radix(sk, 0, 4);
    }
    if (reverse) {
        std::reverse(first.begin(), first.end());
        std::reverse(second.begin(), second.end());
        r.before = std::move(second);
        r.after = std::move(first);
    } else {
        r.before = std::move(first);
        r.after = std::move(second);
    }
    return r;
}

// ---- replace / replacen (replace.rs:20-52) -------------------------------------------------------------
// The reference splits and re-joins: (L + 1) parts of L chars interleaved with copies of `to`, O(L^2) content.
// Here the taken occurrences are overwritten in place: position j keeps its char outside the separators, takes
// to[o] at offset o of a separator (gated by the separator's flag: a leveled product for a clear `to`, one
// bivariate PBS per block for an encrypted one), and when `to` is longer than the pattern its tail goes to
// |to| - |pattern| extra slots behind the separator's last position.  Content: L (1 + max(0, |to| - |pattern|))
// chars, padding anywhere.  (A pattern with padding has an encrypted length: then all of `to` goes behind the
// occurrence's last char, L (1 + |to|) chars.)
struct ReplaceTarget {
    const std::string *clear = nullptr;
    const FheString *enc = nullptr;
    size_t size() const { return clear ? clear->size() : enc->size(); }
};
inline PaddedString string_replace(ServerKey &sk, const FheString &s, Padding pad, const SplitPattern &from, const ReplaceTarget &to,
                                   bool limited, uint64_t n_clear, const RadixCiphertext *n_enc)
{
    if (pad != Padding::None && pad != Padding::Final) throw std::runtime_error("replace: normalize the padding first");
    if (from.enc && from.enc_padding != Padding::None && from.enc_padding != Padding::Final)
        throw std::runtime_error("replace: normalize the padding of the pattern first");
    const size_t L = s.size(), P = from.size(), T = to.size();
    if (L + 1 > sk.max_index()) throw std::runtime_error("replace: more occurrences than the count radix can say (" + std::to_string(sk.max_index()) + " with " + std::to_string(sk.index_blocks()) + " blocks: tfhe_hip_host_key_set_num_blocks)");
    const bool known_empty = P == 0;
    const PatternShape sh = pattern_shape(sk, from);
    PaddedString r;
    if (limited && !n_enc && n_clear == 0) return {s, pad};
    if (!known_empty && !sh.padded && (L == 0 || P > L)) return {s, pad};
    // The EMPTY pattern (known, or an encrypted pattern that is padding only) matches in front of the string and behind
    // every real char: "abc".replace("", "-") == "-a-b-c-" (the reference gets there through its empty-pattern split,
    // replace.rs:20-52 over split.rs:283-312).  Layout: T slots in FRONT of the string, then every char followed by T slots.
    const bool empty_aware = known_empty || sh.padded;
    const Ciphertext is_empty = known_empty ? sk.create_trivial(1) : (sh.padded ? boolean_not_leveled(sk, sh.real[0]) : sk.create_trivial(0));
    const size_t min_len = known_empty ? 1 : sh.min_len();
    std::vector<Ciphertext> ends = ends_of(sk, taken_separators(sk, s, from, sh), sh);
    auto max_before = [min_len](size_t j) { return (uint64_t)(j / min_len); };
    const LookupTable both2 = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    // events of the empty pattern: front (rank 0), behind real char j (rank j + 1); replacen keeps rank < n
    Ciphertext ev_front = is_empty;
    std::vector<Ciphertext> ev_char(empty_aware ? L : 0, is_empty);
    if (empty_aware && pad == Padding::Final && L > 0) {
        const std::vector<Ciphertext> z = char_is_zero_flags(sk, s);
        for (size_t j = 0; j < L; j++) ev_char[j] = boolean_and(sk, is_empty, boolean_not_leveled(sk, z[j]), both2);
    }
    if (limited) { // the first n occurrences: rank < n
        if (n_enc) {
            if (!known_empty) ends = limit_events(sk, ends, prefix_counts(sk, ends, max_before), 0, n_enc);
            if (empty_aware) {
                const Comparator cmp(sk);
                std::vector<Ciphertext> signs;
                for (size_t rank = 0; rank <= L; rank++) signs.push_back(cmp.scalar_compare(*n_enc, rank));
                std::vector<std::function<bool(uint64_t)>> gt{[](uint64_t x) { return x == IS_SUPERIOR; }};
                signs = cmp.map_signs(std::move(signs), gt);
                ev_front = boolean_and(sk, ev_front, signs[0], both2);
                for (size_t j = 0; j < L; j++) ev_char[j] = boolean_and(sk, ev_char[j], signs[j + 1], both2);
            }
        } else {
            if (!known_empty && n_clear < L / min_len + 1) ends = limit_events(sk, ends, prefix_counts(sk, ends, max_before), n_clear, nullptr);
            for (size_t j = 0; j < ev_char.size(); j++)
                if (j + 1 >= n_clear) ev_char[j] = sk.create_trivial(0);
        }
    }
    const std::vector<Ciphertext> gate = outside_separators(sk, ends, sh);
    KeepLuts keep;
    constexpr uint64_t MAX_OUTPUT_NOISE = 10; // outputs leave as (almost) fresh blocks: clean the longer leveled sums
    const LookupTable ident = sk.generate_lookup_table([](uint64_t x) { return x & 3; });
    auto put = [&](RadixCiphertext &slot, size_t o, const Ciphertext &flag) { // slot += to[o] * flag
        if (flag.is_trivial() && flag.body == 0) return;
        if (to.clear) {
            const uint8_t byte = (uint8_t)(*to.clear)[o];
            for (size_t b = 0; b < 4; b++) {
                Ciphertext f = flag;
                sk.unchecked_scalar_mul_assign(f, (byte >> (2 * b)) & 3);
                sk.unchecked_add_assign(slot[b], f);
            }
        } else {
            const RadixCiphertext sel = keep_char_if(sk, (*to.enc)[o], flag, 1, keep);
            for (size_t b = 0; b < 4; b++) sk.unchecked_add_assign(slot[b], sel[b]);
        }
    };
    auto finish = [&](RadixCiphertext &slot) {
        for (auto &b : slot) {
            b.degree = std::min<uint64_t>(b.degree, 3);
            if (b.noise_level > MAX_OUTPUT_NOISE) apply_lut(sk, b, ident);
        }
    };
    r.padding = Padding::Anywhere;
    if (empty_aware) {
        // the length of the occurrence is encrypted (or zero): every char of `to` goes to its own slot behind the END of the
        // separator -- or, for the empty pattern, in front of the string and behind every real char
        const size_t stride = 1 + T;
        r.content.assign(T + L * stride, trivial_radix(sk, 0, 4));
        for (size_t o = 0; o < T; o++) {
            put(r.content[o], o, ev_front);
            finish(r.content[o]);
        }
        for (size_t j = 0; j < L; j++) {
            r.content[T + j * stride] = known_empty ? s[j] : keep_char_if(sk, s[j], gate[j], 1, keep);
            Ciphertext here = ends[j]; // exclusive events: a leveled sum
            sk.unchecked_add_assign(here, ev_char[j]);
            here.degree = 1;
            for (size_t o = 0; o < T; o++) {
                put(r.content[T + j * stride + 1 + o], o, here);
                finish(r.content[T + j * stride + 1 + o]);
            }
        }
        return r;
    }
    const size_t extra = T > P ? T - P : 0, stride = 1 + extra;
    r.content.assign(L * stride, trivial_radix(sk, 0, 4));
    for (size_t j = 0; j < L; j++) {
        RadixCiphertext &slot = r.content[j * stride];
        slot = keep_char_if(sk, s[j], gate[j], 1, keep);
        // char j is offset o of the separator that ends at j + P - 1 - o
        for (size_t o = 0; o < std::min(P, T) && o <= j; o++)
            if (j + P - 1 - o < L) put(slot, o, ends[j + P - 1 - o]);
        finish(slot);
        for (size_t e = 0; e < extra; e++) {
            put(r.content[j * stride + 1 + e], P + e, ends[j]);
            finish(r.content[j * stride + 1 + e]);
        }
    }
    return r;
}

// ---- the remaining padding helpers and the encrypted-char trims (change_padding.rs:36-140, trim.rs:135-200) ---------
// eq with an ENCRYPTED char for every character (trim_*_encrypted): packed-pair differences, all-true per char
inline std::vector<Ciphertext> char_eq_flags_encrypted(ServerKey &sk, const FheString &s, const RadixCiphertext &ch)
{
    const LookupTable eq = equality_lut(sk);
    const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    std::vector<Ciphertext> out(s.size());
    for (size_t n = 0; n < s.size(); n++) {
        std::vector<Ciphertext> d;
        stage_eq_blocks(sk, s[n], ch, d);
        for (auto &x : d) apply_lut(sk, x, eq);
        out[n] = d[0];
        sk.unchecked_add_assign(out[n], d[1]);
        apply_lut(sk, out[n], both);
    }
    return out;
}
inline FheStringOut string_trim_encrypted_char(ServerKey &sk, const FheString &s, const RadixCiphertext &ch, bool left, bool right)
{
    if (ch.size() != 4) throw std::runtime_error("a char is a 4-block radix ciphertext");
    return trim_flagged(sk, s, char_eq_flags_encrypted(sk, s, ch), left, right);
}
// push_padding_to_start (change_padding.rs:36-54): the mirror image of push_padding_to_end
inline FheString push_padding_to_start(ServerKey &sk, const FheString &s)
{
    return reversed_string(push_padding_to_end(sk, reversed_string(s)));
}
// pop_first_non_zero_char / pop_last_non_zero_char (change_padding.rs:74-140): the first (last) real char of the
// content, which becomes a padding zero.  One-hot "first real char" flags, a one-hot select, one masking launch.
struct PoppedChar {
    RadixCiphertext ch;
    FheString content;
};
inline PoppedChar pop_non_zero_char(ServerKey &sk, const FheString &s_in, bool last)
{
    PoppedChar r;
    r.ch = trivial_radix(sk, 0, 4);
    r.content = s_in;
    if (s_in.empty()) return r;
    const FheString s = last ? reversed_string(s_in) : s_in;
    std::vector<Ciphertext> real = char_is_zero_flags(sk, s);
    for (auto &f : real) f = boolean_not_leveled(sk, f);
    const FirstMatch fm = first_one_hot(sk, real);
    KeepLuts keep;
    std::vector<std::vector<Ciphertext>> sel(4);
    FheString rest(s.size());
    for (size_t j = 0; j < s.size(); j++) {
        const RadixCiphertext c = keep_char_if(sk, s[j], fm.first[j], 1, keep);
        for (size_t b = 0; b < 4; b++) sel[b].push_back(c[b]);
        rest[j] = keep_char_if(sk, s[j], fm.first[j], 0, keep);
    }
    for (size_t b = 0; b < 4; b++) r.ch[b] = sum_exclusive_blocks(sk, std::move(sel[b]));
    r.content = last ? reversed_string(rest) : rest;
    return r;
}

// push_padding_to_end for short strings: the rank of a real char (the number of real chars before it) IS its final
// position, so the compaction is the scatter of the split machinery with the ranks as indices -- rank flags on the
// counts (plain leveled sums up to 15 real chars), one masking PBS per block and (rank, position) pair, then the
// carry-free sum over the positions.  2 L^2 bootstraps like the shift network at these sizes, but 5 - 7 levels
// instead of 9 - 11: every operation on a string with padding starts with this.
inline FheString push_padding_to_end_by_rank(ServerKey &sk, const FheString &s)
{
    const size_t L = s.size();
    std::vector<Ciphertext> real = char_is_zero_flags(sk, s);
    for (auto &f : real) f = boolean_not_leveled(sk, f);
    const PrefixCounts pc = prefix_counts(sk, real, [](size_t j) { return (uint64_t)j; });
    std::vector<const RadixCiphertext *> index(L);
    std::vector<uint64_t> bound(L);
    std::vector<const Ciphertext *> direct(L, nullptr);
    for (size_t j = 0; j < L; j++) {
        index[j] = &pc.k[j];
        bound[j] = pc.bound[j];
        if (pc.has_direct[j]) direct[j] = &pc.direct[j];
    }
    // by_rank[q][j] = s[j] iff it is a real char of rank q
    const std::vector<FheString> by_rank = scatter_by_index(sk, s, index, bound, &real, L, 0, false, &direct);
    FheString out(L, trivial_radix(sk, 0, 4));
    for (size_t q = 0; q < L; q++)
        for (size_t b = 0; b < 4; b++) {
            std::vector<Ciphertext> terms;
            for (size_t j = q; j < L; j++) // a char never moves right: rank <= position
                if (!(by_rank[q][j][b].is_trivial() && by_rank[q][j][b].body == 0)) terms.push_back(by_rank[q][j][b]);
            out[q][b] = sum_exclusive_blocks(sk, std::move(terms));
        }
    return out;
}

} // namespace host
} // namespace tfhe_hip
