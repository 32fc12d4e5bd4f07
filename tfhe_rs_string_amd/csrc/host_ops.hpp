// host_ops.hpp -- host-side mirror of the reference's shortint / integer / FheString interface for the
// KS-PBS path, re-scheduled for the GPU: every dependency level of an op-graph becomes ONE batched
// launch instead of one PBS per rayon task, and the ciphertexts stay in HBM between the levels.
//
//   * leveled operations edit lazy linear combinations (Ciphertext = terms over shared Blocks + body);
//   * apply_lookup_table_batch DEFERS: the result is a reference to a pending Block; flush() runs the pending DAG
//     level by level, filling each launch to the kernels' round sizes with bootstraps that have slack;
//   * a level's input rows are formed on the device from resident blocks (tfhe_hip_device_lincomb_rows) -- or, for
//     rows that only combine the caller's arrays, by the host threads into page-locked memory --, bootstrapped by
//     tfhe_hip_ks_pbs_batch_device, and its outputs stay in a pooled device buffer; lookup tables live in a device
//     arena; results are gathered on the device and downloaded once (write_outputs).
//     (TFHE_HIP_HOST_STAGING=1: the older path that stages every level through host memory.)
//
// Mirrors (names, argument meaning, degree / noise bookkeeping, trivial-ciphertext shortcut):
//   shortint::Ciphertext                         tfhe/src/shortint/ciphertext/mod.rs:263-270, 371-374
//   ServerKey::generate_lookup_table(_bivariate) tfhe/src/shortint/server_key/mod.rs:383-399, bivariate_pbs.rs:71-131
//   apply_lookup_table / trivial_pbs_assign      server_key/mod.rs:457-476, 763-857
//   unchecked_add_assign / unchecked_scalar_mul  server_key/add.rs:520-524, scalar_mul.rs:206
//   unchecked_apply_lookup_table_bivariate       bivariate_pbs.rs:167-181
//   message_extract / carry_extract / create_trivial  server_key/mod.rs:539-704
//   integer: unchecked_eq_parallelized           tfhe/src/integer/server_key/radix_parallel/comparison.rs:10-33
//            are_all_comparisons_block_true / is_at_least_one_...   scalar_comparison.rs:147-230
//            unchecked_mul_assign_parallelized   radix_parallel/mul.rs:329-414
//            unchecked_sum_ciphertexts_vec_parallelized, propagate_single_carry_parallelized_low_latency,
//            compute_prefix_sum_hillis_steele, generate_init_carry_array   radix_parallel/add.rs:487-603, 724-956
//   strings: StringServerKey::eq / contains      tfhe/examples/fhe_strings/server_key/comparisons.rs:20-212,
//            contains.rs:18-91, pattern.rs:246-320 (unpadded strings, clear lengths)
// Sequential AND / OR chains of the string layer are re-associated into the 15-ary sum+LUT trees the
// integer layer already uses (are_all_comparisons_block_true): decrypted results are identical,
// ciphertext bits are not.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/tfhe_hip.h"

namespace tfhe_hip {
namespace host {

// Storage of one or more big LWE ciphertexts: a caller's input array, the output block of one batched
// launch, or the not yet computed output of ONE deferred bootstrap (`node` set, `data` null until the
// scheduler has run it; then a view into the launch's output block, kept alive by `storage`).
// Transfer buffers come from tfhe_hip_host_alloc (page-locked on the GPU build: no staging copy, pipelined large
// batches) and are recycled: pinning memory is slow, levels of the same size repeat.
// pool size classes: powers of two (at least 32 KiB), multiples of 128 MiB beyond that -- a handful of classes, so
// the buffers of one operation are found again by the next
inline size_t pool_size_class(size_t words)
{
    constexpr size_t MIN = 4096, STEP = (size_t)16 << 20;
    if (words <= MIN) return MIN;
    if (words > STEP) return (words + STEP - 1) / STEP * STEP;
    size_t c = MIN;
    while (c < words) c <<= 1;
    return c;
}
class HostPool {
  public:
    ~HostPool()
    {
        for (auto &f : free_) tfhe_hip_host_free(f.second);
    }
    uint64_t *acquire(size_t words, size_t *cap)
    {
        words = pool_size_class(words);
        {
            std::lock_guard<std::mutex> lock(mu_);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].first == words) {
                    auto f = free_[i];
                    free_.erase(free_.begin() + i);
                    *cap = f.first;
                    return f.second;
                }
        }
        void *p = nullptr;
        if (tfhe_hip_host_alloc(&p, words * sizeof(uint64_t))) throw std::runtime_error(tfhe_hip_last_error());
        *cap = words;
        return static_cast<uint64_t *>(p);
    }
    void release(uint64_t *p, size_t cap)
    {
        std::lock_guard<std::mutex> lock(mu_);
        if (free_.size() >= 48) { // keep a working set (an operation holds a few buffers per level); drop the smallest
            size_t small = 0;
            for (size_t i = 1; i < free_.size(); i++)
                if (free_[i].first < free_[small].first) small = i;
            if (free_[small].first < cap) {
                tfhe_hip_host_free(free_[small].second);
                free_[small] = {cap, p};
            } else {
                tfhe_hip_host_free(p);
            }
            return;
        }
        free_.push_back({cap, p});
    }

  private:
    std::mutex mu_;
    std::vector<std::pair<size_t, uint64_t *>> free_;
};

// Device buffers of the resident mode (level outputs, uploaded operands), recycled the same way: hipMalloc / hipFree
// synchronise the device.  Everything the scheduler does runs on one stream in order, so a recycled buffer is
// rewritten only after the kernels that read its previous contents.
class DevicePool {
  public:
    explicit DevicePool(tfhe_hip_server_key *sk) : sk_(sk) {}
    ~DevicePool()
    {
        for (auto &f : free_) tfhe_hip_device_free(f.second);
    }
    uint64_t *acquire(size_t words, size_t *cap)
    {
        words = pool_size_class(words);
        {
            std::lock_guard<std::mutex> lock(mu_);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].first == words) {
                    auto f = free_[i];
                    free_.erase(free_.begin() + i);
                    *cap = f.first;
                    return f.second;
                }
        }
        uint64_t *p = nullptr;
        if (tfhe_hip_device_alloc(sk_, &p, words)) throw std::runtime_error(tfhe_hip_last_error());
        *cap = words;
        return p;
    }
    void release(uint64_t *p, size_t cap)
    {
        std::lock_guard<std::mutex> lock(mu_);
        if (free_.size() >= 48) { // keep a working set; drop the smallest
            size_t small = 0;
            for (size_t i = 1; i < free_.size(); i++)
                if (free_[i].first < free_[small].first) small = i;
            if (free_[small].first < cap) {
                tfhe_hip_device_free(free_[small].second);
                free_[small] = {cap, p};
            } else {
                tfhe_hip_device_free(p);
            }
            return;
        }
        free_.push_back({cap, p});
    }

  private:
    tfhe_hip_server_key *sk_;
    std::mutex mu_;
    std::vector<std::pair<size_t, uint64_t *>> free_;
};

struct PbsNode;
struct Block {
    const uint64_t *data = nullptr;
    size_t words = 0;                  // size of the host data (caller's arrays: what an upload copies)
    const uint64_t *ddata = nullptr;   // device copy (resident mode): a level's output rows, or the uploaded operand
    std::shared_ptr<uint64_t> down;    // ... and its owner
    bool staged = false;               // queued for upload in the level being staged
    std::shared_ptr<uint64_t> own;     // set when the block owns its storage (returned to the pool on release)
    std::shared_ptr<const Block> storage;
    std::unique_ptr<PbsNode> node;     // pending bootstrap that produces this block
    size_t depth = 0;                  // dependency level of the pending bootstrap (0: data available)
    Block();
    ~Block();
};
struct Term {
    uint64_t coef;
    std::shared_ptr<Block> blk;
    size_t off; // coef * blk->data[off .. off + big_size)
};

// shortint::Ciphertext (ciphertext/mod.rs:263-270).  The LWE data is a lazily evaluated linear
// combination: the leveled operations (add, scalar_mul, bivariate packing) only edit the term list and
// the combination is evaluated once, straight into the upload buffer of the next batched launch.
struct Ciphertext {
    std::vector<Term> terms;
    uint64_t body = 0;        // + (0, ..., 0, body): the trivial part
    uint64_t degree = 0;
    uint64_t noise_level = 0; // NoiseLevel::ZERO = 0, NOMINAL = 1 (ciphertext/mod.rs)
    // ciphertext/mod.rs:371-374: all-zero mask and NoiseLevel::ZERO
    bool is_trivial() const { return noise_level == 0 && terms.empty(); }
    size_t depth() const
    {
        size_t d = 0;
        for (const Term &t : terms) d = std::max(d, t.blk->depth);
        return d;
    }
    // all blocks must hold data (ServerKey::flush has run)
    void materialize(uint64_t *out, size_t big) const
    {
        if (terms.empty()) {
            std::fill(out, out + big, (uint64_t)0);
        } else {
            const uint64_t *a = terms[0].blk->data + terms[0].off;
            const uint64_t ca = terms[0].coef;
            if (terms.size() == 1) {
                if (ca == 1) std::copy(a, a + big, out);
                else for (size_t i = 0; i < big; i++) out[i] = ca * a[i];
            } else {
                const uint64_t *b = terms[1].blk->data + terms[1].off;
                const uint64_t cb = terms[1].coef;
                for (size_t i = 0; i < big; i++) out[i] = ca * a[i] + cb * b[i];
                for (size_t t = 2; t < terms.size(); t++) {
                    const uint64_t *c = terms[t].blk->data + terms[t].off;
                    const uint64_t cc = terms[t].coef;
                    for (size_t i = 0; i < big; i++) out[i] += cc * c[i];
                }
            }
        }
        out[big - 1] += body;
    }
};

struct LookupTable {
    std::shared_ptr<const std::vector<uint64_t>> acc; // GLWE accumulator, (k+1) N; shared with pending bootstraps
    uint64_t degree = 0;
};
// a deferred bootstrap: output = PBS_lut(input)
struct PbsNode {
    Ciphertext input;
    LookupTable lut;
};
inline Block::Block() = default;
inline Block::~Block() = default;

// static work split over a few host threads (staging of large levels)
template <typename F> inline void parallel_rows(size_t n, size_t min_rows_per_thread, F &&fn)
{
    size_t nt = std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())),
                                 n / std::max<size_t>(1, min_rows_per_thread));
    if (nt <= 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nt);
    for (size_t t = 0; t < nt; t++) th.emplace_back([&, t] { fn(n * t / nt, n * (t + 1) / nt); });
    for (auto &x : th) x.join();
}


using RadixCiphertext = std::vector<Ciphertext>; // blocks, least significant first
using FheString = std::vector<RadixCiphertext>;  // unpadded: one 4-block radix per ASCII char

class ServerKey {
  public:
    ServerKey(tfhe_hip_server_key *sk, const tfhe_hip_params &p) : sk_(sk), p_(p)
    {
        big_ = (size_t)p.glwe_dim * p.poly_size + 1;
        lut_len_ = (size_t)(p.glwe_dim + 1) * p.poly_size;
        delta_ = ((uint64_t)1 << 63) / ((uint64_t)p.message_modulus * p.carry_modulus);
        if (std::getenv("TFHE_HIP_HOST_ASAP") == nullptr) tfhe_hip_server_key_batch_rounds(sk, &round_small_, &round_large_);
        // resident mode (default): ciphertexts stay on the device between the levels; TFHE_HIP_HOST_STAGING=1 selects the
        // older path that stages every level through host memory (kept for comparison)
        resident_ = std::getenv("TFHE_HIP_HOST_STAGING") == nullptr;
        if (const char *v = std::getenv("TFHE_HIP_LUT_SLOTS")) lut_capacity_ = std::max<size_t>(1, std::strtoull(v, nullptr, 10));
        dpool_ = std::make_shared<DevicePool>(sk);
    }
    const tfhe_hip_params &params() const { return p_; }
    size_t big_size() const { return big_; }
    uint64_t message_modulus() const { return p_.message_modulus; }
    uint64_t carry_modulus() const { return p_.carry_modulus; }
    // blocks of the radix ciphertexts that carry lengths, indices and counts (num_blocks of gen_keys_number_blocks,
    // fhe_strings/ciphertext.rs:113-126); the largest value they can say
    size_t index_blocks() const { return index_blocks_; }
    uint64_t max_index() const { return index_blocks_ >= 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * index_blocks_)) - 1); }
    void set_index_blocks(size_t nb)
    {
        if (p_.message_modulus != 4) throw std::runtime_error("num_blocks: the string layer is built on 2-bit message blocks");
        if (nb < 4 || nb > 16) throw std::runtime_error("num_blocks must be 4 .. 16 (4 blocks = one ASCII char = lengths up to 255)");
        index_blocks_ = nb;
    }

    LookupTable generate_lookup_table(const std::function<uint64_t(uint64_t)> &f) const
    {
        const uint32_t m = p_.message_modulus * p_.carry_modulus;
        std::vector<uint64_t> table(m);
        for (uint32_t i = 0; i < m; i++) table[i] = f(i);
        // equal tables share one accumulator: a level then uploads each distinct table once, however many
        // operations staged it
        std::lock_guard<std::mutex> lock(lut_mutex_);
        auto hit = lut_cache_.find(table);
        if (hit != lut_cache_.end()) return hit->second;
        LookupTable lut;
        auto acc = std::make_shared<std::vector<uint64_t>>(lut_len_);
        if (tfhe_hip_lut_generate(&p_, table.data(), acc->data(), &lut.degree))
            throw std::runtime_error(tfhe_hip_last_error());
        lut.acc = std::move(acc);
        if (lut_cache_.size() < 4096) lut_cache_.emplace(std::move(table), lut);
        return lut;
    }
    // bivariate_pbs.rs:71-131, factor = message_modulus
    LookupTable generate_lookup_table_bivariate(const std::function<uint64_t(uint64_t, uint64_t)> &f) const
    {
        const uint64_t factor = p_.message_modulus, mm = p_.message_modulus;
        return generate_lookup_table([&](uint64_t in) { return f((in / factor) % mm, (in % factor) % mm); });
    }
    Ciphertext create_trivial(uint64_t value) const
    {
        Ciphertext c;
        const uint64_t v = value % p_.message_modulus;
        c.body = v * delta_;
        c.degree = v;
        c.noise_level = 0;
        return c;
    }
    // a view of `lwe` (big_size words); the memory must outlive the ciphertext (the C-ABI entry points
    // wrap the caller's arrays for the duration of one call)
    Ciphertext from_raw(const uint64_t *lwe, uint64_t degree, uint64_t noise = 1) const
    {
        if (noise == 0 && std::all_of(lwe, lwe + big_ - 1, [](uint64_t x) { return x == 0; })) {
            Ciphertext c; // a trivial ciphertext (ciphertext/mod.rs:371-374)
            c.body = lwe[big_ - 1];
            c.degree = degree;
            return c;
        }
        auto blk = std::make_shared<Block>();
        blk->data = lwe;
        blk->words = big_;
        Ciphertext c;
        c.terms.push_back(Term{1, std::move(blk), 0});
        c.degree = degree;
        c.noise_level = noise;
        return c;
    }
    // rows [0, rows) of one array as ciphertexts sharing a single block
    std::vector<Ciphertext> from_raw_rows(const uint64_t *lwe, size_t rows, uint64_t degree, uint64_t noise = 1) const
    {
        auto blk = std::make_shared<Block>();
        blk->data = lwe;
        blk->words = rows * big_;
        std::vector<Ciphertext> out(rows);
        for (size_t r = 0; r < rows; r++) {
            out[r].terms.push_back(Term{1, blk, r * big_});
            out[r].degree = degree;
            out[r].noise_level = noise;
        }
        return out;
    }
    // Output contract (shortint/ciphertext/mod.rs:263-270, server_key/mod.rs:855-856): the reference hands every result
    // back as Ciphertext{degree, noise_level = NOMINAL}.  With strict outputs (the default) a result whose noise level is
    // above NOMINAL -- an op-graph that ends in a leveled step: the selected side of a cmux, a case-adjusted block, ... --
    // goes through one more, cleaning bootstrap (identity table, same degree) before it is written, so that every
    // block handed out is a direct bootstrap output (level 1), the caller's own input block (its level) or trivial (0).
    // Without, such results are handed out as they are and output_metadata() reports their level.
    void set_strict_outputs(bool on) { strict_outputs_ = on; }
    bool strict_outputs() const { return strict_outputs_; }
    void to_raw(const Ciphertext &c_in, uint64_t *out)
    {
        Ciphertext c = c_in;
        if (strict_outputs_ && c.noise_level > 1) {
            const uint64_t degree = c.degree;
            const LookupTable clean = generate_lookup_table([](uint64_t x) { return x; });
            apply_lookup_table_batch({&c}, {&clean});
            c.degree = std::min<uint64_t>(degree, clean.degree); // the value did not change
        }
        {
            std::lock_guard<std::mutex> lock(lut_mutex_);
            out_meta_[out] = {c.degree, c.noise_level};
        }
        if (batching_) { // written by end_batch, together with everything else that was staged since begin_batch
            deferred_.push_back({c, out});
            return;
        }
        flush();
        std::vector<std::pair<Ciphertext, uint64_t *>> one{{c, out}};
        write_outputs(one);
    }
    // degree and noise level of the block written to `out` by the last call (or in the last batch); false if unknown
    bool output_metadata(const uint64_t *out, uint64_t *degree, uint64_t *noise_level) const
    {
        std::lock_guard<std::mutex> lock(lut_mutex_);
        auto hit = out_meta_.find(out);
        if (hit == out_meta_.end()) return false;
        *degree = hit->second.first;
        *noise_level = hit->second.second;
        return true;
    }
    // Batch mode: between begin_batch and end_batch the operations only STAGE their bootstraps and register
    // their outputs; end_batch runs the union of all their graphs level by level -- many independent
    // operations (a column of strings against one pattern, ...) then share their launches and reach the
    // throughput regime of the kernels -- and writes the outputs.  Input and output arrays must stay valid
    // until end_batch.
    void begin_batch()
    {
        if (batching_) throw std::runtime_error("a batch is already open on this host key");
        batching_ = true;
        {
            std::lock_guard<std::mutex> lock(lut_mutex_);
            out_meta_.clear(); // metadata describes the outputs of the last call / batch
        }
    }
    void end_batch()
    {
        if (!batching_) throw std::runtime_error("no batch is open on this host key");
        batching_ = false;
        std::vector<std::pair<Ciphertext, uint64_t *>> outs;
        outs.swap(deferred_);
        try {
            static const bool trace = std::getenv("TFHE_HIP_HOST_TRACE") != nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            flush();
            const auto t1 = std::chrono::steady_clock::now();
            write_outputs(outs);
            if (trace) {
                auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                std::fprintf(stderr, "[host] batch: flush %.2f ms, %zu outputs written in %.2f ms\n", ms(t0, t1), outs.size(),
                             ms(t1, std::chrono::steady_clock::now()));
            }
        } catch (...) {
            pending_.clear();
            throw;
        }
        pending_.clear();
    }
    // leaves batch mode without running anything (an operation failed while it was staging)
    void abort_batch()
    {
        batching_ = false;
        deferred_.clear();
        pending_.clear();
    }
    bool batching() const { return batching_; }
    void unchecked_add_assign(Ciphertext &l, const Ciphertext &r) const
    {
        for (const Term &t : r.terms) {
            bool merged = false;
            for (Term &u : l.terms)
                if (u.blk == t.blk && u.off == t.off) { u.coef += t.coef; merged = true; break; }
            if (!merged) l.terms.push_back(t);
        }
        l.body += r.body;
        l.degree += r.degree;
        l.noise_level += r.noise_level;
    }
    void unchecked_scalar_mul_assign(Ciphertext &c, uint64_t s) const
    {
        for (Term &t : c.terms) t.coef *= s;
        c.terms.erase(std::remove_if(c.terms.begin(), c.terms.end(), [](const Term &t) { return t.coef == 0; }), c.terms.end());
        c.body *= s;
        c.degree *= s;
        c.noise_level *= s;
    }
    // core_crypto lwe_ciphertext_sub_assign (the "true" subtraction comparator.rs:193-224 asks for)
    void lwe_sub_assign(Ciphertext &l, const Ciphertext &r) const
    {
        Ciphertext n = r;
        for (Term &t : n.terms) t.coef = (uint64_t)0 - t.coef;
        n.body = (uint64_t)0 - n.body;
        const uint64_t deg = l.degree;
        unchecked_add_assign(l, n);
        l.degree = deg;
    }
    // server_key/scalar_add.rs: the clear value goes into the body
    void unchecked_scalar_add_assign(Ciphertext &c, uint64_t scalar) const
    {
        c.body += scalar * delta_;
        c.degree += scalar;
    }
    void lwe_plaintext_sub_assign(Ciphertext &c, uint64_t scalar) const { c.body -= scalar * delta_; }
    // lhs <- lhs * factor + rhs (the leveled half of unchecked_apply_lookup_table_bivariate_assign)
    void bivariate_pack_assign(Ciphertext &l, const Ciphertext &r) const
    {
        if (r.degree + 1 > p_.message_modulus) throw std::runtime_error("bivariate PBS: rhs degree exceeds the packing factor");
        unchecked_scalar_mul_assign(l, p_.message_modulus);
        unchecked_add_assign(l, r);
    }

    // apply luts[i] to *cts[i]: trivial ciphertexts take the table-lookup shortcut at once
    // (server_key/mod.rs:763-781); every other bootstrap is DEFERRED: the ciphertext becomes a reference
    // to a pending output block whose dependency level is one more than its deepest input.  `flush`
    // runs the pending bootstraps level by level -- all those of one level in ONE batched launch,
    // whichever operation or sub-graph staged them -- and drops those nobody refers to any more.
    void apply_lookup_table_batch(const std::vector<Ciphertext *> &cts, const std::vector<const LookupTable *> &luts)
    {
        if (cts.size() != luts.size()) throw std::runtime_error("apply_lookup_table_batch: size mismatch");
        for (size_t i = 0; i < cts.size(); i++) {
            Ciphertext &c = *cts[i];
            const LookupTable &lut = *luts[i];
            if (c.is_trivial()) {
                trivial_pbs_assign(c, lut);
                continue;
            }
            auto blk = std::make_shared<Block>();
            blk->depth = c.depth() + 1;
            blk->node.reset(new PbsNode{std::move(c), lut});
            pending_.push_back(blk);
            c = Ciphertext();
            c.terms.push_back(Term{1, std::move(blk), 0});
            c.degree = lut.degree;
            c.noise_level = 1; // NoiseLevel::NOMINAL
        }
    }

    // Runs every pending bootstrap that is still referenced: one launch per dependency level, and WHICH level a
    // bootstrap with slack runs in is chosen for the hardware.  A launch costs whole rounds of the kernels (one
    // ciphertext per CU for the latency kernel, four for the throughput kernel: tfhe_hip_server_key_batch_rounds), so
    // a level that must run n bootstraps (those on the critical path: as-late-as-possible level == this level) is
    // filled up to the next round boundary with ready bootstraps that could wait (earliest deadline first): they ride
    // for free, and the later levels shrink.
    void flush()
    {
        if (pending_.empty()) return;
        std::vector<std::shared_ptr<Block>> work;
        work.swap(pending_);
        // dead-code elimination, consumers before producers
        for (size_t i = work.size(); i-- > 0;)
            if (work[i].use_count() == 1) work[i].reset();
        std::vector<Block *> nodes; // creation order: producers before consumers
        size_t max_depth = 0;
        for (const auto &b : work)
            if (b) {
                nodes.push_back(b.get());
                max_depth = std::max(max_depth, b->depth);
            }
        const size_t n = nodes.size();
        if (n == 0) return;
        std::unordered_map<const Block *, uint32_t> id;
        id.reserve(2 * n);
        for (size_t i = 0; i < n; i++) id.emplace(nodes[i], (uint32_t)i);
        std::vector<std::vector<uint32_t>> consumers(n);
        std::vector<uint32_t> waiting(n, 0);
        for (size_t i = 0; i < n; i++)
            for (const Term &t : nodes[i]->node->input.terms)
                if (t.blk->node) { // still pending: produced in this flush
                    consumers[id.at(t.blk.get())].push_back((uint32_t)i);
                    waiting[i]++;
                }
        std::vector<size_t> alap(n, max_depth);
        for (size_t i = n; i-- > 0;)
            for (uint32_t c : consumers[i]) alap[i] = std::min(alap[i], alap[c] - 1);
        std::vector<std::vector<uint32_t>> ready(max_depth + 1); // by deadline
        for (size_t i = 0; i < n; i++)
            if (waiting[i] == 0) ready[alap[i]].push_back((uint32_t)i);
        std::vector<uint32_t> level;
        std::vector<Block *> run;
        for (size_t step = 1; step <= max_depth; step++) {
            level.swap(ready[step]);
            ready[step].clear();
            if (level.empty()) continue;
            const size_t target = fill_target(level.size());
            for (size_t a = step + 1; a <= max_depth && level.size() < target; a++)
                while (!ready[a].empty() && level.size() < target) {
                    level.push_back(ready[a].back());
                    ready[a].pop_back();
                }
            run.clear();
            for (uint32_t i : level) run.push_back(nodes[i]);
            run_level(run);
            for (uint32_t i : level)
                for (uint32_t c : consumers[i])
                    if (--waiting[c] == 0) ready[alap[c]].push_back(c);
            level.clear();
        }
    }
    // forget pending work (end of an API call: everything returned has been materialised)
    void discard_pending()
    {
        if (!batching_) pending_.clear();
    }

    uint64_t pbs_count = 0, launches = 0;
    uint64_t uploaded_words_ = 0; // operands copied to the device (resident mode)

  private:
    // the size a level of n mandatory bootstraps is filled up to: whole rounds of the throughput kernel, the rest in
    // rounds of the latency kernel (the launch policy of tfhe_hip_ks_pbs_batch)
    size_t fill_target(size_t n) const
    {
        if (round_small_ == 0) return n;
        const size_t big = round_large_ ? round_large_ : 4 * round_small_;
        const size_t full = n / big * big, r = n - full;
        if (r == 0) return n;
        if (r <= big - round_small_) return full + (r + round_small_ - 1) / round_small_ * round_small_;
        return full + big;
    }
    // the device address of a term's row; operands that still live in the caller's arrays are uploaded on first use
    const uint64_t *device_row(const Term &t)
    {
        Block &b = *t.blk;
        if (!b.ddata) {
            if (!b.data || b.words == 0) throw std::logic_error("scheduler: a bootstrap was issued before its input");
            size_t cap = 0;
            b.down = dpooled(b.words, &cap);
            if (tfhe_hip_device_upload(sk_, b.down.get(), b.data, b.words * 8)) throw std::runtime_error(tfhe_hip_last_error());
            b.ddata = b.down.get();
            uploaded_words_ += b.words;
        }
        return b.ddata + t.off;
    }
    // CSR of the lazy linear combinations of `rows` ciphertexts
    struct Csr {
        std::vector<uint32_t> row_ptr;
        std::vector<uint64_t> coef, body;
        std::vector<const uint64_t *> src;
    };
    template <typename GetCt> Csr build_csr(size_t rows, GetCt &&ct_of)
    {
        Csr c;
        c.row_ptr.reserve(rows + 1);
        c.row_ptr.push_back(0);
        c.body.reserve(rows);
        for (size_t j = 0; j < rows; j++) {
            const Ciphertext &ct = ct_of(j);
            for (const Term &t : ct.terms) {
                c.coef.push_back(t.coef);
                c.src.push_back(device_row(t));
            }
            c.row_ptr.push_back((uint32_t)c.coef.size());
            c.body.push_back(ct.body);
        }
        return c;
    }
    void lincomb(uint64_t *d_dst, size_t rows, const Csr &c)
    {
        if (tfhe_hip_device_lincomb_rows(sk_, d_dst, rows, big_, c.row_ptr.data(), c.coef.data(), c.src.data(), c.body.data()))
            throw std::runtime_error(tfhe_hip_last_error());
    }
    static bool on_host(const Ciphertext &c)
    {
        for (const Term &t : c.terms)
            if (!t.blk->data) return false;
        return true;
    }
    // Resident level.  Rows that only combine the caller's own arrays (the first level of an operation) are evaluated by
    // the host threads into page-locked memory and uploaded as finished rows -- a packed-pair difference is one row
    // instead of four operand blocks -- UNLESS the arrays they read are smaller than the rows themselves (the sliding
    // windows of `contains` / `find` read every block of the string once per pattern char): then the arrays go up once
    // and these rows are formed on the device like the others.  The other rows are formed on the device from resident
    // blocks (operands they still need from the caller's arrays ride in the same staging buffer).  The outputs stay on
    // the device.
    void run_level_resident(const std::vector<Block *> &nodes_in)
    {
        static const bool trace = std::getenv("TFHE_HIP_HOST_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const size_t rows = nodes_in.size();
        bool form_on_device = false;
        {
            size_t host_rows = 0, fresh_words = 0;
            std::vector<Block *> seen;
            for (Block *b : nodes_in)
                if (on_host(b->node->input)) {
                    host_rows++;
                    for (const Term &t : b->node->input.terms) {
                        Block &o = *t.blk;
                        if (!o.ddata && !o.staged) {
                            o.staged = true;
                            seen.push_back(&o);
                            fresh_words += o.words;
                        }
                    }
                }
            for (Block *o : seen) o->staged = false;
            form_on_device = host_rows > 0 && 2 * fresh_words <= host_rows * big_;
        }
        std::vector<Block *> nodes;
        nodes.reserve(rows);
        if (!form_on_device)
            for (Block *b : nodes_in)
                if (on_host(b->node->input)) nodes.push_back(b);
        const size_t n_host = nodes.size();
        for (Block *b : nodes_in)
            if (form_on_device || !on_host(b->node->input)) nodes.push_back(b);
        // lookup tables live in a device arena, one slot each, uploaded the first time they are used
        std::vector<uint32_t> idx(rows);
        size_t n_tables = 0;
        {
            std::vector<const std::vector<uint64_t> *> fresh; // tables of this level that are not in the arena yet
            for (size_t j = 0; j < rows; j++) {
                const std::vector<uint64_t> *acc = nodes[j]->node->lut.acc.get();
                if (!lut_slots_.count(acc) && std::find(fresh.begin(), fresh.end(), acc) == fresh.end()) fresh.push_back(acc);
            }
            if (lut_slots_.size() + fresh.size() > lut_capacity_) { // recycle the arena: this level's tables all get new slots
                lut_slots_.clear();
                fresh.clear();
                for (size_t j = 0; j < rows; j++) {
                    const std::vector<uint64_t> *acc = nodes[j]->node->lut.acc.get();
                    if (std::find(fresh.begin(), fresh.end(), acc) == fresh.end()) fresh.push_back(acc);
                }
                if (fresh.size() > lut_capacity_) throw std::runtime_error("a level uses more distinct lookup tables than the device arena holds");
            }
            const std::vector<uint64_t> *last = nullptr;
            uint32_t last_slot = 0;
            for (size_t j = 0; j < rows; j++) {
                const auto &acc = nodes[j]->node->lut.acc;
                if (acc.get() != last) {
                    last = acc.get();
                    last_slot = lut_slot(acc, &n_tables);
                }
                idx[j] = last_slot;
            }
        }
        // operands of the device rows that are still in the caller's arrays
        std::vector<Block *> operands;
        size_t operand_words = 0;
        for (size_t j = n_host; j < rows; j++)
            for (const Term &t : nodes[j]->node->input.terms) {
                Block &b = *t.blk;
                if (!b.ddata && b.data && !b.staged) {
                    b.staged = true;
                    operands.push_back(&b);
                    operand_words += b.words;
                }
            }
        if (d_in_cap_ < rows * big_) {
            d_in_.reset();
            d_in_ = dpooled(rows * big_, &d_in_cap_);
        }
        // page-locked staging: [operand blocks][table indices] now, the rows evaluated on the host below
        const size_t idx_words = (rows + 1) / 2;
        size_t hcap = 0, icap = 0, out_cap = 0;
        {
            std::shared_ptr<uint64_t> stage = pooled(operand_words + idx_words, &hcap);
            uint64_t *h = stage.get();
            if (operand_words) {
                std::vector<size_t> at(operands.size());
                size_t o = 0;
                for (size_t i = 0; i < operands.size(); i++) { at[i] = o; o += operands[i]->words; }
                parallel_rows(operands.size(), 16, [&](size_t r0, size_t r1) {
                    for (size_t i = r0; i < r1; i++) std::copy(operands[i]->data, operands[i]->data + operands[i]->words, h + at[i]);
                });
                size_t dcap = 0;
                std::shared_ptr<uint64_t> d_ops = dpooled(operand_words, &dcap);
                if (tfhe_hip_device_upload(sk_, d_ops.get(), h, operand_words * 8)) throw std::runtime_error(tfhe_hip_last_error());
                for (size_t i = 0; i < operands.size(); i++) {
                    operands[i]->ddata = d_ops.get() + at[i];
                    operands[i]->down = d_ops;
                }
                uploaded_words_ += operand_words;
            }
            std::memcpy(h + operand_words, idx.data(), rows * 4);
            in_flight_.push_back(std::move(stage)); // page-locked: read by the DMA engine later; released after the next sync
        }
        std::shared_ptr<uint64_t> d_idx = dpooled(idx_words, &icap);
        if (tfhe_hip_device_upload(sk_, d_idx.get(), in_flight_.back().get() + operand_words, idx_words * 8)) throw std::runtime_error(tfhe_hip_last_error());
        std::shared_ptr<uint64_t> d_out = dpooled(rows * big_, &out_cap);
        auto launch = [&](size_t first, size_t count) {
            if (tfhe_hip_ks_pbs_batch_device(sk_, d_in_.get() + first * big_, d_luts_.get(), lut_capacity_,
                                             reinterpret_cast<const uint32_t *>(d_idx.get()) + first, d_out.get() + first * big_, count, nullptr))
                throw std::runtime_error(tfhe_hip_last_error());
        };
        // rows [first, first + count) evaluated by the host threads into page-locked memory and uploaded
        auto stage_host_rows = [&](size_t first, size_t count) {
            if (count == 0) return;
            size_t cap = 0;
            std::shared_ptr<uint64_t> stage = pooled(count * big_, &cap);
            uint64_t *h = stage.get();
            parallel_rows(count, 128, [&](size_t r0, size_t r1) {
                for (size_t j = r0; j < r1; j++) nodes[first + j]->node->input.materialize(h + j * big_, big_);
            });
            // on the key's transfer stream: the first piece waits for everything enqueued so far (the rows it overwrites
            // may still be read), the later ones only for the pieces before them -- they go up while the kernels of the
            // earlier rounds of this level run
            if (tfhe_hip_device_upload_overlapped(sk_, d_in_.get() + first * big_, h, count * big_ * 8, first == 0))
                throw std::runtime_error(tfhe_hip_last_error());
            in_flight_.push_back(std::move(stage));
        };
        // A big first level starts on the GPU while the host is still evaluating the rest of it: whole rounds of the
        // throughput kernel are launched as soon as their rows are up.
        const size_t round = round_large_ ? round_large_ : 1024;
        size_t done = 0;
        while (n_host - done >= 2 * round) {
            stage_host_rows(done, round);
            launch(done, round);
            done += round;
        }
        stage_host_rows(done, n_host - done);
        size_t n_terms = 0;
        if (n_host < rows) {
            const Csr csr = build_csr(rows - n_host, [&](size_t j) -> const Ciphertext & { return nodes[n_host + j]->node->input; });
            n_terms = csr.coef.size();
            lincomb(d_in_.get() + n_host * big_, rows - n_host, csr);
        }
        const auto t1 = std::chrono::steady_clock::now();
        launch(done, rows - done);
        for (size_t j = 0; j < rows; j++) {
            Block &b = *nodes[j];
            b.ddata = d_out.get() + j * big_;
            b.down = d_out;
            b.depth = 0;
            b.node.reset(); // releases the references to the inputs
        }
        pbs_count += rows;
        launches += 1;
        if (trace) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "[host] level: %zu PBS (%zu rows from the host, %zu operand blocks, %zu device terms), %zu new LUTs: staged in %.2f ms, "
                                 "enqueued in %.2f ms\n", rows, n_host, operands.size(), n_terms, n_tables, ms(t0, t1),
                         ms(t1, std::chrono::steady_clock::now()));
        }
    }
    // writes the requested ciphertexts to the caller's arrays.  Trivial ones and combinations of the caller's own
    // arrays are evaluated on the host; the others are gathered on the device (a whole row of a level's output needs no
    // arithmetic: contiguous runs of them are copied as they are), come down through page-locked memory and are copied
    // out by several host threads.
    void write_outputs(std::vector<std::pair<Ciphertext, uint64_t *>> &outs)
    {
        if (!resident_) {
            parallel_rows(outs.size(), 256, [&](size_t r0, size_t r1) {
                for (size_t j = r0; j < r1; j++) outs[j].first.materialize(outs[j].second, big_);
            });
            return;
        }
        std::vector<size_t> live;
        for (size_t j = 0; j < outs.size(); j++)
            if (!on_host(outs[j].first)) live.push_back(j);
        parallel_rows(outs.size(), 256, [&](size_t r0, size_t r1) {
            for (size_t j = r0; j < r1; j++)
                if (on_host(outs[j].first)) outs[j].first.materialize(outs[j].second, big_);
        });
        auto plain_row = [](const Ciphertext &c) -> const uint64_t * { // a row of a level's output as it is
            if (c.terms.size() != 1 || c.terms[0].coef != 1 || c.body != 0 || !c.terms[0].blk->ddata) return nullptr;
            return c.terms[0].blk->ddata + c.terms[0].off;
        };
        constexpr size_t CHUNK = 8192; // 128 MiB per transfer
        for (size_t c0 = 0; c0 < live.size(); c0 += CHUNK) {
            const size_t rows = std::min(CHUNK, live.size() - c0);
            size_t hcap = 0;
            std::shared_ptr<uint64_t> h_buf = pooled(rows * big_, &hcap);
            // runs of consecutive outputs that are consecutive device rows
            std::vector<std::pair<size_t, size_t>> runs; // (first, count)
            bool all_plain = true;
            for (size_t j = 0; j < rows && all_plain; j++) {
                const uint64_t *d = plain_row(outs[live[c0 + j]].first);
                if (!d) { all_plain = false; break; }
                if (!runs.empty()) {
                    const uint64_t *prev = plain_row(outs[live[c0 + j - 1]].first);
                    if (d == prev + big_) { runs.back().second++; continue; }
                }
                runs.push_back({j, 1});
            }
            if (all_plain && runs.size() <= 8) {
                for (const auto &r : runs)
                    if (tfhe_hip_device_download(sk_, h_buf.get() + r.first * big_, plain_row(outs[live[c0 + r.first]].first), r.second * big_ * 8))
                        throw std::runtime_error(tfhe_hip_last_error());
            } else {
                const Csr csr = build_csr(rows, [&](size_t j) -> const Ciphertext & { return outs[live[c0 + j]].first; });
                size_t dcap = 0;
                std::shared_ptr<uint64_t> d_buf = dpooled(rows * big_, &dcap);
                lincomb(d_buf.get(), rows, csr);
                if (tfhe_hip_device_download(sk_, h_buf.get(), d_buf.get(), rows * big_ * 8)) throw std::runtime_error(tfhe_hip_last_error());
            }
            const uint64_t *h = h_buf.get();
            parallel_rows(rows, 128, [&](size_t r0, size_t r1) {
                for (size_t j = r0; j < r1; j++) std::copy(h + j * big_, h + (j + 1) * big_, outs[live[c0 + j]].second);
            });
        }
        if (!in_flight_.empty()) {
            if (live.empty() && tfhe_hip_device_synchronize(sk_)) throw std::runtime_error(tfhe_hip_last_error());
            in_flight_.clear(); // a download has waited for everything enqueued before it
        }
    }
    void run_level(const std::vector<Block *> &nodes)
    {
        if (resident_) {
            run_level_resident(nodes);
            return;
        }
        for (const Block *b : nodes)
            for (const Term &t : b->node->input.terms)
                if (!t.blk->data) throw std::logic_error("scheduler: a bootstrap was issued before its input");
        static const bool trace = std::getenv("TFHE_HIP_HOST_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const size_t rows = nodes.size();
        std::vector<const std::vector<uint64_t> *> uniq;
        std::vector<uint32_t> idx(rows);
        for (size_t j = 0; j < rows; j++) {
            const std::vector<uint64_t> *acc = nodes[j]->node->lut.acc.get();
            size_t u = 0;
            while (u < uniq.size() && uniq[u] != acc) u++;
            if (u == uniq.size()) uniq.push_back(acc);
            idx[j] = (uint32_t)u;
        }
        if (in_cap_ < rows * big_) {
            in_buf_.reset();
            in_buf_ = pooled(rows * big_, &in_cap_);
        }
        auto out = std::make_shared<Block>();
        size_t out_cap = 0;
        out->own = pooled(rows * big_, &out_cap);
        out->data = out->own.get();
        if (tabs_cap_ < uniq.size() * lut_len_) {
            tabs_buf_.reset();
            tabs_buf_ = pooled(uniq.size() * lut_len_, &tabs_cap_);
        }
        uint64_t *tabs = tabs_buf_.get();
        uint64_t *in = in_buf_.get();
        parallel_rows(rows, 128, [&](size_t r0, size_t r1) {
            for (size_t j = r0; j < r1; j++) nodes[j]->node->input.materialize(in + j * big_, big_);
        });
        for (size_t u = 0; u < uniq.size(); u++) std::copy(uniq[u]->begin(), uniq[u]->end(), tabs + u * lut_len_);
        const auto t1 = std::chrono::steady_clock::now();
        if (tfhe_hip_ks_pbs_batch(sk_, in, tabs, uniq.size(), idx.data(), out->own.get(), rows))
            throw std::runtime_error(tfhe_hip_last_error());
        const auto t2 = std::chrono::steady_clock::now();
        for (size_t j = 0; j < rows; j++) {
            Block &b = *nodes[j];
            b.data = out->data + j * big_;
            b.storage = out;
            b.depth = 0;
            b.node.reset(); // releases the references to the inputs
        }
        pbs_count += rows;
        launches += 1;
        if (trace) {
            const auto t3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "[host] level: %zu PBS, %zu LUTs: pack %.2f ms, ks_pbs_batch %.2f ms, unpack %.2f ms\n", rows,
                         uniq.size(), ms(t0, t1), ms(t1, t2), ms(t2, t3));
        }
    }
    std::vector<std::shared_ptr<Block>> pending_;
    mutable std::mutex lut_mutex_;
    mutable std::map<std::vector<uint64_t>, LookupTable> lut_cache_;
    bool batching_ = false;
    bool strict_outputs_ = true;
    std::unordered_map<const uint64_t *, std::pair<uint64_t, uint64_t>> out_meta_; // output address -> (degree, noise level)
    std::vector<std::pair<Ciphertext, uint64_t *>> deferred_;

  private:
    void trivial_pbs_assign(Ciphertext &c, const LookupTable &lut) const
    {
        const uint64_t modulus_sup = (uint64_t)p_.message_modulus * p_.carry_modulus;
        const uint64_t box = p_.poly_size / modulus_sup;
        uint64_t v = c.body / delta_;
        const uint64_t *body = lut.acc->data() + (size_t)p_.glwe_dim * p_.poly_size;
        uint64_t r;
        if (v >= modulus_sup) r = (uint64_t)0 - body[(v % modulus_sup) * box];
        else r = body[v * box];
        c.body = r;
        c.degree = lut.degree;
    }
    tfhe_hip_server_key *sk_;
    tfhe_hip_params p_;
    std::shared_ptr<uint64_t> pooled(size_t words, size_t *cap)
    {
        std::shared_ptr<HostPool> pool = pool_;
        size_t c = 0;
        uint64_t *p = pool->acquire(words, &c);
        *cap = c;
        return std::shared_ptr<uint64_t>(p, [pool, c](uint64_t *q) { pool->release(q, c); });
    }
    std::shared_ptr<uint64_t> dpooled(size_t words, size_t *cap)
    {
        std::shared_ptr<DevicePool> pool = dpool_;
        size_t c = 0;
        uint64_t *p = pool->acquire(words, &c);
        *cap = c;
        return std::shared_ptr<uint64_t>(p, [pool, c](uint64_t *q) { pool->release(q, c); });
    }
    // the device slot of a lookup table (uploaded on first use; the arena is recycled wholesale when it is full --
    // earlier launches have their tables until then, and everything runs in stream order)
    size_t lut_capacity_ = 512; // TFHE_HIP_LUT_SLOTS overrides (tests exercise the recycling with a tiny arena)
    uint32_t lut_slot(const std::shared_ptr<const std::vector<uint64_t>> &acc, size_t *n_new)
    {
        auto hit = lut_slots_.find(acc.get());
        if (hit != lut_slots_.end()) return hit->second.first;
        if (!d_luts_) {
            size_t cap = 0;
            d_luts_ = dpooled(lut_capacity_ * lut_len_, &cap);
        }
        if (lut_slots_.size() >= lut_capacity_) throw std::logic_error("lookup-table arena overflow");
        const uint32_t slot = (uint32_t)lut_slots_.size();
        if (tfhe_hip_device_upload(sk_, d_luts_.get() + slot * lut_len_, acc->data(), lut_len_ * 8)) throw std::runtime_error(tfhe_hip_last_error());
        lut_slots_.emplace(acc.get(), std::make_pair(slot, acc)); // the entry keeps the table (and so its address) alive
        ++*n_new;
        return slot;
    }
    std::unordered_map<const std::vector<uint64_t> *, std::pair<uint32_t, std::shared_ptr<const std::vector<uint64_t>>>> lut_slots_;
    std::shared_ptr<uint64_t> d_luts_;
    bool resident_ = true;
    std::shared_ptr<DevicePool> dpool_;
    std::shared_ptr<uint64_t> d_in_; // the level's input rows
    std::vector<std::shared_ptr<uint64_t>> in_flight_; // page-locked upload sources the DMA engine may still be reading
    size_t d_in_cap_ = 0;
    std::shared_ptr<HostPool> pool_ = std::make_shared<HostPool>();
    std::shared_ptr<uint64_t> in_buf_, tabs_buf_; // upload staging, reused across levels
    size_t in_cap_ = 0, tabs_cap_ = 0;
    size_t big_, lut_len_;
    size_t index_blocks_ = 4;
    uint64_t delta_;
    size_t round_small_ = 0, round_large_ = 0;
};

// one level of PBS work under construction
struct Level {
    std::vector<Ciphertext *> cts;
    std::vector<const LookupTable *> luts;
    void add(Ciphertext &c, const LookupTable &l) { cts.push_back(&c); luts.push_back(&l); }
    void run(ServerKey &sk)
    {
        sk.apply_lookup_table_batch(cts, luts);
        cts.clear();
        luts.clear();
    }
};

// ---------------------------------------------------------------------------------------------
// integer layer
// ---------------------------------------------------------------------------------------------

// Tree reduction of many independent groups of boolean blocks, one launch per tree level.
// all == true : are_all_comparisons_block_true (sum of <= 15 blocks, LUT x == chunk_len)
// all == false: is_at_least_one_comparisons_block_true (LUT x != 0)
inline std::vector<Ciphertext> reduce_boolean_groups(ServerKey &sk, std::vector<std::vector<Ciphertext>> groups, bool all)
{
    const size_t max_value = (size_t)sk.message_modulus() * sk.carry_modulus() - 1;
    std::vector<LookupTable> eq_lut(max_value + 1);
    std::vector<bool> have(max_value + 1, false);
    const LookupTable not_zero = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x != 0); });
    for (auto &g : groups)
        if (g.empty()) g.push_back(sk.create_trivial(1)); // "if the vec is empty, a trivial 1 is returned"
    for (;;) {
        bool work = false;
        std::vector<std::vector<Ciphertext>> next(groups.size());
        std::vector<std::vector<size_t>> lens(groups.size());
        for (size_t gi = 0; gi < groups.size(); gi++) {
            auto &g = groups[gi];
            if (g.size() <= 1) { next[gi] = std::move(g); continue; }
            work = true;
            for (size_t b = 0; b < g.size(); b += max_value) {
                // "Since all blocks encrypt either 0 or 1, we can sum max_value of them"
                const size_t len = std::min(max_value, g.size() - b);
                Ciphertext sum = g[b];
                for (size_t j = 1; j < len; j++) sk.unchecked_add_assign(sum, g[b + j]);
                next[gi].push_back(std::move(sum));
                lens[gi].push_back(len);
                if (all && !have[len]) {
                    eq_lut[len] = sk.generate_lookup_table([len](uint64_t x) { return (uint64_t)(x == len); });
                    have[len] = true;
                }
            }
        }
        if (work) {
            Level lvl; // every chunk of every group of this tree level: one launch
            for (size_t gi = 0; gi < groups.size(); gi++)
                for (size_t c = 0; c < lens[gi].size(); c++) lvl.add(next[gi][c], all ? eq_lut[lens[gi][c]] : not_zero);
            lvl.run(sk);
        }
        groups = std::move(next);
        if (!work) break;
    }
    std::vector<Ciphertext> out;
    out.reserve(groups.size());
    for (auto &g : groups) out.push_back(std::move(g[0]));
    return out;
}

inline Ciphertext are_all_comparisons_block_true(ServerKey &sk, std::vector<Ciphertext> blocks)
{
    std::vector<std::vector<Ciphertext>> g(1);
    g[0] = std::move(blocks);
    return reduce_boolean_groups(sk, std::move(g), true)[0];
}
inline Ciphertext is_at_least_one_comparisons_block_true(ServerKey &sk, std::vector<Ciphertext> blocks)
{
    std::vector<std::vector<Ciphertext>> g(1);
    g[0] = std::move(blocks);
    return reduce_boolean_groups(sk, std::move(g), false)[0];
}

// pack_block_chunk (scalar_comparison.rs:104-138): high * message_modulus + low
inline Ciphertext pack_block_pair(const ServerKey &sk, const RadixCiphertext &ct, size_t i)
{
    if (i + 1 >= ct.size()) return ct[i];
    Ciphertext p = ct[i + 1];
    sk.unchecked_scalar_mul_assign(p, sk.message_modulus());
    sk.unchecked_add_assign(p, ct[i]);
    return p;
}

// Equality of two radix ciphertexts, staged: the reference compares block by block with a bivariate LUT
// (comparison.rs:18-28: one PBS per block).  Here the blocks are packed two by two (pack_block_chunk) and the
// packed values subtracted (the "true" LWE subtraction of comparator.rs:193-224, which uses the padding bit:
// the difference lies in [-15, 15]); the LUT x == 0 then gives 1 exactly for equal pairs -- a negative
// difference reads the negacyclic image -f(16 + d) = -0.  One PBS per PAIR of blocks.
inline LookupTable equality_lut(const ServerKey &sk)
{
    return sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 0); });
}
inline void stage_eq_blocks(ServerKey &sk, const RadixCiphertext &lhs, const RadixCiphertext &rhs,
                            std::vector<Ciphertext> &out)
{
    if (lhs.size() != rhs.size()) throw std::runtime_error("eq: radix ciphertexts have different block counts");
    for (size_t b = 0; b < lhs.size(); b += 2) {
        Ciphertext d = pack_block_pair(sk, lhs, b);
        sk.lwe_sub_assign(d, pack_block_pair(sk, rhs, b)); // the PBS is staged by the caller with equality_lut
        out.push_back(std::move(d));
    }
}

// unchecked_eq_parallelized: 1 level of bivariate PBS + tree
inline Ciphertext eq_parallelized(ServerKey &sk, const RadixCiphertext &lhs, const RadixCiphertext &rhs)
{
    const LookupTable lut = equality_lut(sk);
    std::vector<Ciphertext> cmp;
    Level lvl;
    stage_eq_blocks(sk, lhs, rhs, cmp);
    for (auto &c : cmp) lvl.add(c, lut);
    lvl.run(sk);
    return are_all_comparisons_block_true(sk, std::move(cmp));
}


// ---- comparisons (integer/server_key/comparator.rs) ----------------------------------------------
constexpr uint64_t IS_INFERIOR = 0, IS_EQUAL = 1, IS_SUPERIOR = 2;

struct Comparator {
    ServerKey &sk;
    LookupTable sign_lut;
    explicit Comparator(ServerKey &k) : sk(k)
    {
        sign_lut = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x != 0); }); // comparator.rs:60
    }
    // compare_block_assign on packed pairs (comparator.rs:193-224, 422-443): the difference of two packed
    // blocks uses the padding bit; after the sign LUT the block holds the CENTERED sign -1, 0 or 1 (the
    // reference adds 1 at once; here that happens after the reduction).  The blocks are appended to `out`
    // (least significant first), the PBS is staged by the caller.
    void stage(const RadixCiphertext &lhs, const RadixCiphertext &rhs, std::vector<Ciphertext> &out) const
    {
        if (lhs.size() != rhs.size()) throw std::runtime_error("compare: radix ciphertexts have different block counts");
        for (size_t i = 0; i < lhs.size(); i += 2) {
            Ciphertext d = pack_block_pair(sk, lhs, i);
            sk.lwe_sub_assign(d, pack_block_pair(sk, rhs, i));
            out.push_back(std::move(d));
        }
    }
    // scalar_compare_block_assign (comparator.rs:226-238); scalar_blocks in base message_modulus^2
    void stage_scalar(const RadixCiphertext &lhs, uint64_t scalar, std::vector<Ciphertext> &out) const
    {
        const uint64_t packed_mod = sk.message_modulus() * sk.message_modulus();
        for (size_t i = 0; i < lhs.size(); i += 2) {
            Ciphertext d = pack_block_pair(sk, lhs, i);
            sk.lwe_plaintext_sub_assign(d, scalar % packed_mod);
            scalar /= packed_mod;
            out.push_back(std::move(d));
        }
    }
    void run_signs(std::vector<std::vector<Ciphertext>> &groups) const
    {
        Level lvl;
        for (auto &g : groups)
            for (auto &c : g) lvl.add(c, sign_lut);
        lvl.run(sk);
    }
    // Reduction of centered sign blocks to one sign per group, many independent comparisons at once, one
    // launch per tree level.  The reference combines TWO signs per PBS (4 * msb + lsb through a selection LUT,
    // reduce_signs_parallelized, comparator.rs:240-283).  With centered signs the most significant non-zero
    // sign of THREE blocks is the sign of 9 a + 3 b + c (balanced ternary, |x| <= 13, inside the padded 5-bit
    // space), i.e. the same `x != 0` LUT through the padding bit: a ternary tree, log3 instead of log2 levels.
    // Returns the signs in the reference's encoding (IS_INFERIOR / IS_EQUAL / IS_SUPERIOR = centered + 1).
    std::vector<Ciphertext> reduce(std::vector<std::vector<Ciphertext>> groups) const
    {
        for (auto &g : groups)
            if (g.empty()) g.push_back(sk.create_trivial(0)); // equal
        for (;;) {
            std::vector<std::vector<Ciphertext>> next(groups.size());
            for (size_t gi = 0; gi < groups.size(); gi++) {
                auto &g = groups[gi];
                if (g.size() == 1) { next[gi].push_back(std::move(g[0])); continue; }
                for (size_t i = 0; i < g.size(); i += 3) {
                    const size_t len = std::min<size_t>(3, g.size() - i);
                    Ciphertext x = g[i + len - 1]; // most significant of the chunk
                    for (size_t t = len - 1; t-- > 0;) {
                        sk.unchecked_scalar_mul_assign(x, 3);
                        sk.unchecked_add_assign(x, g[i + t]);
                    }
                    next[gi].push_back(std::move(x)); // a last chunk of one block passes through unchanged
                }
            }
            Level lvl;
            for (size_t gi = 0; gi < groups.size(); gi++)
                for (size_t c = 0; c < next[gi].size(); c++)
                    if (groups[gi].size() > 1 && (c + 1 < next[gi].size() || groups[gi].size() % 3 != 1)) lvl.add(next[gi][c], sign_lut);
            const bool work = !lvl.cts.empty();
            lvl.run(sk);
            groups = std::move(next);
            if (!work) break;
        }
        std::vector<Ciphertext> out;
        for (auto &g : groups) {
            sk.unchecked_scalar_add_assign(g[0], 1);
            g[0].degree = 2;
            out.push_back(std::move(g[0]));
        }
        return out;
    }
    Ciphertext trivial_sign(uint64_t v) const { return sk.create_trivial(v); }
    // map_sign_result (comparator.rs:957-972): one PBS per comparison, all in one launch
    std::vector<Ciphertext> map_signs(std::vector<Ciphertext> signs, const std::vector<std::function<bool(uint64_t)>> &fn) const
    {
        std::vector<LookupTable> luts(signs.size());
        Level lvl;
        for (size_t i = 0; i < signs.size(); i++) {
            const auto f = fn[fn.size() == 1 ? 0 : i];
            luts[i] = sk.generate_lookup_table([f](uint64_t x) { return (uint64_t)f(x); });
        }
        for (size_t i = 0; i < signs.size(); i++) lvl.add(signs[i], luts[i]);
        lvl.run(sk);
        return signs;
    }
    // unchecked_compare_parallelized: the sign block of lhs vs rhs
    Ciphertext compare(const RadixCiphertext &lhs, const RadixCiphertext &rhs) const
    {
        std::vector<std::vector<Ciphertext>> g(1);
        stage(lhs, rhs, g[0]);
        run_signs(g);
        return reduce(std::move(g))[0];
    }
    // unsigned_unchecked_scalar_compare_blocks_parallelized (comparator.rs:677-783)
    Ciphertext scalar_compare(const RadixCiphertext &lhs, uint64_t scalar) const
    {
        const uint64_t bits = 2 * lhs.size(); // message_modulus 4: 2 bits per block
        if (bits < 64 && (scalar >> bits) != 0) return trivial_sign(IS_INFERIOR); // scalar does not fit: lhs < scalar
        std::vector<std::vector<Ciphertext>> g(1);
        stage_scalar(lhs, scalar, g[0]);
        run_signs(g);
        return reduce(std::move(g))[0];
    }
};

enum class Cmp { GT, GE, LT, LE };
inline std::function<bool(uint64_t)> sign_predicate(Cmp op)
{
    switch (op) {
    case Cmp::GT: return [](uint64_t x) { return x == IS_SUPERIOR; };
    case Cmp::GE: return [](uint64_t x) { return x == IS_SUPERIOR || x == IS_EQUAL; };
    case Cmp::LT: return [](uint64_t x) { return x == IS_INFERIOR; };
    default: return [](uint64_t x) { return x == IS_INFERIOR || x == IS_EQUAL; };
    }
}
// gt / ge / lt / le _parallelized on clean inputs (comparator.rs:1079-1130, 1267-1330)
inline Ciphertext compare_parallelized(ServerKey &sk, const RadixCiphertext &lhs, const RadixCiphertext &rhs, Cmp op)
{
    const Comparator c(sk);
    return c.map_signs({c.compare(lhs, rhs)}, {sign_predicate(op)})[0];
}
inline Ciphertext scalar_compare_parallelized(ServerKey &sk, const RadixCiphertext &lhs, uint64_t scalar, Cmp op)
{
    const Comparator c(sk);
    return c.map_signs({c.scalar_compare(lhs, scalar)}, {sign_predicate(op)})[0];
}

// unchecked_programmable_if_then_else_parallelized (radix_parallel/cmux.rs:194-252, zero_out_if :281-315):
// one launch zeroes the unselected side of every block pair; the sums are leveled
inline RadixCiphertext programmable_if_then_else(ServerKey &sk, const Ciphertext &condition, const RadixCiphertext &true_ct,
                                                 const RadixCiphertext &false_ct, const std::function<bool(uint64_t)> &predicate)
{
    if (true_ct.size() != false_ct.size()) throw std::runtime_error("if_then_else: radix ciphertexts have different block counts");
    const LookupTable keep_if = sk.generate_lookup_table_bivariate(
        [&](uint64_t block, uint64_t cond) { return predicate(cond) ? block : (uint64_t)0; });
    const LookupTable keep_unless = sk.generate_lookup_table_bivariate(
        [&](uint64_t block, uint64_t cond) { return predicate(cond) ? (uint64_t)0 : block; });
    RadixCiphertext t = true_ct, f = false_ct;
    Level lvl;
    for (size_t i = 0; i < t.size(); i++) {
        sk.bivariate_pack_assign(t[i], condition);
        sk.bivariate_pack_assign(f[i], condition);
        lvl.add(t[i], keep_if);
        lvl.add(f[i], keep_unless);
    }
    lvl.run(sk);
    // one of the two sides is zero: the sum is the selected block, no carry; it is handed on as it is (two units of
    // output noise) instead of going through the reference's second, cleaning bootstrap
    for (size_t i = 0; i < t.size(); i++) {
        sk.unchecked_add_assign(t[i], f[i]);
        t[i].degree = sk.message_modulus() - 1;
    }
    return t;
}
// if_then_else_parallelized / cmux_parallelized (cmux.rs:72-145)
inline RadixCiphertext if_then_else_parallelized(ServerKey &sk, const Ciphertext &condition, const RadixCiphertext &true_ct,
                                                 const RadixCiphertext &false_ct)
{
    return programmable_if_then_else(sk, condition, true_ct, false_ct, [](uint64_t c) { return c == 1; });
}
// unchecked_min_or_max_parallelized (comparator.rs:849-875)
inline RadixCiphertext min_or_max_parallelized(ServerKey &sk, const RadixCiphertext &lhs, const RadixCiphertext &rhs, bool want_max)
{
    const Comparator c(sk);
    const Ciphertext sign = c.compare(lhs, rhs);
    if (want_max) return programmable_if_then_else(sk, sign, lhs, rhs, [](uint64_t s) { return s == IS_SUPERIOR; });
    return programmable_if_then_else(sk, sign, lhs, rhs, [](uint64_t s) { return s == IS_INFERIOR; });
}

// ---- radix multiplication (mul.rs:329-414) ----------------------------------------------------

inline RadixCiphertext blockshift(const ServerKey &sk, const RadixCiphertext &ct, size_t shift)
{
    RadixCiphertext r(ct.size());
    for (size_t j = 0; j < ct.size(); j++) r[j] = (j >= shift) ? ct[j - shift] : sk.create_trivial(0);
    return r;
}

// add.rs:518-603, 724-771: carry propagation with the Hillis-Steele prefix sum (the reference picks
// this or a sequential variant from rayon::current_num_threads(); the build fixes Hillis-Steele)
inline void propagate_single_carry_parallelized_low_latency(ServerKey &sk, RadixCiphertext &ct)
{
    const uint64_t modulus = sk.message_modulus();
    const size_t nb = ct.size();
    // generate_init_carry_array (add.rs:724-771) with CENTERED states: generate +1, propagate 0, kill -1
    // (the LUT gives 2 / 1 / 0, the plaintext 1 is subtracted: same trick as the comparator's signs)
    const LookupTable lut_state = sk.generate_lookup_table(
        [modulus](uint64_t x) { return (uint64_t)(x >= modulus ? 2 : (x == modulus - 1 ? 1 : 0)); });
    const LookupTable sign_lut = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x != 0); });
    std::vector<Ciphertext> gp(ct.begin(), ct.end());
    Level lvl;
    for (size_t i = 0; i < nb; i++) lvl.add(gp[i], lut_state);
    lvl.run(sk);
    for (size_t i = 0; i < nb; i++) {
        sk.lwe_plaintext_sub_assign(gp[i], 1);
        gp[i].degree = 1;
    }
    // compute_prefix_sum_hillis_steele (add.rs:518-603) with prefix_sum_carry_propagation(msb, lsb) =
    // msb == propagate ? lsb : msb.  On centered states that is "the most significant non-zero one", and for
    // THREE states the sign of 9 a + 3 b + c (|x| <= 13, inside the padded 5-bit space; see Comparator::reduce):
    // the prefix scan triples its span per launch instead of doubling it.
    for (size_t stride = 1; stride < nb; stride *= 3) {
        std::vector<Ciphertext> step(gp.begin(), gp.end());
        for (size_t i = stride; i < nb; i++) {
            sk.unchecked_scalar_mul_assign(step[i], 3);
            sk.unchecked_add_assign(step[i], gp[i - stride]);
            if (i >= 2 * stride) {
                sk.unchecked_scalar_mul_assign(step[i], 3);
                sk.unchecked_add_assign(step[i], gp[i - 2 * stride]);
            }
            lvl.add(step[i], sign_lut);
        }
        lvl.run(sk);
        for (size_t i = stride; i < nb; i++) gp[i] = std::move(step[i]);
    }
    // block 0 cannot propagate, so every prefix state is +-1 and the carry is (state + 1) / 2:
    // x = 2 * block + state (in [-1, 15]) -> ((x + 1) / 2) mod modulus is (block + carry) mod modulus for both
    // parities of the state; x = -1 reads -f(15) = -0 through the padding bit.
    const LookupTable msg_carry = sk.generate_lookup_table([modulus](uint64_t x) { return ((x + 1) / 2) % modulus; });
    const LookupTable msg = sk.generate_lookup_table([modulus](uint64_t x) { return x % modulus; });
    for (size_t i = 0; i < nb; i++) {
        if (i > 0) {
            sk.unchecked_scalar_mul_assign(ct[i], 2);
            sk.unchecked_add_assign(ct[i], gp[i - 1]);
            lvl.add(ct[i], msg_carry);
        } else {
            lvl.add(ct[i], msg);
        }
    }
    lvl.run(sk);
}

// add.rs:789-956
inline RadixCiphertext unchecked_sum_ciphertexts_vec_parallelized(ServerKey &sk, std::vector<RadixCiphertext> cts)
{
    if (cts.empty()) throw std::runtime_error("sum of zero ciphertexts");
    const size_t nb = cts[0].size();
    const uint64_t mm = sk.message_modulus();
    const size_t fill = (size_t)((mm * sk.carry_modulus() - 1) / (mm - 1)); // 5
    const LookupTable msg = sk.generate_lookup_table([mm](uint64_t x) { return x % mm; });
    const LookupTable carry = sk.generate_lookup_table([mm](uint64_t x) { return x / mm; });
    auto add_radix = [&](RadixCiphertext &a, const RadixCiphertext &b) {
        for (size_t j = 0; j < nb; j++) sk.unchecked_add_assign(a[j], b[j]);
    };
    if (cts.size() == 1) return cts[0];
    if (cts.size() > 2) {
        while (cts.size() > fill) {
            const size_t chunks = cts.size() / fill, rem = cts.size() % fill;
            std::vector<RadixCiphertext> out(2 * chunks);
            Level lvl;
            for (size_t c = 0; c < chunks; c++) {
                RadixCiphertext &s = cts[c * fill];
                size_t first = nb - 1, last = 0;
                for (size_t t = 1; t < fill; t++) {
                    const RadixCiphertext &a = cts[c * fill + t];
                    size_t f = nb, l = nb - 1;
                    for (size_t j = 0; j < nb; j++)
                        if (a[j].degree != 0) { f = j; break; }
                    for (size_t j = nb; j-- > 0;)
                        if (a[j].degree != 0) { l = j; break; }
                    first = std::min(first, f);
                    last = std::max(last, l);
                    for (size_t j = f; j <= l && j < nb; j++) sk.unchecked_add_assign(s[j], a[j]);
                }
                RadixCiphertext cy = s;
                // "This carry would be thrown away, so don't compute it"
                const long end = (last == nb - 1) ? (long)last - 1 : (long)last;
                for (size_t j = first; j <= last; j++) lvl.add(s[j], msg);
                for (size_t j = 0; j < nb; j++) {
                    if (j >= first && (long)j <= end) lvl.add(cy[j], carry);
                    else cy[j] = sk.create_trivial(0);
                }
                out[2 * c] = std::move(s);
                out[2 * c + 1] = std::move(cy);
            }
            lvl.run(sk);
            std::vector<RadixCiphertext> next;
            for (size_t c = 0; c < chunks; c++) {
                next.push_back(std::move(out[2 * c]));
                RadixCiphertext &cy = out[2 * c + 1];
                RadixCiphertext rot(nb); // carry_ct.blocks_mut().rotate_right(1)
                for (size_t j = 0; j < nb; j++) rot[(j + 1) % nb] = std::move(cy[j]);
                next.push_back(std::move(rot));
            }
            for (size_t r = 0; r < rem; r++) next.push_back(std::move(cts[chunks * fill + r]));
            cts = std::move(next);
        }
        RadixCiphertext result = cts[0];
        for (size_t t = 1; t < cts.size(); t++) add_radix(result, cts[t]);
        RadixCiphertext m = result, cy(nb);
        Level lvl;
        for (size_t j = 0; j < nb; j++) lvl.add(m[j], msg);
        cy[0] = sk.create_trivial(0);
        for (size_t j = 0; j + 1 < nb; j++) {
            cy[j + 1] = result[j];
            lvl.add(cy[j + 1], carry);
        }
        lvl.run(sk);
        add_radix(m, cy);
        propagate_single_carry_parallelized_low_latency(sk, m);
        return m;
    }
    RadixCiphertext r = cts[0];
    add_radix(r, cts[1]);
    propagate_single_carry_parallelized_low_latency(sk, r);
    return r;
}

// unchecked_mul_assign_parallelized for clean inputs (== mul_parallelized on fresh ciphertexts)
inline RadixCiphertext mul_parallelized(ServerKey &sk, const RadixCiphertext &lhs, const RadixCiphertext &rhs)
{
    if (lhs.size() != rhs.size()) throw std::runtime_error("mul: radix ciphertexts have different block counts");
    const uint64_t mm = sk.message_modulus();
    const size_t nb = lhs.size();
    const LookupTable lsb = sk.generate_lookup_table_bivariate([mm](uint64_t x, uint64_t y) { return (x * y) % mm; });
    const LookupTable msb = sk.generate_lookup_table_bivariate([mm](uint64_t x, uint64_t y) { return (x * y) / mm; });
    std::vector<RadixCiphertext> terms;
    terms.reserve(2 * nb);
    for (size_t i = 0; i < nb; i++)
        if (rhs[i].degree != 0) terms.push_back(blockshift(sk, lhs, i));
    const size_t n_lsb = terms.size();
    if (mm > 2)
        for (size_t i = 0; i < nb; i++)
            if (rhs[i].degree != 0) terms.push_back(blockshift(sk, lhs, i + 1));
    Level lvl;
    size_t t = 0;
    for (size_t i = 0; i < nb; i++) {
        if (rhs[i].degree == 0) continue;
        for (size_t j = i; j < nb; j++)
            if (terms[t][j].degree != 0) {
                sk.bivariate_pack_assign(terms[t][j], rhs[i]);
                lvl.add(terms[t][j], lsb);
            }
        t++;
    }
    if (mm > 2) {
        t = n_lsb;
        for (size_t i = 0; i < nb; i++) {
            if (rhs[i].degree == 0) continue;
            for (size_t j = i + 1; j < nb; j++)
                if (terms[t][j].degree != 0) {
                    sk.bivariate_pack_assign(terms[t][j], rhs[i]);
                    lvl.add(terms[t][j], msb);
                }
            t++;
        }
    }
    lvl.run(sk);
    if (terms.empty()) {
        RadixCiphertext z(nb);
        for (auto &b : z) b = sk.create_trivial(0);
        return z;
    }
    return unchecked_sum_ciphertexts_vec_parallelized(sk, std::move(terms));
}


// ---- addition / subtraction on clean inputs -------------------------------------------------------
// add_parallelized (radix_parallel/add.rs:110-170): block-wise add + carry propagation
inline RadixCiphertext add_parallelized(ServerKey &sk, const RadixCiphertext &lhs, const RadixCiphertext &rhs)
{
    if (lhs.size() != rhs.size()) throw std::runtime_error("add: radix ciphertexts have different block counts");
    RadixCiphertext r = lhs;
    for (size_t i = 0; i < r.size(); i++) sk.unchecked_add_assign(r[i], rhs[i]);
    propagate_single_carry_parallelized_low_latency(sk, r);
    return r;
}
// scalar_add_parallelized (radix_parallel/scalar_add.rs): the scalar's digits go into the bodies
inline RadixCiphertext scalar_add_parallelized(ServerKey &sk, const RadixCiphertext &lhs, uint64_t scalar)
{
    const uint64_t mm = sk.message_modulus();
    RadixCiphertext r = lhs;
    for (size_t i = 0; i < r.size(); i++) {
        sk.unchecked_scalar_add_assign(r[i], scalar % mm);
        scalar /= mm;
    }
    propagate_single_carry_parallelized_low_latency(sk, r);
    return r;
}
// unchecked_neg (radix/neg.rs:50-95): block_i <- z_i - (block_i + z_{i-1} / message_modulus)
inline RadixCiphertext unchecked_neg(const ServerKey &sk, const RadixCiphertext &ct)
{
    const uint64_t mm = sk.message_modulus();
    RadixCiphertext r = ct;
    uint64_t z_b = 0;
    for (size_t i = 0; i < r.size(); i++) {
        Ciphertext &b = r[i];
        if (z_b != 0) sk.unchecked_scalar_add_assign(b, z_b);
        // unchecked_neg_assign_with_correcting_term (shortint/server_key/neg.rs): z = ceil(degree / mm) * mm
        const uint64_t z = (b.degree + mm - 1) / mm * mm;
        const uint64_t noise = b.noise_level;
        sk.unchecked_scalar_mul_assign(b, (uint64_t)0 - 1);
        b.noise_level = noise;
        b.degree = 0;
        sk.unchecked_scalar_add_assign(b, z);
        b.degree = z;
        z_b = z / mm;
    }
    return r;
}
// sub_parallelized (radix_parallel/sub.rs:162-243): lhs + (-rhs), then carry propagation
inline RadixCiphertext sub_parallelized(ServerKey &sk, const RadixCiphertext &lhs, const RadixCiphertext &rhs)
{
    if (lhs.size() != rhs.size()) throw std::runtime_error("sub: radix ciphertexts have different block counts");
    RadixCiphertext r = lhs;
    const RadixCiphertext n = unchecked_neg(sk, rhs);
    for (size_t i = 0; i < r.size(); i++) sk.unchecked_add_assign(r[i], n[i]);
    propagate_single_carry_parallelized_low_latency(sk, r);
    return r;
}
// neg_parallelized (radix_parallel/neg.rs)
inline RadixCiphertext neg_parallelized(ServerKey &sk, const RadixCiphertext &ct)
{
    RadixCiphertext r = unchecked_neg(sk, ct);
    propagate_single_carry_parallelized_low_latency(sk, r);
    return r;
}
// scalar_mul_parallelized (radix_parallel/scalar_mul.rs) on clean inputs: one launch forms, for every
// non-trivial base-4 digit d of the scalar, the message and carry parts of d * block; the shifted terms
// go through the same sum as the radix multiplication
inline RadixCiphertext scalar_mul_parallelized(ServerKey &sk, const RadixCiphertext &lhs, uint64_t scalar)
{
    const uint64_t mm = sk.message_modulus();
    const size_t nb = lhs.size();
    std::vector<LookupTable> lo(mm), hi(mm);
    for (uint64_t d = 2; d < mm; d++) {
        lo[d] = sk.generate_lookup_table([mm, d](uint64_t x) { return (x % mm) * d % mm; });
        hi[d] = sk.generate_lookup_table([mm, d](uint64_t x) { return (x % mm) * d / mm; });
    }
    std::vector<RadixCiphertext> terms;
    Level lvl;
    std::vector<std::pair<size_t, uint64_t>> staged; // (term index, digit) of terms that need the LUTs
    for (size_t j = 0; j < nb && scalar != 0; j++, scalar /= mm) {
        const uint64_t d = scalar % mm;
        if (d == 0) continue;
        terms.push_back(blockshift(sk, lhs, j));
        if (d >= 2) {
            staged.push_back({terms.size() - 1, d});
            terms.push_back(blockshift(sk, lhs, j + 1));
            staged.push_back({terms.size() - 1, d | 0x100});
        }
    }
    for (auto &st : staged) {
        const bool is_hi = st.second & 0x100;
        const uint64_t d = st.second & 0xff;
        for (auto &b : terms[st.first])
            if (b.degree != 0) lvl.add(b, is_hi ? hi[d] : lo[d]);
    }
    lvl.run(sk);
    if (terms.empty()) {
        RadixCiphertext z(nb);
        for (auto &b : z) b = sk.create_trivial(0);
        return z;
    }
    return unchecked_sum_ciphertexts_vec_parallelized(sk, std::move(terms));
}

// ---------------------------------------------------------------------------------------------
// string layer (unpadded strings with clear lengths)
// ---------------------------------------------------------------------------------------------

// StringServerKey::eq -> eq_encrypted -> eq_no_init_padding (comparisons.rs:101-122, 184-212)
inline Ciphertext string_eq(ServerKey &sk, const FheString &s1, const FheString &s2)
{
    if (s1.size() != s2.size()) return sk.create_trivial(0); // clear length mismatch -> create_false
    const LookupTable lut = equality_lut(sk);
    std::vector<Ciphertext> cmp;
    Level lvl;
    for (size_t c = 0; c < s1.size(); c++) stage_eq_blocks(sk, s1[c], s2[c], cmp);
    for (auto &c : cmp) lvl.add(c, lut);
    lvl.run(sk);
    return are_all_comparisons_block_true(sk, std::move(cmp)); // empty strings: trivial true
}

// per-window matches of `needle` in `hay` for windows [w_begin, w_end)
// (contains_unpadded_string / is_prefix_of_slice with Padding::None, contains.rs:77-91, pattern.rs:263-274)
inline std::vector<Ciphertext> string_contains_windows(ServerKey &sk, const FheString &hay, const FheString &needle,
                                                       size_t w_begin, size_t w_end)
{
    const LookupTable lut = equality_lut(sk);
    const size_t nw = w_end - w_begin;
    std::vector<std::vector<Ciphertext>> groups(nw);
    Level lvl;
    for (size_t w = 0; w < nw; w++)
        for (size_t j = 0; j < needle.size(); j++) stage_eq_blocks(sk, hay[w_begin + w + j], needle[j], groups[w]);
    for (auto &g : groups)
        for (auto &c : g) lvl.add(c, lut);
    lvl.run(sk);
    return reduce_boolean_groups(sk, std::move(groups), true);
}

// ---- clear patterns: compare_clear_char = scalar_eq_parallelized(char, byte) ------------------------
// (comparisons.rs:524-560 -> integer scalar_comparison.rs:366-460): the 4 blocks of a char are packed
// two by two (pack_block_chunk, :104-138: high * 4 + low) and each packed block goes through the LUT
// x == <clear nibble>.  (For bytes < 16 the reference compares the high blocks with zero through
// compare_blocks_with_zero; here the x == 0 LUT is used on the packed block: same decrypted value.)
struct ScalarEqLuts {
    std::vector<LookupTable> lut;
    std::vector<bool> have;
    ScalarEqLuts() : lut(16), have(16, false) {}
    const LookupTable &get(ServerKey &sk, unsigned k)
    {
        if (!have[k]) {
            lut[k] = sk.generate_lookup_table([k](uint64_t x) { return (uint64_t)(x == k); });
            have[k] = true;
        }
        return lut[k];
    }
};

inline void stage_scalar_eq_char(ServerKey &sk, const RadixCiphertext &ch, uint8_t byte, std::vector<Ciphertext> &out,
                                 std::vector<unsigned> &nibbles)
{
    if (ch.size() != 4) throw std::runtime_error("a char is a 4-block radix ciphertext");
    for (int half = 0; half < 2; half++) {
        Ciphertext p = ch[2 * half + 1];
        sk.unchecked_scalar_mul_assign(p, sk.message_modulus());
        sk.unchecked_add_assign(p, ch[2 * half]);
        out.push_back(std::move(p));
        nibbles.push_back((byte >> (4 * half)) & 15u);
    }
}

// groups[g] = block comparisons of (s[first_g + j] == pattern[j]) for all j; one launch
inline std::vector<Ciphertext> match_clear_at(ServerKey &sk, const FheString &s, const std::string &pattern,
                                              const std::vector<size_t> &starts)
{
    // "nibble h of char i equals v" is bootstrapped ONCE, however many windows ask for it: overlapping windows look at
    // the same char, and the high nibbles of text repeat (every lowercase letter is 6 or 7), so a pattern like "needle"
    // costs P + 1 flags per position instead of 2 P
    ScalarEqLuts luts;
    std::map<std::tuple<size_t, int, unsigned>, Ciphertext> flags;
    std::vector<std::vector<Ciphertext>> groups(starts.size());
    for (size_t g = 0; g < starts.size(); g++)
        for (size_t j = 0; j < pattern.size(); j++) {
            const size_t i = starts[g] + j;
            const RadixCiphertext &ch = s[i];
            if (ch.size() != 4) throw std::runtime_error("a char is a 4-block radix ciphertext");
            for (int half = 0; half < 2; half++) {
                const unsigned v = ((uint8_t)pattern[j] >> (4 * half)) & 15u;
                auto key = std::make_tuple(i, half, v);
                auto hit = flags.find(key);
                if (hit == flags.end()) {
                    Ciphertext p = pack_block_pair(sk, ch, 2 * (size_t)half);
                    sk.apply_lookup_table_batch({&p}, {&luts.get(sk, v)});
                    hit = flags.emplace(key, std::move(p)).first;
                }
                groups[g].push_back(hit->second);
            }
        }
    return reduce_boolean_groups(sk, std::move(groups), true);
}
inline std::vector<Ciphertext> match_encrypted_at(ServerKey &sk, const FheString &s, const FheString &pattern,
                                                  const std::vector<size_t> &starts)
{
    const LookupTable lut = equality_lut(sk);
    std::vector<std::vector<Ciphertext>> groups(starts.size());
    for (size_t g = 0; g < starts.size(); g++)
        for (size_t j = 0; j < pattern.size(); j++) stage_eq_blocks(sk, s[starts[g] + j], pattern[j], groups[g]);
    Level lvl;
    for (auto &g : groups)
        for (auto &c : g) lvl.add(c, lut);
    lvl.run(sk);
    return reduce_boolean_groups(sk, std::move(groups), true);
}

// StringServerKey::eq_clear (comparisons.rs:156-168), unpadded string with clear length
inline Ciphertext string_eq_clear(ServerKey &sk, const FheString &s, const std::string &clear)
{
    if (s.size() != clear.size()) return sk.create_trivial(0);
    if (s.empty()) return sk.create_trivial(1);
    return match_clear_at(sk, s, clear, {0})[0];
}
// StringServerKey::starts_with / ends_with (starts_with.rs:8, ends_with.rs:18), unpadded strings
inline Ciphertext string_starts_with(ServerKey &sk, const FheString &s, const FheString &prefix)
{
    if (prefix.size() > s.size()) return sk.create_trivial(0);
    if (prefix.empty()) return sk.create_trivial(1);
    return match_encrypted_at(sk, s, prefix, {0})[0];
}
inline Ciphertext string_ends_with(ServerKey &sk, const FheString &s, const FheString &suffix)
{
    if (suffix.size() > s.size()) return sk.create_trivial(0);
    if (suffix.empty()) return sk.create_trivial(1);
    return match_encrypted_at(sk, s, suffix, {s.size() - suffix.size()})[0];
}
inline Ciphertext string_starts_with_clear(ServerKey &sk, const FheString &s, const std::string &prefix)
{
    if (prefix.size() > s.size()) return sk.create_trivial(0);
    if (prefix.empty()) return sk.create_trivial(1);
    return match_clear_at(sk, s, prefix, {0})[0];
}
inline Ciphertext string_ends_with_clear(ServerKey &sk, const FheString &s, const std::string &suffix)
{
    if (suffix.size() > s.size()) return sk.create_trivial(0);
    if (suffix.empty()) return sk.create_trivial(1);
    return match_clear_at(sk, s, suffix, {s.size() - suffix.size()})[0];
}
// StringServerKey::contains with a clear pattern (contains_clear_string, contains.rs:42-74)
inline Ciphertext string_contains_clear(ServerKey &sk, const FheString &s, const std::string &pattern)
{
    if (pattern.empty()) return sk.create_trivial(1);
    if (pattern.size() > s.size()) return sk.create_trivial(0);
    std::vector<size_t> starts(s.size() - pattern.size() + 1);
    for (size_t w = 0; w < starts.size(); w++) starts[w] = w;
    return is_at_least_one_comparisons_block_true(sk, match_clear_at(sk, s, pattern, starts));
}
// boolean_bitnot = scalar_bitxor(block, 1) (integer/server_key/radix/bitwise_op.rs:732-738): one PBS
inline Ciphertext boolean_bitnot(ServerKey &sk, Ciphertext b)
{
    const uint64_t mm = sk.message_modulus();
    const LookupTable lut = sk.generate_lookup_table([mm](uint64_t x) { return ((x % mm) ^ 1) % mm; });
    Level lvl;
    lvl.add(b, lut);
    lvl.run(sk);
    return b;
}
// StringServerKey::ne
inline Ciphertext string_ne(ServerKey &sk, const FheString &s1, const FheString &s2);

inline size_t string_contains_window_count(size_t hay_len, size_t needle_len)
{
    return needle_len > hay_len ? 0 : hay_len - needle_len + 1;
}

// StringServerKey::contains
inline Ciphertext string_contains(ServerKey &sk, const FheString &hay, const FheString &needle)
{
    const size_t nw = string_contains_window_count(hay.size(), needle.size());
    if (nw == 0) return sk.create_trivial(0);
    if (needle.empty()) return sk.create_trivial(1);
    return is_at_least_one_comparisons_block_true(sk, string_contains_windows(sk, hay, needle, 0, nw));
}

inline Ciphertext string_ne(ServerKey &sk, const FheString &s1, const FheString &s2)
{
    return boolean_bitnot(sk, string_eq(sk, s1, s2));
}

// ---- find / rfind (find.rs:25-46, 97-142, 164-230) -----------------------------------------------
// FheOptionInt = (found, index): the reference walks the haystack with a running `found` flag
// (boolean_bitor per position) and adds `!found` to a 4-block index at every position (one
// carry-propagating radix addition per character).  Here: the window matches of `contains`, then
//   1. first_one_hot: "this window is the first match" flags by a 7-ary blocked scan
//      (x = m_w + 2 * (matches before w in its chunk) -> LUT x == 1; chunks recursively), and
//   2. the index as a one-hot weighted sum: block j = sum_w digit_j(w) * first_w (+ digit_j(len) * none):
//      at most one term is non-zero, so no carry ever forms; 15-ary sum + identity-LUT trees.
struct FirstMatch {
    std::vector<Ciphertext> first; // one-hot
    Ciphertext any, none;
};
inline FirstMatch first_one_hot(ServerKey &sk, const std::vector<Ciphertext> &m)
{
    constexpr size_t CH = 7; // m_w + 2 * 6 <= 13 fits the 4-bit message space
    const size_t n = m.size(), chunks = (n + CH - 1) / CH;
    const LookupTable lut_first = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 1); });
    const LookupTable lut_any = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x != 0); });
    const LookupTable lut_none = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 0); });
    const LookupTable lut_both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    FirstMatch r;
    r.first.resize(n);
    std::vector<Ciphertext> chunk_any(chunks);
    Level lvl;
    for (size_t c = 0; c < chunks; c++) {
        Ciphertext before = sk.create_trivial(0);
        for (size_t w = c * CH; w < std::min(n, (c + 1) * CH); w++) {
            Ciphertext x = before;
            sk.unchecked_scalar_mul_assign(x, 2);
            sk.unchecked_add_assign(x, m[w]);
            r.first[w] = std::move(x);
            lvl.add(r.first[w], lut_first);
            sk.unchecked_add_assign(before, m[w]);
        }
        chunk_any[c] = std::move(before); // sum of the chunk's matches
    }
    if (chunks == 1) {
        r.any = chunk_any[0];
        r.none = chunk_any[0];
        lvl.add(r.any, lut_any);
        lvl.add(r.none, lut_none);
        lvl.run(sk);
        return r;
    }
    for (auto &c : chunk_any) lvl.add(c, lut_any);
    lvl.run(sk);
    if (chunks <= 8) { // 1 + 2 * (chunks with a match before this one) <= 15: no scan over the chunks, two levels in all
        Ciphertext run = sk.create_trivial(0);
        for (size_t c = 0; c < chunks; c++) {
            Ciphertext twice = run;
            sk.unchecked_scalar_mul_assign(twice, 2);
            for (size_t w = c * CH; w < std::min(n, (c + 1) * CH); w++) {
                sk.unchecked_add_assign(r.first[w], twice);
                lvl.add(r.first[w], lut_first);
            }
            sk.unchecked_add_assign(run, chunk_any[c]);
        }
        r.any = run;
        r.none = run;
        lvl.add(r.any, lut_any);
        lvl.add(r.none, lut_none);
        lvl.run(sk);
        return r;
    }
    FirstMatch up = first_one_hot(sk, chunk_any);
    for (size_t w = 0; w < n; w++) {
        sk.unchecked_add_assign(r.first[w], up.first[w / CH]);
        lvl.add(r.first[w], lut_both);
    }
    lvl.run(sk);
    r.any = std::move(up.any);
    r.none = std::move(up.none);
    return r;
}

// index = sum_k values[k] * flags[k] for one-hot `flags` (at most one encrypts 1), as `blocks` clean blocks
inline RadixCiphertext one_hot_index(ServerKey &sk, const std::vector<const Ciphertext *> &flags, const std::vector<uint64_t> &values,
                                     size_t blocks)
{
    const uint64_t mm = sk.message_modulus();
    // The flags are one-hot: whatever the number of terms, a sum holds at most ONE non-zero digit (<= 3) and cannot
    // overflow.  What grows with the terms is the noise (variance ~ sum of digit^2 output-noise units, which stay far
    // below the keyswitch noise up to a few hundred), so the groups are cut by that budget, not by the value 15 the
    // reference's boolean sums are bound by (scalar_comparison.rs:164): a 64-char index is one level.
    const size_t budget = 512;
    const LookupTable ident = sk.generate_lookup_table([mm](uint64_t x) { return x % mm; });
    std::vector<std::vector<Ciphertext>> groups(blocks);
    for (size_t j = 0; j < blocks; j++) {
        Ciphertext acc = sk.create_trivial(0);
        size_t weight = 0;
        for (size_t k = 0; k < flags.size(); k++) {
            const uint64_t d = (values[k] >> (2 * j)) % mm; // message_modulus 4: 2 bits per block
            if (d == 0) continue;
            if (weight + d * d > budget) {
                groups[j].push_back(std::move(acc));
                acc = sk.create_trivial(0);
                weight = 0;
            }
            Ciphertext t = *flags[k];
            sk.unchecked_scalar_mul_assign(t, d);
            sk.unchecked_add_assign(acc, t);
            weight += d * d;
        }
        groups[j].push_back(std::move(acc));
    }
    for (;;) {
        Level lvl;
        for (auto &g : groups)
            for (auto &c : g)
                if (!c.is_trivial()) lvl.add(c, ident);
        lvl.run(sk);
        bool done = true;
        for (auto &g : groups) {
            if (g.size() == 1) continue;
            done = false;
            std::vector<Ciphertext> next;
            for (size_t b = 0; b < g.size(); b += budget) { // clean blocks now: one unit of noise each
                Ciphertext acc = g[b];
                for (size_t t = b + 1; t < std::min(g.size(), b + budget); t++) sk.unchecked_add_assign(acc, g[t]);
                next.push_back(std::move(acc));
            }
            g = std::move(next);
        }
        if (done) break;
    }
    RadixCiphertext out(blocks);
    for (size_t j = 0; j < blocks; j++) out[j] = std::move(groups[j][0]);
    return out;
}

// lengths, indices and counts are radix ciphertexts of sk.index_blocks() blocks: 4 (8 bits: create_zero(), trim.rs:42-44) unless the
// key was told otherwise -- the reference's gen_keys_number_blocks (fhe_strings/ciphertext.rs:113-126: "larger than 4 to work with
// strings of length larger than 8 bits")
struct FheOptionInt {
    Ciphertext found;
    RadixCiphertext index;
};
inline RadixCiphertext trivial_radix(const ServerKey &sk, uint64_t v, size_t blocks)
{
    RadixCiphertext r(blocks);
    for (size_t j = 0; j < blocks; j++) r[j] = sk.create_trivial((v >> (2 * j)) & 3);
    return r;
}
// matches[w]: "the pattern starts at position w" for w < matches.size(); reverse: rfind
inline FheOptionInt option_from_matches(ServerKey &sk, std::vector<Ciphertext> matches, size_t hay_len, bool reverse)
{
    if (hay_len > sk.max_index()) throw std::runtime_error("find: the string is longer than the index radix can say (" + std::to_string(sk.max_index()) + " chars with " + std::to_string(sk.index_blocks()) + " blocks: tfhe_hip_host_key_set_num_blocks)");
    if (reverse) std::reverse(matches.begin(), matches.end());
    const size_t W = matches.size();
    FirstMatch fm = first_one_hot(sk, matches);
    std::vector<const Ciphertext *> flags;
    std::vector<uint64_t> values;
    for (size_t k = 0; k < W; k++) {
        flags.push_back(&fm.first[k]);
        values.push_back(reverse ? W - 1 - k : k);
    }
    if (!reverse) { // never found: the index has been incremented at every position (find.rs:173-184)
        flags.push_back(&fm.none);
        values.push_back(hay_len);
    }
    FheOptionInt r;
    r.index = one_hot_index(sk, flags, values, sk.index_blocks());
    r.found = std::move(fm.any);
    return r;
}
inline FheOptionInt string_find(ServerKey &sk, const FheString &s, const FheString &pattern, bool reverse)
{
    if (pattern.empty()) return {sk.create_trivial(1), trivial_radix(sk, reverse ? s.size() : 0, sk.index_blocks())};
    if (pattern.size() > s.size()) return {sk.create_trivial(0), trivial_radix(sk, reverse ? 0 : s.size(), sk.index_blocks())};
    std::vector<size_t> starts(s.size() - pattern.size() + 1);
    for (size_t w = 0; w < starts.size(); w++) starts[w] = w;
    return option_from_matches(sk, match_encrypted_at(sk, s, pattern, starts), s.size(), reverse);
}
inline FheOptionInt string_find_clear(ServerKey &sk, const FheString &s, const std::string &pattern, bool reverse)
{
    if (pattern.empty()) return {sk.create_trivial(1), trivial_radix(sk, reverse ? s.size() : 0, sk.index_blocks())};
    if (pattern.size() > s.size()) return {sk.create_trivial(0), trivial_radix(sk, 0, sk.index_blocks())}; // find.rs:128-134
    std::vector<size_t> starts(s.size() - pattern.size() + 1);
    for (size_t w = 0; w < starts.size(); w++) starts[w] = w;
    return option_from_matches(sk, match_clear_at(sk, s, pattern, starts), s.size(), reverse);
}

// ---- lexicographic order (comparisons.rs:74-99, 299-330, 386-507) -----------------------------------
// For strings without padding (no zero bytes) `le` / `ge` are the byte-wise lexicographic order with
// the clear lengths as tie-break.  The reference walks the characters with a chain of
// eq / le / cmux per character; here the common prefix is ONE radix comparison (character 0 most
// significant): packed-pair sign blocks in one launch, the comparator's sign tree, one final PBS.
inline RadixCiphertext string_as_radix(const ServerKey &sk, const FheString &s, size_t chars)
{
    RadixCiphertext r; // least significant block first: last character first, its low block first;
    for (size_t n = chars; n-- > 0;)  // contents shorter than `chars` are extended with trivial zero chars
        for (size_t b = 0; b < 4; b++) r.push_back(n < s.size() ? s[n][b] : sk.create_trivial(0));
    return r;
}
// `le` / `ge` for strings with no padding or final padding (compare_no_init_padding, comparisons.rs:386-441):
// a padding zero is smaller than every character and the shorter content extended with zeros encrypts the
// same string, so the order is the radix order of the zero-extended contents.
inline Ciphertext string_compare(ServerKey &sk, const FheString &s1, const FheString &s2, bool want_le)
{
    const Comparator c(sk);
    const size_t chars = std::max(s1.size(), s2.size());
    if (chars == 0) return sk.create_trivial(1);
    Ciphertext sign = c.compare(string_as_radix(sk, s1, chars), string_as_radix(sk, s2, chars));
    return c.map_signs({std::move(sign)}, {sign_predicate(want_le ? Cmp::LE : Cmp::GE)})[0];
}
inline Ciphertext string_compare_clear(ServerKey &sk, const FheString &s1, const std::string &s2, bool want_le)
{
    const Comparator c(sk);
    const size_t chars = std::max(s1.size(), s2.size());
    if (chars == 0) return sk.create_trivial(1);
    const RadixCiphertext zero_char = trivial_radix(sk, 0, 4);
    std::vector<std::vector<Ciphertext>> g(1);
    for (size_t n = chars; n-- > 0;) c.stage_scalar(n < s1.size() ? s1[n] : zero_char, n < s2.size() ? (uint8_t)s2[n] : 0, g[0]);
    c.run_signs(g);
    Ciphertext sign = c.reduce(std::move(g))[0];
    return c.map_signs({std::move(sign)}, {sign_predicate(want_le ? Cmp::LE : Cmp::GE)})[0];
}

// ---- strings with FINAL padding (Padding::Final: zero chars after the last character) -----------------
enum class Padding { None = 0, Final = 1, Initial = 2, InitialAndFinal = 3, Anywhere = 4 }; // ciphertext.rs:21-29

// the two packed blocks of every listed char through the LUT x == 0, summed: Z = 2 <=> the char is zero
// (scalar_eq_parallelized(char, 0), scalar_comparison.rs:366-460); one launch
inline std::vector<Ciphertext> char_zero_sums(ServerKey &sk, const std::vector<const RadixCiphertext *> &chars)
{
    const LookupTable is_zero = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 0); });
    std::vector<Ciphertext> lo(chars.size()), hi(chars.size());
    Level lvl;
    for (size_t i = 0; i < chars.size(); i++) {
        lo[i] = pack_block_pair(sk, *chars[i], 0);
        hi[i] = pack_block_pair(sk, *chars[i], 2);
        lvl.add(lo[i], is_zero);
        lvl.add(hi[i], is_zero);
    }
    lvl.run(sk);
    for (size_t i = 0; i < chars.size(); i++) sk.unchecked_add_assign(lo[i], hi[i]);
    return lo;
}

// eq_no_init_padding (comparisons.rs:184-212): the overlapping contents must be equal and the first extra
// char of the longer content must be a padding zero
inline Ciphertext string_eq_padded(ServerKey &sk, const FheString &s1, Padding p1, const FheString &s2, Padding p2)
{
    if (p1 == Padding::None && p2 == Padding::None) return string_eq(sk, s1, s2);
    const FheString &longer = s1.size() >= s2.size() ? s1 : s2;
    const size_t common = std::min(s1.size(), s2.size());
    if (longer.size() > common && (s1.size() > s2.size() ? p1 : p2) == Padding::None) return sk.create_trivial(0);
    const LookupTable eq = equality_lut(sk);
    const LookupTable is_zero = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 0); });
    std::vector<Ciphertext> cmp;
    for (size_t c = 0; c < common; c++) stage_eq_blocks(sk, s1[c], s2[c], cmp);
    const size_t n_eq = cmp.size();
    if (longer.size() > common) {
        cmp.push_back(pack_block_pair(sk, longer[common], 0));
        cmp.push_back(pack_block_pair(sk, longer[common], 2));
    }
    Level lvl;
    for (size_t i = 0; i < cmp.size(); i++) lvl.add(cmp[i], i < n_eq ? eq : is_zero);
    lvl.run(sk);
    return are_all_comparisons_block_true(sk, std::move(cmp));
}

// starts_with_encrypted_vec (contains.rs:93-130) for every listed start position of `s` at once; `pattern`
// may carry final padding: a pattern char matches if it equals the string's char OR is a padding zero; a
// pattern that overhangs the content must have a padding zero right after it.
//   launch 1: packed-pair equalities of every (window, char) + zero bits of the pattern chars
//   launch 2: per (window, char)  E + 5 Z -> (E == 2 || Z == 2)   (E: equal pairs, Z: zero bits)
//   then the all-true tree per window.
inline std::vector<Ciphertext> match_padded_at(ServerKey &sk, const FheString &s, const FheString &pattern,
                                               const std::vector<size_t> &starts)
{
    const LookupTable eq = equality_lut(sk);
    const LookupTable char_ok = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x % 5 == 2 || x / 5 == 2); });
    const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    std::vector<const RadixCiphertext *> pchars;
    for (const auto &c : pattern) pchars.push_back(&c);
    std::vector<std::vector<Ciphertext>> eqb(starts.size()); // [window][2 * j + packed pair]
    Level lvl;
    for (size_t g = 0; g < starts.size(); g++)
        for (size_t j = 0; j < pattern.size() && starts[g] + j < s.size(); j++) stage_eq_blocks(sk, s[starts[g] + j], pattern[j], eqb[g]);
    // the zero bits ride in the same launch as the block equalities
    const LookupTable is_zero = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 0); });
    std::vector<Ciphertext> zlo(pattern.size()), zhi(pattern.size());
    for (size_t j = 0; j < pattern.size(); j++) {
        zlo[j] = pack_block_pair(sk, pattern[j], 0);
        zhi[j] = pack_block_pair(sk, pattern[j], 2);
        lvl.add(zlo[j], is_zero);
        lvl.add(zhi[j], is_zero);
    }
    for (auto &g : eqb)
        for (auto &c : g) lvl.add(c, eq);
    lvl.run(sk);
    std::vector<Ciphertext> Z(pattern.size());
    for (size_t j = 0; j < pattern.size(); j++) {
        Z[j] = zlo[j];
        sk.unchecked_add_assign(Z[j], zhi[j]);
    }
    std::vector<std::vector<Ciphertext>> groups(starts.size());
    std::vector<Ciphertext> overhang(pattern.size()); // "pattern char j is a padding zero", where needed
    std::vector<bool> need(pattern.size(), false);
    for (size_t g = 0; g < starts.size(); g++) {
        const size_t fit = std::min(pattern.size(), s.size() - starts[g]);
        for (size_t j = 0; j < fit; j++) {
            Ciphertext x = Z[j];
            sk.unchecked_scalar_mul_assign(x, 5);
            for (size_t b = 0; b < 2; b++) sk.unchecked_add_assign(x, eqb[g][2 * j + b]);
            groups[g].push_back(std::move(x));
        }
        if (fit < pattern.size()) need[fit] = true;
    }
    for (auto &g : groups)
        for (auto &c : g) lvl.add(c, char_ok);
    for (size_t j = 0; j < pattern.size(); j++)
        if (need[j]) {
            overhang[j] = Z[j];
            lvl.add(overhang[j], both);
        }
    lvl.run(sk);
    for (size_t g = 0; g < starts.size(); g++) {
        const size_t fit = std::min(pattern.size(), s.size() - starts[g]);
        if (fit < pattern.size()) groups[g].push_back(overhang[fit]);
    }
    return reduce_boolean_groups(sk, std::move(groups), true);
}
inline std::vector<size_t> all_positions(size_t n)
{
    std::vector<size_t> v(n);
    for (size_t i = 0; i < n; i++) v[i] = i;
    return v;
}
// pattern "is the empty string" (possibly with padding): its first char is a padding zero
inline Ciphertext pattern_is_empty(ServerKey &sk, const FheString &pattern)
{
    if (pattern.empty()) return sk.create_trivial(1);
    const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    std::vector<Ciphertext> z = char_zero_sums(sk, {&pattern[0]});
    Level lvl;
    lvl.add(z[0], both);
    lvl.run(sk);
    return z[0];
}
// starts_with_encrypted_no_init_padding (starts_with.rs:22-62)
inline Ciphertext string_starts_with_padded(ServerKey &sk, const FheString &s, Padding ps, const FheString &prefix, Padding pp)
{
    (void)ps; // padding zeros of `s` are ordinary chars for the comparison
    if (pp == Padding::None) return string_starts_with(sk, s, prefix);
    if (prefix.empty()) return sk.create_trivial(1);
    if (s.empty()) return pattern_is_empty(sk, prefix);
    return match_padded_at(sk, s, prefix, {0})[0];
}
// contains_unpadded_string (contains.rs:77-91)
inline Ciphertext string_contains_padded(ServerKey &sk, const FheString &s, Padding ps, const FheString &pattern, Padding pp)
{
    (void)ps;
    if (pp == Padding::None) return string_contains(sk, s, pattern);
    if (pattern.empty()) return sk.create_trivial(1);
    if (s.empty()) return pattern_is_empty(sk, pattern);
    return is_at_least_one_comparisons_block_true(sk, match_padded_at(sk, s, pattern, all_positions(s.size())));
}
inline RadixCiphertext string_len_of(ServerKey &sk, const FheString &s, Padding p); // = string_len, defined below
// connected_find_unpadded_string (find.rs:164-184) and its mirror from the end
inline FheOptionInt string_find_padded(ServerKey &sk, const FheString &s, Padding ps, const FheString &pattern, Padding pp, bool reverse)
{
    if (pp == Padding::None) return string_find(sk, s, pattern, reverse);
    if (pattern.empty()) return {sk.create_trivial(1), trivial_radix(sk, 0, sk.index_blocks())};
    if (s.empty()) return {pattern_is_empty(sk, pattern), trivial_radix(sk, 0, sk.index_blocks())};
    FheOptionInt r = option_from_matches(sk, match_padded_at(sk, s, pattern, all_positions(s.size())), s.size(), reverse);
    if (!reverse) return r; // a pattern that is padding only matches at every position: Some(0), like "abc".find("")
    // from the end the EMPTY pattern is found behind the last char: "abc".rfind("") == Some(3) = the length (encrypted when `s`
    // carries padding) -- rfind_from_final_padding_allow_empty_pattern, find.rs:435-470, 545-565
    const Ciphertext is_empty = pattern_is_empty(sk, pattern);
    r.index = if_then_else_parallelized(sk, is_empty, string_len_of(sk, s, ps), r.index);
    return r;
}

// ---- trim / strip (trim.rs:123-230, strip.rs:12-260) ---------------------------------------------
// The reference walks the content with a running `continue_triming` flag (bitand + cmux + radix sub per
// character).  Here: the per-character class flags in two launches, a 15-ary blocked prefix-OR scan for
// "a character that stays has been seen", one launch that zeroes the trimmed characters, and the new
// length as a one-hot weighted sum.  Outputs carry padding zeros where characters were removed
// (Padding::InitialAndFinal, as in the reference) and an encrypted 4-block length.
struct FheStringOut {
    FheString content;
    RadixCiphertext length; // encrypted length (FheStrLength::Encrypted), index_blocks() blocks
};

// inclusive prefix OR of boolean blocks: out[n] = m[0] | ... | m[n]
inline std::vector<Ciphertext> prefix_or(ServerKey &sk, const std::vector<Ciphertext> &m)
{
    const size_t CH = (size_t)(sk.message_modulus() * sk.carry_modulus() - 1); // 15
    const size_t n = m.size(), chunks = (n + CH - 1) / CH;
    const LookupTable lut_any = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x != 0); });
    std::vector<Ciphertext> local(n);
    Level lvl;
    for (size_t c = 0; c < chunks; c++) {
        Ciphertext run = sk.create_trivial(0);
        for (size_t w = c * CH; w < std::min(n, (c + 1) * CH); w++) {
            sk.unchecked_add_assign(run, m[w]);
            local[w] = run;
            lvl.add(local[w], lut_any);
        }
    }
    lvl.run(sk);
    if (chunks <= 1) return local;
    std::vector<Ciphertext> totals(chunks);
    for (size_t c = 0; c < chunks; c++) totals[c] = local[std::min(n, (c + 1) * CH) - 1];
    // "a flag is set in an earlier chunk": up to 15 chunks (225 flags) the plain sum of the chunks' own flags fits the
    // message space -- no scan over the chunks, two levels in all; beyond that the chunk flags are scanned recursively
    std::vector<Ciphertext> before(chunks);
    if (chunks <= CH) {
        Ciphertext run = sk.create_trivial(0);
        for (size_t c = 0; c < chunks; c++) {
            before[c] = run;
            sk.unchecked_add_assign(run, totals[c]);
        }
    } else {
        const std::vector<Ciphertext> up = prefix_or(sk, totals);
        before[0] = sk.create_trivial(0);
        for (size_t c = 1; c < chunks; c++) before[c] = up[c - 1];
    }
    for (size_t w = CH; w < n; w++) {
        sk.unchecked_add_assign(local[w], before[w / CH]);
        lvl.add(local[w], lut_any);
    }
    lvl.run(sk);
    return local;
}

// is_ascii_white_space (split_ascii_whitespace.rs:63-71): 9 <= c <= 13 or c == 32
inline std::vector<Ciphertext> ascii_whitespace_flags(ServerKey &sk, const FheString &s)
{
    const LookupTable lut_h = sk.generate_lookup_table([](uint64_t h) { return h == 0 ? (uint64_t)1 : (h == 2 ? (uint64_t)2 : (uint64_t)0); });
    const LookupTable lut_l = sk.generate_lookup_table([](uint64_t l) { return (uint64_t)(l >= 9 && l <= 13) | ((uint64_t)(l == 0) << 1); });
    const LookupTable lut_ws = sk.generate_lookup_table_bivariate(
        [](uint64_t h, uint64_t l) { return (uint64_t)((h == 1 && (l & 1)) || (h == 2 && (l & 2))); });
    std::vector<Ciphertext> hs(s.size()), ls(s.size());
    Level lvl;
    for (size_t n = 0; n < s.size(); n++) {
        hs[n] = pack_block_pair(sk, s[n], 2);
        ls[n] = pack_block_pair(sk, s[n], 0);
        lvl.add(hs[n], lut_h);
        lvl.add(ls[n], lut_l);
    }
    lvl.run(sk);
    for (size_t n = 0; n < s.size(); n++) {
        sk.bivariate_pack_assign(hs[n], ls[n]);
        lvl.add(hs[n], lut_ws);
    }
    lvl.run(sk);
    return hs;
}
// eq_clear_char for every character (comparisons.rs:528)
inline std::vector<Ciphertext> char_eq_flags(ServerKey &sk, const FheString &s, uint8_t byte)
{
    ScalarEqLuts luts;
    const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    std::vector<Ciphertext> packed;
    std::vector<unsigned> nib;
    for (const auto &c : s) stage_scalar_eq_char(sk, c, byte, packed, nib);
    Level lvl;
    for (size_t i = 0; i < packed.size(); i++) lvl.add(packed[i], luts.get(sk, nib[i]));
    lvl.run(sk);
    std::vector<Ciphertext> out(s.size());
    for (size_t n = 0; n < s.size(); n++) {
        out[n] = packed[2 * n];
        sk.unchecked_add_assign(out[n], packed[2 * n + 1]);
        lvl.add(out[n], both);
    }
    lvl.run(sk);
    return out;
}

// removes the leading (left) and / or trailing (right) run of flagged characters of an unpadded string
inline FheStringOut trim_flagged(ServerKey &sk, const FheString &s, const std::vector<Ciphertext> &flagged, bool left, bool right)
{
    const size_t L = s.size();
    if (L > sk.max_index()) throw std::runtime_error("trim: the string is longer than the length radix can say (" + std::to_string(sk.max_index()) + " with " + std::to_string(sk.index_blocks()) + " blocks: tfhe_hip_host_key_set_num_blocks)");
    FheStringOut out;
    out.content = s;
    if (L == 0 || (!left && !right)) {
        out.length = trivial_radix(sk, L, sk.index_blocks());
        return out;
    }
    std::vector<Ciphertext> stays(L); // 1 - flagged
    for (size_t n = 0; n < L; n++) {
        stays[n] = sk.create_trivial(1);
        sk.lwe_sub_assign(stays[n], flagged[n]);
        stays[n].degree = 1;
        stays[n].noise_level = flagged[n].noise_level;
    }
    std::vector<Ciphertext> seen_l, seen_r; // a staying character at or before / at or after n
    if (left) seen_l = prefix_or(sk, stays);
    if (right) {
        std::vector<Ciphertext> rev(stays.rbegin(), stays.rend());
        rev = prefix_or(sk, rev);
        seen_r.assign(rev.rbegin(), rev.rend());
    }
    // keep the character iff all requested sides have seen a staying character
    const uint64_t want = (uint64_t)left + (uint64_t)right;
    const LookupTable keep = sk.generate_lookup_table_bivariate([want](uint64_t block, uint64_t c) { return c == want ? block : (uint64_t)0; });
    Level lvl;
    for (size_t n = 0; n < L; n++) {
        Ciphertext cond = left ? seen_l[n] : seen_r[n];
        if (left && right) sk.unchecked_add_assign(cond, seen_r[n]);
        for (size_t b = 0; b < 4; b++) {
            sk.bivariate_pack_assign(out.content[n][b], cond);
            lvl.add(out.content[n][b], keep);
        }
    }
    lvl.run(sk);
    // one-hot "first staying character" flags are differences of neighbouring scan values
    auto first_flags = [&](const std::vector<Ciphertext> &seen, bool from_left) {
        std::vector<Ciphertext> f(L);
        for (size_t n = 0; n < L; n++) {
            f[n] = seen[n];
            const bool has_prev = from_left ? n > 0 : n + 1 < L;
            if (has_prev) sk.lwe_sub_assign(f[n], seen[from_left ? n - 1 : n + 1]);
            f[n].degree = 1;
        }
        return f;
    };
    std::vector<const Ciphertext *> flags;
    std::vector<uint64_t> values;
    std::vector<Ciphertext> fl, fr;
    if (left && !right) { // length = L - (index of the first staying character); nothing stays: 0
        fl = first_flags(seen_l, true);
        for (size_t n = 0; n < L; n++) { flags.push_back(&fl[n]); values.push_back(L - n); }
        out.length = one_hot_index(sk, flags, values, sk.index_blocks());
    } else if (right && !left) { // length = (index of the last staying character) + 1
        fr = first_flags(seen_r, false);
        for (size_t n = 0; n < L; n++) { flags.push_back(&fr[n]); values.push_back(n + 1); }
        out.length = one_hot_index(sk, flags, values, sk.index_blocks());
    } else { // (last + 1) - first; both sums are 0 when nothing stays
        fl = first_flags(seen_l, true);
        fr = first_flags(seen_r, false);
        std::vector<const Ciphertext *> fa, fb;
        std::vector<uint64_t> va, vb;
        for (size_t n = 0; n < L; n++) { fa.push_back(&fr[n]); va.push_back(n + 1); fb.push_back(&fl[n]); vb.push_back(n); }
        out.length = sub_parallelized(sk, one_hot_index(sk, fa, va, sk.index_blocks()), one_hot_index(sk, fb, vb, sk.index_blocks()));
    }
    return out;
}
// trim_start (trim.rs:202-226): ASCII whitespace
inline FheStringOut string_trim_start(ServerKey &sk, const FheString &s) { return trim_flagged(sk, s, ascii_whitespace_flags(sk, s), true, false); }
// trim_end / trim (trim.rs:129-147): the reference trims the space character (ASCII_WHITE_SPACE = 32) here
inline FheStringOut string_trim_end(ServerKey &sk, const FheString &s) { return trim_flagged(sk, s, char_eq_flags(sk, s, 32), false, true); }
inline FheStringOut string_trim(ServerKey &sk, const FheString &s) { return trim_flagged(sk, s, char_eq_flags(sk, s, 32), true, true); }
// trim_start_char / trim_end_char / trim_char (trim.rs:123-163)
inline FheStringOut string_trim_char(ServerKey &sk, const FheString &s, uint8_t ch, bool left, bool right)
{
    return trim_flagged(sk, s, char_eq_flags(sk, s, ch), left, right);
}

// FheOptionString (strip.rs:9): (the pattern was stripped, the string with zeros where it was)
struct FheOptionString {
    Ciphertext found;
    FheStringOut value;
};
// strip_prefix / strip_suffix (strip.rs:12-28, 93-260) on unpadded strings: `found` from starts_with / ends_with,
// the pattern's characters zeroed under it (one launch), the length chosen by it (one launch)
inline FheOptionString strip_at(ServerKey &sk, const FheString &s, Ciphertext found, size_t first, size_t count)
{
    if (s.size() > sk.max_index()) throw std::runtime_error("strip: the string is longer than the length radix can say (" + std::to_string(sk.max_index()) + " with " + std::to_string(sk.index_blocks()) + " blocks: tfhe_hip_host_key_set_num_blocks)");
    FheOptionString r;
    r.value.content = s;
    const size_t L = s.size();
    const LookupTable zero_if = sk.generate_lookup_table_bivariate([](uint64_t block, uint64_t c) { return c ? (uint64_t)0 : block; });
    std::vector<LookupTable> len_lut(sk.index_blocks());
    r.value.length.resize(sk.index_blocks());
    Level lvl;
    for (size_t n = first; n < first + count; n++)
        for (size_t b = 0; b < 4; b++) {
            sk.bivariate_pack_assign(r.value.content[n][b], found);
            lvl.add(r.value.content[n][b], zero_if);
        }
    for (size_t j = 0; j < sk.index_blocks(); j++) {
        const uint64_t a = ((L - count) >> (2 * j)) & 3, b = (L >> (2 * j)) & 3;
        len_lut[j] = sk.generate_lookup_table([a, b](uint64_t c) { return c ? a : b; });
        r.value.length[j] = found;
        lvl.add(r.value.length[j], len_lut[j]);
    }
    lvl.run(sk);
    r.found = std::move(found);
    return r;
}
inline FheOptionString string_strip(ServerKey &sk, const FheString &s, const FheString &pattern, bool suffix)
{
    if (pattern.size() > s.size()) return {sk.create_trivial(0), {s, trivial_radix(sk, s.size(), sk.index_blocks())}};
    Ciphertext found = suffix ? string_ends_with(sk, s, pattern) : string_starts_with(sk, s, pattern);
    return strip_at(sk, s, std::move(found), suffix ? s.size() - pattern.size() : 0, pattern.size());
}
inline FheOptionString string_strip_clear(ServerKey &sk, const FheString &s, const std::string &pattern, bool suffix)
{
    if (pattern.size() > s.size()) return {sk.create_trivial(0), {s, trivial_radix(sk, s.size(), sk.index_blocks())}};
    Ciphertext found = suffix ? string_ends_with_clear(sk, s, pattern) : string_starts_with_clear(sk, s, pattern);
    return strip_at(sk, s, std::move(found), suffix ? s.size() - pattern.size() : 0, pattern.size());
}

// ---- push_padding_to_end (change_padding.rs:5-21, 56-93) --------------------------------------------
// The reference pops the first non-zero character L times; every pop walks the content with a radix
// mul / add / sub per character: O(L^2) carry-propagating radix operations, strictly sequential.
// Here: stable compaction by a shift network (the classic `compress`: an element moves left by
// digit_j * 4^j for the base-4 digits of its displacement, least significant first; the final positions
// never collide, and a base-4 step is the composition of two collision-free binary steps).  The
// displacement of a character is the number of padding zeros before it; to get its DIGITS without a radix
// addition the string is compacted in two phases: inside chunks of 8 (displacement <= 7: a leveled
// count, one PBS per digit), then globally by the number of zeros in the preceding chunks (a radix sum
// per chunk: its blocks ARE the base-4 digits).  One routing step = one launch: per position and block
// the sum over the (up to four) sources of "take it iff its digit says so".
struct RoutedChar {
    RadixCiphertext data;            // 4 blocks
    std::vector<Ciphertext> digits;  // displacement digits still to be consumed (index 0 = next step), values < radix
    Ciphertext present;              // the position holds a character
};

// elements move left by digit * unit where digit = digits[0] (< radix); positions only interact inside windows
inline void route_stage(ServerKey &sk, std::vector<RoutedChar> &e, size_t unit, size_t radix, size_t window)
{
    const uint64_t mm = sk.message_modulus();
    std::vector<LookupTable> take(radix);
    for (size_t t = 0; t < radix; t++)
        take[t] = sk.generate_lookup_table_bivariate([t](uint64_t x, uint64_t digit) { return digit == t ? x : (uint64_t)0; });
    const size_t L = e.size();
    const size_t nd = e[0].digits.size() - 1; // digits that travel on
    // staged[p][t]: what source p + t * unit contributes to position p: 4 data blocks, nd digits, the presence flag
    struct Contribution {
        std::vector<Ciphertext> v;
    };
    std::vector<std::vector<Contribution>> staged(L);
    Level lvl;
    for (size_t p = 0; p < L; p++) {
        for (size_t t = 0; t < radix; t++) {
            const size_t q = p + t * unit;
            if (q >= L || q / window != p / window) break;
            const RoutedChar &src = e[q];
            Contribution c;
            auto push = [&](const Ciphertext &x) {
                Ciphertext y = x;
                sk.bivariate_pack_assign(y, src.digits[0]);
                c.v.push_back(std::move(y));
            };
            for (size_t b = 0; b < 4; b++) push(src.data[b]);
            for (size_t i = 1; i <= nd; i++) push(src.digits[i]);
            push(src.present);
            staged[p].push_back(std::move(c));
        }
    }
    for (size_t p = 0; p < L; p++)
        for (size_t t = 0; t < staged[p].size(); t++)
            for (auto &x : staged[p][t].v) lvl.add(x, take[t]);
    lvl.run(sk);
    std::vector<RoutedChar> out(L);
    for (size_t p = 0; p < L; p++) {
        std::vector<Ciphertext> sum = std::move(staged[p][0].v);
        for (size_t t = 1; t < staged[p].size(); t++)
            for (size_t i = 0; i < sum.size(); i++) sk.unchecked_add_assign(sum[i], staged[p][t].v[i]);
        RoutedChar &o = out[p];
        o.data.assign(sum.begin(), sum.begin() + 4);
        o.digits.assign(sum.begin() + 4, sum.begin() + 4 + nd);
        o.present = sum.back();
        // at most one source contributes a non-zero value
        for (auto &x : o.data) x.degree = mm - 1;
        for (auto &x : o.digits) x.degree = mm - 1;
        o.present.degree = 1;
    }
    e = std::move(out);
}

// short strings: compaction as a scatter by rank (host_split.hpp), half the levels of the shift network
inline FheString push_padding_to_end_by_rank(ServerKey &sk, const FheString &s);
constexpr size_t COMPACT_BY_RANK_MAX = 24;

inline FheString push_padding_to_end(ServerKey &sk, const FheString &s)
{
    const size_t L = s.size();
    if (L <= 1) return s;
    if (L > sk.max_index()) throw std::runtime_error("push_padding_to_end: the string is longer than the position radix can say (" + std::to_string(sk.max_index()) + " with " + std::to_string(sk.index_blocks()) + " blocks: tfhe_hip_host_key_set_num_blocks)");
    if (L <= COMPACT_BY_RANK_MAX && std::getenv("TFHE_HIP_COMPACT_NETWORK") == nullptr) return push_padding_to_end_by_rank(sk, s);
    constexpr size_t CH = 8;
    const size_t chunks = (L + CH - 1) / CH;
    const uint64_t mm = sk.message_modulus();
    // presence flags: char != 0
    std::vector<const RadixCiphertext *> chars;
    for (const auto &c : s) chars.push_back(&c);
    std::vector<Ciphertext> zs = char_zero_sums(sk, chars); // 2 <=> zero
    const LookupTable lut_zero = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    {
        Level lvl;
        for (auto &z : zs) lvl.add(z, lut_zero);
        lvl.run(sk);
    }
    std::vector<RoutedChar> e(L);
    // phase 1 displacement: zeros before the char inside its chunk (<= 7), masked by presence: x = before + 8 * zero;
    // digit 0 = x mod 4 (base 4), digit 1 = x / 4 (0 or 1)
    const LookupTable d_lo = sk.generate_lookup_table([](uint64_t x) { return x >= 8 ? (uint64_t)0 : x & 3; });
    const LookupTable d_hi = sk.generate_lookup_table([](uint64_t x) { return x >= 8 ? (uint64_t)0 : x >> 2; });
    const LookupTable lut_not = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 0); });
    std::vector<Ciphertext> chunk_zeros(chunks);
    {
        Level lvl;
        for (size_t c = 0; c < chunks; c++) {
            Ciphertext before = sk.create_trivial(0);
            for (size_t p = c * CH; p < std::min(L, (c + 1) * CH); p++) {
                Ciphertext x = zs[p];
                sk.unchecked_scalar_mul_assign(x, 8);
                sk.unchecked_add_assign(x, before);
                e[p].data = s[p];
                e[p].digits.assign(2, x); // Level keeps pointers to the elements: sized before staging
                lvl.add(e[p].digits[0], d_lo);
                lvl.add(e[p].digits[1], d_hi);
                e[p].present = zs[p];
                lvl.add(e[p].present, lut_not);
                sk.unchecked_add_assign(before, zs[p]);
            }
            chunk_zeros[c] = std::move(before); // zeros in the chunk (<= 8); a short last chunk counts what it has
        }
        lvl.run(sk);
    }
    route_stage(sk, e, 1, 4, CH);
    route_stage(sk, e, 4, 2, CH);
    if (chunks > 1) {
        // phase 2 displacement of chunk c: zeros in the chunks before it, as a radix sum whose blocks are its base-4 digits
        const LookupTable lo = sk.generate_lookup_table([mm](uint64_t x) { return x % mm; });
        const LookupTable hi = sk.generate_lookup_table([mm](uint64_t x) { return x / mm; });
        std::vector<RadixCiphertext> zr(chunks);
        {
            Level lvl;
            for (size_t c = 0; c + 1 < chunks; c++) {
                zr[c] = trivial_radix(sk, 0, sk.index_blocks());
                zr[c][0] = chunk_zeros[c];
                zr[c][1] = chunk_zeros[c];
                lvl.add(zr[c][0], lo);
                lvl.add(zr[c][1], hi);
            }
            lvl.run(sk);
        }
        size_t ndig = 0;
        while (((size_t)1 << (2 * ndig)) < L) ndig++; // the largest displacement is < L <= 4^ndig
        std::vector<RadixCiphertext> offsets(chunks);
        for (size_t c = 1; c < chunks; c++) {
            std::vector<RadixCiphertext> terms(zr.begin(), zr.begin() + c);
            offsets[c] = terms.size() == 1 ? terms[0] : unchecked_sum_ciphertexts_vec_parallelized(sk, std::move(terms));
        }
        // the digits of an absent position must be zero: digit * present (one PBS per digit and position)
        const LookupTable masked = sk.generate_lookup_table_bivariate([](uint64_t digit, uint64_t present) { return present ? digit : (uint64_t)0; });
        {
            Level lvl;
            for (size_t p = 0; p < L; p++) {
                const size_t c = p / CH;
                e[p].digits.assign(ndig, sk.create_trivial(0));
                if (c == 0) continue;
                for (size_t j = 0; j < ndig; j++) {
                    e[p].digits[j] = offsets[c][j];
                    sk.bivariate_pack_assign(e[p].digits[j], e[p].present);
                    lvl.add(e[p].digits[j], masked);
                }
            }
            lvl.run(sk);
        }
        size_t unit = 1;
        for (size_t j = 0; j < ndig; j++, unit *= 4) route_stage(sk, e, unit, 4, L);
    }
    FheString out(L);
    for (size_t p = 0; p < L; p++) out[p] = std::move(e[p].data);
    return out;
}

// what every operation of the reference does first with initial / anywhere padding (contains.rs:25-36, find.rs:112-122, ...)
inline FheString normalize_padding(ServerKey &sk, const FheString &s, Padding &p)
{
    if (p == Padding::None || p == Padding::Final) return s;
    p = Padding::Final;
    return push_padding_to_end(sk, s);
}

// ---- len / is_empty on padded strings (is_empty.rs:22-62; the reference carries the encrypted length with the
// string, here it is recomputed from the content: the number of non-zero characters) -----------------------
inline std::vector<Ciphertext> char_is_zero_flags(ServerKey &sk, const FheString &s)
{
    std::vector<const RadixCiphertext *> chars;
    for (const auto &c : s) chars.push_back(&c);
    std::vector<Ciphertext> z = char_zero_sums(sk, chars);
    const LookupTable both = sk.generate_lookup_table([](uint64_t x) { return (uint64_t)(x == 2); });
    Level lvl;
    for (auto &c : z) lvl.add(c, both);
    lvl.run(sk);
    return z;
}
inline RadixCiphertext string_len(ServerKey &sk, const FheString &s, Padding p)
{
    if (s.size() > sk.max_index()) throw std::runtime_error("len: the string is longer than the length radix can say (" + std::to_string(sk.max_index()) + " with " + std::to_string(sk.index_blocks()) + " blocks: tfhe_hip_host_key_set_num_blocks)");
    if (p == Padding::None || s.empty()) return trivial_radix(sk, s.size(), sk.index_blocks());
    std::vector<Ciphertext> z = char_is_zero_flags(sk, s);
    // the real chars are counted 15 at a time by leveled sums; each chunk count becomes a two-digit radix number (one
    // launch for all chunks) and only those few numbers go through the carry-propagating sum: a 12-char string is 3
    // launches, 100 chars add 7 numbers instead of 100 bits
    constexpr size_t CH = 15;
    const uint64_t mm = sk.message_modulus();
    const LookupTable lo = sk.generate_lookup_table([mm](uint64_t x) { return x % mm; });
    const LookupTable hi = sk.generate_lookup_table([mm](uint64_t x) { return x / mm; });
    std::vector<RadixCiphertext> terms;
    Level lvl;
    const size_t chunks = (s.size() + CH - 1) / CH;
    terms.reserve(chunks);
    for (size_t c = 0; c < chunks; c++) {
        const size_t n1 = std::min(s.size(), (c + 1) * CH);
        Ciphertext count = sk.create_trivial((uint64_t)0);
        for (size_t n = c * CH; n < n1; n++) { // + (1 - zero flag)
            sk.unchecked_scalar_add_assign(count, 1);
            sk.lwe_sub_assign(count, z[n]);
            count.noise_level += z[n].noise_level;
        }
        count.degree = n1 - c * CH;
        terms.push_back(trivial_radix(sk, 0, sk.index_blocks()));
        terms.back()[0] = count;
        terms.back()[1] = count;
    }
    for (auto &t : terms) { // (the vector is complete: Level keeps pointers)
        lvl.add(t[0], lo);
        lvl.add(t[1], hi);
    }
    lvl.run(sk);
    if (terms.size() == 1) return terms[0];
    if (terms.size() <= 4) { // up to 60 chars: the two digit columns fit the message space, a carry per column: two launches
        Ciphertext c0 = terms[0][0], c1 = terms[0][1];
        for (size_t t = 1; t < terms.size(); t++) {
            sk.unchecked_add_assign(c0, terms[t][0]);
            sk.unchecked_add_assign(c1, terms[t][1]);
        }
        RadixCiphertext r = trivial_radix(sk, 0, sk.index_blocks());
        Ciphertext carry0 = c0;
        r[0] = c0;
        lvl.add(r[0], lo);
        lvl.add(carry0, hi);
        lvl.run(sk);
        sk.unchecked_add_assign(c1, carry0); // <= 4 * 3 + 3
        Ciphertext carry1 = c1;
        r[1] = c1;
        lvl.add(r[1], lo);
        lvl.add(carry1, hi);
        lvl.run(sk);
        r[2] = carry1; // <= 3: the length is below 64
        return r;
    }
    return unchecked_sum_ciphertexts_vec_parallelized(sk, std::move(terms));
}
inline RadixCiphertext string_len_of(ServerKey &sk, const FheString &s, Padding p) { return string_len(sk, s, p); }
inline Ciphertext string_is_empty(ServerKey &sk, const FheString &s, Padding p)
{
    if (s.empty()) return sk.create_trivial(1);
    if (p == Padding::None) return sk.create_trivial(0);
    return are_all_comparisons_block_true(sk, char_is_zero_flags(sk, s));
}

// ---- case (change_case.rs:18-83) ---------------------------------------------------------------------
// The reference forms (c > 96) & (c < 123) with two scalar comparisons and a bitand, multiplies the
// boolean by 32 and subtracts with carry propagation.  A letter differs from its other case only in
// block 2 (bits 4,5: 0x40 | 0x20 | ...), so three launches per string suffice, 4 PBS per character:
//   1. packed nibbles: H -> {1 if H == hi(first letter row), 2 if H == hi(second row), 0}, L -> 2 bits
//      (L >= 1, L <= 10);  2. both -> "is a letter of the case to change";  3. block 2 +- 2 if so.
// leveled == true (default): block 2 is adjusted by a LEVELED +-2 * "is a letter of the other case" instead of a third
// bootstrap (a letter's block 2 is 0 / 1 in upper and 2 / 3 in lower case: no carry can form).  The block then carries
// 1 + 4 units of output noise instead of 1 -- inside the budget results are handed back with (cf. replace) -- and the
// operation is two launches, three bootstraps per char.
inline FheString change_case(ServerKey &sk, const FheString &s, bool to_upper, bool leveled = true)
{
    const uint64_t mm = sk.message_modulus();
    const uint64_t row0 = to_upper ? 6 : 4; // high nibble of 'a'..'o' / 'A'..'O'; the next row holds 'p'..'z'
    const LookupTable lut_h = sk.generate_lookup_table([row0](uint64_t h) { return h == row0 ? (uint64_t)1 : (h == row0 + 1 ? (uint64_t)2 : (uint64_t)0); });
    const LookupTable lut_l = sk.generate_lookup_table([](uint64_t l) { return (uint64_t)(l >= 1) | ((uint64_t)(l <= 10) << 1); });
    const LookupTable lut_is = sk.generate_lookup_table_bivariate(
        [](uint64_t h, uint64_t l) { return (uint64_t)((h == 1 && (l & 1)) || (h == 2 && (l & 2))); });
    const LookupTable lut_b2 = sk.generate_lookup_table_bivariate(
        [mm, to_upper](uint64_t b, uint64_t is) { return is ? (to_upper ? (b + mm - 2) % mm : (b + 2) % mm) : b; });
    FheString out = s;
    std::vector<Ciphertext> hs(s.size()), ls(s.size());
    Level lvl;
    for (size_t n = 0; n < s.size(); n++) {
        if (s[n].size() != 4) throw std::runtime_error("a char is a 4-block radix ciphertext");
        hs[n] = pack_block_pair(sk, s[n], 2);
        ls[n] = pack_block_pair(sk, s[n], 0);
        lvl.add(hs[n], lut_h);
        lvl.add(ls[n], lut_l);
    }
    lvl.run(sk);
    for (size_t n = 0; n < s.size(); n++) {
        sk.bivariate_pack_assign(hs[n], ls[n]);
        lvl.add(hs[n], lut_is);
    }
    lvl.run(sk);
    if (leveled) {
        for (size_t n = 0; n < s.size(); n++) {
            Ciphertext delta = hs[n];
            sk.unchecked_scalar_mul_assign(delta, 2);
            if (to_upper) sk.lwe_sub_assign(out[n][2], delta);
            else sk.unchecked_add_assign(out[n][2], delta);
            out[n][2].degree = mm - 1;
        }
        return out;
    }
    for (size_t n = 0; n < s.size(); n++) {
        sk.bivariate_pack_assign(out[n][2], hs[n]);
        lvl.add(out[n][2], lut_b2);
    }
    lvl.run(sk);
    return out;
}
inline FheString string_to_uppercase(ServerKey &sk, const FheString &s) { return change_case(sk, s, true); }
inline FheString string_to_lowercase(ServerKey &sk, const FheString &s) { return change_case(sk, s, false); }
// eq_ignore_case (comparisons.rs:55-72, 126-150): both sides to lower case, then eq
inline Ciphertext string_eq_ignore_case(ServerKey &sk, const FheString &s1, const FheString &s2)
{
    if (s1.size() != s2.size()) return sk.create_trivial(0);
    FheString both = s1;
    both.insert(both.end(), s2.begin(), s2.end());
    const FheString low = change_case(sk, both, false, true); // one set of launches for the two strings; block 2 adjusted leveled
    return string_eq(sk, FheString(low.begin(), low.begin() + s1.size()), FheString(low.begin() + s1.size(), low.end()));
}
inline Ciphertext string_eq_ignore_case_clear(ServerKey &sk, const FheString &s, std::string clear)
{
    for (auto &ch : clear)
        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32);
    if (s.size() != clear.size()) return sk.create_trivial(0);
    return string_eq_clear(sk, change_case(sk, s, false, true), clear);
}

} // namespace host
} // namespace tfhe_hip

#include "host_split.hpp"
