// csprng.hip -- decompression of seeded (compressed) keys on the GPU.
//
// A seeded key (shortint CompressedServerKey, tfhe/src/shortint/server_key/compressed.rs:44-55) stores only the
// BODIES of its ciphertexts and one 128-bit compression seed; the masks are the output of the reference's CSPRNG
// (concrete-csprng/src/generators/aes_ctr: AES-128 in counter mode, key = the seed's 16 native bytes, block i =
// AES(i as 16 little-endian bytes); a fresh generator starts at byte 1 of block 0 (generic.rs:38, TableIndex::SECOND);
// forked children take consecutive byte ranges, so a key's masks are one contiguous stretch of the stream in memory
// order; a u64 is 8 consecutive bytes, little-endian (core_crypto/commons/math/random/uniform.rs:15-24)).
// Decompression (core_crypto/algorithms/seeded_lwe_bootstrap_key_decompression.rs,
// seeded_lwe_keyswitch_key_decompression.rs and the list / GGSW / GLWE functions they call) is therefore a pure
// data-parallel fill: `units` ciphertexts of mask_words stream words followed by body_words stored words --
// 85 MB of keystream for the 2_2 keys, generated straight into HBM instead of crossing PCIe.
//
// Kernel: one thread = 4 consecutive AES blocks (+ the first byte of the fifth: stream words straddle the blocks
// because of the one-byte offset) = 8 output words.  AES as one 256-entry T-table in LDS (the other three are its
// byte rotations), 16 lookups per round; integer / byte work bound by LDS lookups and the 8-byte scattered stores,
// no MFMA.
#include <hip/hip_runtime.h>

#include <cstring>

#include "kernels.hpp"

namespace tfhe_hip {

namespace {

// GF(2^8) by log / antilog tables over the generator 3 (FIPS-197 4.2; the S-box is the inverse followed by the
// affine map of 5.1.1)
struct AesTables {
    uint8_t sbox[256];
    uint32_t te0[256];
    AesTables()
    {
        uint8_t exp[256], log[256];
        uint8_t x = 1;
        for (int i = 0; i < 255; i++) {
            exp[i] = x;
            log[x] = (uint8_t)i;
            x = (uint8_t)(x ^ (uint8_t)(x << 1) ^ ((x & 0x80) ? 0x1b : 0)); // x * 3
        }
        exp[255] = exp[0];
        log[0] = 0;
        for (int v = 0; v < 256; v++) {
            const uint8_t inv = v ? exp[(255 - log[v]) % 255] : 0;
            uint8_t s = 0x63;
            for (int b = 0; b < 8; b++) {
                const int bit = ((inv >> b) ^ (inv >> ((b + 4) & 7)) ^ (inv >> ((b + 5) & 7)) ^ (inv >> ((b + 6) & 7)) ^
                                 (inv >> ((b + 7) & 7))) & 1;
                s ^= (uint8_t)(bit << b);
            }
            sbox[v] = s;
        }
        for (int v = 0; v < 256; v++) {
            const uint32_t s = sbox[v];
            const uint32_t s2 = ((s << 1) ^ ((s & 0x80) ? 0x11b : 0)) & 0xff;
            te0[v] = s2 | (s << 8) | (s << 16) | ((s2 ^ s) << 24); // row-0 byte through MixColumns: (2s, s, s, 3s)
        }
    }
};
const AesTables &tables()
{
    static const AesTables t;
    return t;
}

} // namespace

// FIPS-197 5.2: the 11 round keys as 44 little-endian words
void aes128_round_keys(const uint8_t key[16], uint32_t rk[44])
{
    const AesTables &t = tables();
    uint8_t w[176];
    std::memcpy(w, key, 16);
    uint8_t rcon = 1;
    for (int i = 16; i < 176; i += 4) {
        uint8_t a[4] = {w[i - 4], w[i - 3], w[i - 2], w[i - 1]};
        if (i % 16 == 0) {
            const uint8_t a0 = a[0];
            a[0] = (uint8_t)(t.sbox[a[1]] ^ rcon);
            a[1] = t.sbox[a[2]];
            a[2] = t.sbox[a[3]];
            a[3] = t.sbox[a0];
            rcon = (uint8_t)((rcon << 1) ^ ((rcon & 0x80) ? 0x1b : 0));
        }
        for (int j = 0; j < 4; j++) w[i + j] = (uint8_t)(w[i - 16 + j] ^ a[j]);
    }
    for (int i = 0; i < 44; i++)
        rk[i] = (uint32_t)w[4 * i] | ((uint32_t)w[4 * i + 1] << 8) | ((uint32_t)w[4 * i + 2] << 16) | ((uint32_t)w[4 * i + 3] << 24);
}

struct RoundKeys {
    uint32_t w[44];
};

namespace {

constexpr int BLOCKS_PER_THREAD = 4;

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// AES-128 of the 128-bit little-endian counter `ctr` (< 2^64); state column c = bytes 4c .. 4c + 3, row 0 lowest
__device__ __forceinline__ void aes_block(const uint32_t *T, const RoundKeys &rk, uint64_t ctr, uint32_t out[4])
{
    uint32_t s0 = (uint32_t)ctr ^ rk.w[0], s1 = (uint32_t)(ctr >> 32) ^ rk.w[1], s2 = rk.w[2], s3 = rk.w[3];
#pragma unroll
    for (int r = 1; r < 10; r++) {
        const uint32_t t0 = T[s0 & 255] ^ rotl32(T[(s1 >> 8) & 255], 8) ^ rotl32(T[(s2 >> 16) & 255], 16) ^ rotl32(T[s3 >> 24], 24) ^ rk.w[4 * r];
        const uint32_t t1 = T[s1 & 255] ^ rotl32(T[(s2 >> 8) & 255], 8) ^ rotl32(T[(s3 >> 16) & 255], 16) ^ rotl32(T[s0 >> 24], 24) ^ rk.w[4 * r + 1];
        const uint32_t t2 = T[s2 & 255] ^ rotl32(T[(s3 >> 8) & 255], 8) ^ rotl32(T[(s0 >> 16) & 255], 16) ^ rotl32(T[s1 >> 24], 24) ^ rk.w[4 * r + 2];
        const uint32_t t3 = T[s3 & 255] ^ rotl32(T[(s0 >> 8) & 255], 8) ^ rotl32(T[(s1 >> 16) & 255], 16) ^ rotl32(T[s2 >> 24], 24) ^ rk.w[4 * r + 3];
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
    }
    // last round: SubBytes (byte 1 of a table entry is the S-box value) and ShiftRows only
#define SB(x) ((T[(x) & 255] >> 8) & 255u)
    out[0] = (SB(s0) | (SB(s1 >> 8) << 8) | (SB(s2 >> 16) << 16) | (SB(s3 >> 24) << 24)) ^ rk.w[40];
    out[1] = (SB(s1) | (SB(s2 >> 8) << 8) | (SB(s3 >> 16) << 16) | (SB(s0 >> 24) << 24)) ^ rk.w[41];
    out[2] = (SB(s2) | (SB(s3 >> 8) << 8) | (SB(s0 >> 16) << 16) | (SB(s1 >> 24) << 24)) ^ rk.w[42];
    out[3] = (SB(s3) | (SB(s0 >> 8) << 8) | (SB(s1 >> 16) << 16) | (SB(s2 >> 24) << 24)) ^ rk.w[43];
#undef SB
}

// generator words [first_word, first_word + count) -> dst[(w / mask_words) * unit_words + w % mask_words] (w counted
// from first_word's unit 0); mask_words == unit_words writes the plain stream
__global__ __launch_bounds__(256) void csprng_fill_kernel(const uint32_t *__restrict__ te0, const RoundKeys rk, uint64_t first_word,
                                                          uint64_t count, uint64_t mask_words, uint64_t unit_words,
                                                          uint64_t *__restrict__ dst)
{
    __shared__ uint32_t T[256];
    T[threadIdx.x] = te0[threadIdx.x];
    __syncthreads();
    // thread g covers stream words [2 G g, 2 G (g + 1)) relative to the even word at or below first_word
    const uint64_t base_word = first_word & ~(uint64_t)1;
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t w0 = base_word + g * (2 * BLOCKS_PER_THREAD);
    if (w0 >= first_word + count) return;
    uint32_t b[BLOCKS_PER_THREAD + 1][4];
#pragma unroll
    for (int i = 0; i <= BLOCKS_PER_THREAD; i++) aes_block(T, rk, w0 / 2 + i, b[i]);
#pragma unroll
    for (int i = 0; i < BLOCKS_PER_THREAD; i++) {
        // stream word 2 blk = bytes 1..8 of the block, word 2 blk + 1 = bytes 9..15 and byte 0 of the next block
        const uint64_t lo = (uint64_t)b[i][0] | ((uint64_t)b[i][1] << 32), hi = (uint64_t)b[i][2] | ((uint64_t)b[i][3] << 32);
        const uint64_t even = (lo >> 8) | (hi << 56);
        const uint64_t odd = (hi >> 8) | ((uint64_t)(b[i + 1][0] & 255u) << 56);
        const uint64_t we = w0 + 2 * i;
        if (we >= first_word && we < first_word + count) {
            const uint64_t r = we - first_word;
            dst[(r / mask_words) * unit_words + r % mask_words] = even;
        }
        if (we + 1 >= first_word && we + 1 < first_word + count) {
            const uint64_t r = we + 1 - first_word;
            dst[(r / mask_words) * unit_words + r % mask_words] = odd;
        }
    }
}

// One generator PER UNIT (compressed ciphertexts: every seeded LWE ciphertext has its own seed): unit u = blockIdx.y,
// its round keys at rks + 44 u, its first mask_words generator words -> dst[u * unit_words + ..], then its body
__global__ __launch_bounds__(256) void csprng_fill_units_kernel(const uint32_t *__restrict__ te0, const uint32_t *__restrict__ rks,
                                                                const uint64_t *__restrict__ bodies, uint64_t mask_words,
                                                                uint64_t *__restrict__ dst)
{
    __shared__ uint32_t T[256];
    __shared__ RoundKeys rk;
    const uint64_t u = blockIdx.y, unit_words = mask_words + 1;
    T[threadIdx.x] = te0[threadIdx.x];
    if (threadIdx.x < 44) rk.w[threadIdx.x] = rks[u * 44 + threadIdx.x];
    __syncthreads();
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t w0 = g * (2 * BLOCKS_PER_THREAD);
    if (g == 0) dst[u * unit_words + mask_words] = bodies[u];
    if (w0 >= mask_words) return;
    uint32_t b[BLOCKS_PER_THREAD + 1][4];
#pragma unroll
    for (int i = 0; i <= BLOCKS_PER_THREAD; i++) aes_block(T, rk, w0 / 2 + i, b[i]);
#pragma unroll
    for (int i = 0; i < BLOCKS_PER_THREAD; i++) {
        const uint64_t lo = (uint64_t)b[i][0] | ((uint64_t)b[i][1] << 32), hi = (uint64_t)b[i][2] | ((uint64_t)b[i][3] << 32);
        const uint64_t we = w0 + 2 * i;
        if (we < mask_words) dst[u * unit_words + we] = (lo >> 8) | (hi << 56);
        if (we + 1 < mask_words) dst[u * unit_words + we + 1] = (hi >> 8) | ((uint64_t)(b[i + 1][0] & 255u) << 56);
    }
}

// dst[u * unit_words + mask_words + j] = bodies[u * body_words + j]
__global__ __launch_bounds__(256) void seeded_bodies_kernel(const uint64_t *__restrict__ bodies, uint64_t total, uint64_t body_words,
                                                            uint64_t mask_words, uint64_t *__restrict__ dst)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t u = i / body_words, j = i % body_words;
    dst[u * (mask_words + body_words) + mask_words + j] = bodies[i];
}

} // namespace

hipError_t upload_aes_table(uint32_t **d_te0)
{
    hipError_t e = hipMalloc((void **)d_te0, 256 * sizeof(uint32_t));
    if (e != hipSuccess) return e;
    return hipMemcpy(*d_te0, tables().te0, 256 * sizeof(uint32_t), hipMemcpyHostToDevice);
}

hipError_t launch_csprng_fill(const uint32_t *d_te0, const uint8_t seed[16], uint64_t first_word, uint64_t count, uint64_t mask_words,
                              uint64_t unit_words, uint64_t *d_dst, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    if (mask_words == 0 || unit_words < mask_words) return hipErrorInvalidValue;
    RoundKeys rk;
    aes128_round_keys(seed, rk.w);
    const uint64_t base_word = first_word & ~(uint64_t)1;
    const uint64_t span = first_word + count - base_word;
    const uint64_t threads = (span + 2 * BLOCKS_PER_THREAD - 1) / (2 * BLOCKS_PER_THREAD);
    const uint64_t grid = (threads + 255) / 256;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    csprng_fill_kernel<<<(unsigned)grid, 256, 0, s>>>(d_te0, rk, first_word, count, mask_words, unit_words, d_dst);
    return hipGetLastError();
}

// decompress_seeded_lwe_ciphertext for `units` ciphertexts with one seed each: d_rks = [units][44] round-key words
hipError_t launch_seeded_expand_units(const uint32_t *d_te0, const uint32_t *d_rks, const uint64_t *d_bodies, uint64_t units,
                                      uint64_t mask_words, uint64_t *d_dst, hipStream_t s)
{
    if (units == 0) return hipSuccess;
    if (mask_words == 0 || units > 65535) return hipErrorInvalidValue;
    const uint64_t threads = (mask_words + 2 * BLOCKS_PER_THREAD - 1) / (2 * BLOCKS_PER_THREAD);
    const dim3 grid((unsigned)((threads + 255) / 256), (unsigned)units);
    csprng_fill_units_kernel<<<grid, 256, 0, s>>>(d_te0, d_rks, d_bodies, mask_words, d_dst);
    return hipGetLastError();
}

// seeded_*_decompression: `units` ciphertexts of mask_words generator words then body_words stored words
hipError_t launch_seeded_expand(const uint32_t *d_te0, const uint8_t seed[16], const uint64_t *d_bodies, uint64_t units, uint64_t mask_words,
                                uint64_t body_words, uint64_t *d_dst, hipStream_t s)
{
    hipError_t e = launch_csprng_fill(d_te0, seed, 0, units * mask_words, mask_words, mask_words + body_words, d_dst, s);
    if (e != hipSuccess) return e;
    const uint64_t total = units * body_words;
    if (total == 0) return hipSuccess;
    seeded_bodies_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(d_bodies, total, body_words, mask_words, d_dst);
    return hipGetLastError();
}

} // namespace tfhe_hip
