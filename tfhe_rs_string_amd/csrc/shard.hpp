// shard.hpp -- how a batch of independent ciphertexts is split over the devices of a multi-device context
// (shared by api.cpp and the CPU test backend, tests/emu/oracle_backend.cpp, so that both shard identically).
// The reference's counterpart is rayon's par_iter over blocks (tfhe/src/integer/server_key/radix_parallel/comparison.rs:22-28):
// independent units, no exchange except handing the results back.
#pragma once
#include <stddef.h>

namespace tfhe_hip {

// contiguous shard [*begin, *end) of `total` units for `rank` of `world`: ceil split, the first ranks take one more
inline void shard_range(size_t total, size_t rank, size_t world, size_t *begin, size_t *end)
{
    const size_t base = total / world, extra = total % world;
    *begin = rank * base + (rank < extra ? rank : extra);
    *end = *begin + base + (rank < extra ? 1 : 0);
}

// devices that take part in a batch of B: a device is only worth its transfers for at least `min_per_device` ciphertexts
inline size_t active_devices(size_t B, size_t n_devices, size_t min_per_device)
{
    if (n_devices <= 1 || min_per_device == 0) return n_devices ? n_devices : 1;
    const size_t fit = B / min_per_device;
    return fit < 1 ? 1 : (fit < n_devices ? fit : n_devices);
}

} // namespace tfhe_hip
