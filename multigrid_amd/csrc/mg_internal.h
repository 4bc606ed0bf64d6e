// Shared by the translation units of libmg_hip that implement the C ABI (mg_api.hip,
// mg_dist.hip).  Not part of the public boundary.
#pragma once

namespace mgk {

// Records the calling thread's error text (what mg_last_error() returns) and returns `code`.
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Device bytes the rank `rank` of `world` allocates for a partitioned solve of the global grid n
// with `depth` coarse grids and nu[l] sweeps on level l (its slabs of every distributed level with
// their halo planes, the whole small levels, the partial sums); < 0 with the error set when the
// grid cannot be partitioned that way.
long long dist_rank_device_bytes(const long long n[3], int depth, int world, int coarse_cut, int rank,
                                 const int *nu);

}  // namespace mgk
