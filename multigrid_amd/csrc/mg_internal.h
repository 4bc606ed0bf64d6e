// Shared by the translation units of libmg_hip that implement the C ABI (mg_api.hip,
// mg_dist.hip).  Not part of the public boundary.
#pragma once

namespace mgk {

// Records the calling thread's error text (what mg_last_error() returns) and returns `code`.
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace mgk
