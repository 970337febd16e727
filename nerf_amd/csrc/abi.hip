// Library-wide pieces of the C ABI: version and the thread-local error string.
#include "common.h"

namespace nerf {
char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
__global__ void zero_words_kernel(unsigned* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
}  // namespace nerf

extern "C" int nerf_abi_version(void) { return NERF_ABI_VERSION; }
extern "C" const char* nerf_last_error(void) { return nerf::err_buf(); }
