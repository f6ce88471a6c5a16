// abi_common.hpp -- what every translation unit of libburgers_hip.so shares at the C-ABI boundary:
// the per-thread record of the last failed launch (read back by bg_last_hip_error) and the
// cached per-device CU count.  No other process-wide state exists in the library.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/burgers_hip.h"

namespace bg {

extern thread_local int tls_last_hip_error;     // defined in fom.hip

// Call right after a kernel launch: BG_OK, or BG_ERR_LAUNCH with the hipError_t kept for bg_last_hip_error().
inline int check_launch()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return BG_OK;
    tls_last_hip_error = (int)e;
    return BG_ERR_LAUNCH;
}

// Compute units of the current device, queried once per device (a launch must not pay two runtime calls).
inline int device_cu_count()
{
    static std::atomic<int> cached[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    int c = cached[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cached[dev].store(c, std::memory_order_relaxed);
    return c;
}

}  // namespace bg
