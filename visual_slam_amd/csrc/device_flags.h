// Per-device "done once" flags for launchers.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace vsm {

// A kernel that takes more dynamic LDS than the default limit needs the limit raised once per device it
// is launched on.  The flag is set after the attribute, so a second thread never launches in between
// (two threads may both set the attribute: harmless).
struct DeviceFlags {
    std::atomic<bool> seen[64] = {};
};
inline void ensure_dynamic_lds(DeviceFlags& f, const void* kernel, int bytes)
{
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (known && f.seen[dev].load(std::memory_order_acquire)) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (known) f.seen[dev].store(true, std::memory_order_release);
}

}  // namespace vsm
