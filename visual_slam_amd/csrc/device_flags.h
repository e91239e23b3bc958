// Per-device "done once" flags for launchers.
#pragma once
#include <hip/hip_runtime.h>

namespace vsm {

// Function attributes (dynamic LDS size) are set once per device a kernel is launched on.
struct DeviceFlags {
    bool seen[64] = {};
};
inline bool first_launch_on_device(DeviceFlags& f)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (f.seen[dev]) return false;
    f.seen[dev] = true;
    return true;
}

}  // namespace vsm
