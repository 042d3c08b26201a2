// device_guard.h -- makes `device` current for the scope of a C-ABI call and puts the caller's device back afterwards
// (a host process with several GPUs, e.g. torch, must not find its current device changed by this library).
#pragma once
#include <hip/hip_runtime.h>

namespace dvbs2 {

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess; else prev = -1; // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

} // namespace dvbs2
