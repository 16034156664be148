// numa_placement.h — keeps a file thread and its page-locked block buffers on the host NUMA node of its GPU.
//
// On an 8 x MI355X node every GPU hangs off one socket's PCIe root; PCM that crosses the socket interconnect on
// its way to the bus costs bandwidth the 8 GPUs share.  With placement on (SetNumaPlacement / FOLVE_AMD_NUMA=1),
// SoundProcessor allocates its page-locked ring while the calling thread is moved next to the GPU the router
// picked (page-locked pages are placed where the allocating thread runs), and a host may pin its file threads
// the same way (PinThreadNearDevice).  Everything here is best effort: without sysfs information, or inside a
// cpuset that excludes the GPU's CPUs, nothing is changed.
#pragma once

#include <sched.h>

namespace folve {

void SetNumaPlacement(bool on);
bool NumaPlacement();
// CPUs local to HIP device `device` (sysfs local_cpulist of its PCI function), intersected with the CPUs this
// thread may run on.  False if unknown or empty.
bool DeviceLocalCpus(int device, cpu_set_t* out);
// Move the calling thread onto the device's local CPUs for good.
bool PinThreadNearDevice(int device);
// ... or for the lifetime of the object (restores the previous affinity).
class ScopedDeviceAffinity {
public:
    explicit ScopedDeviceAffinity(int device);
    ~ScopedDeviceAffinity();
private:
    bool moved_;
    cpu_set_t saved_;
};

}  // namespace folve
