// processor-pool.h — as sound-processor.h beside it: the reference's
// `#include "processor-pool.h"` (/root/reference/processor-pool.h:30) gets the GPU-sharding pool.
#ifndef FOLVE_PROCESSOR_POOL_
#define FOLVE_PROCESSOR_POOL_
#include "../../folve_amd/csrc/host/processor_pool.h"
using folve::ProcessorPool;
#endif  // FOLVE_PROCESSOR_POOL_
