"""folve_amd — MI355X (gfx950) convolution engine behind folve's SoundProcessor seam.

Python is only the test/bench harness here: the product is the C-ABI shared
library ``libfolve_amd.so`` (include/folve_engine.h, include/folve_host.h).
Importing this package loads that library and fails loudly when it is missing;
there is no CPU fallback anywhere in the package.
"""
from .capi import (  # noqa: F401
    FolveError, lib, lib_path, build_library, fragm_for_size, device_count,
    Engine, Filter, Stream, batch_process,
    FE_HOST_PTRS, FE_DEVICE_PTRS, FE_ASYNC,
)
