// sound-processor.h — put this directory in front of folve's include path and the
// reference's own `#include "sound-processor.h"` (convolve-file-handler.h, folve-filesystem.h)
// resolves to the GPU-backed class, in the global namespace as the reference declares it
// (/root/reference/sound-processor.h:28).
#ifndef FOLVE_SOUND_PROCESSOR_H
#define FOLVE_SOUND_PROCESSOR_H
#include "../../folve_amd/csrc/host/sound_processor.h"
using folve::SoundProcessor;
#endif  // FOLVE_SOUND_PROCESSOR_H
