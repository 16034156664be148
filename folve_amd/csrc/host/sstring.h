// sstring.h — quoted/escaped token scanner used by the filter config loader.
// Behavioural mirror of /root/reference/zita-sstring.h:26-43 (spec) and
// zita-sstring.cc:32-116: same consumed-character counts, same error cases.
#pragma once

namespace folve {

// Scans `srce` for a possibly quoted string into `dest` (at most size-1 chars
// plus the terminator).  Returns the number of characters consumed, 0 on error.
int sstring(const char* srce, char* dest, int size);

}  // namespace folve
