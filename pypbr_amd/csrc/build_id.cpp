// build_id.cpp -- which sources this libpbr_hip.so was built from: the Makefile hashes every file that goes into the library
// (csrc/*.hip, csrc/*.hpp, include/pbr_hip.h, the Makefile itself) and compiles the digest in.  Evidence collected with a library
// that is older than the sources next to it is how round 3 ended up with a committed rocprof summary of a binary that was not the one
// shipped; tools/collect_round4.sh and bench.py compare this with the digest of the sources they find (pypbr_amd/_native.py: build_stamp).
#include "../../include/pbr_hip.h"

#ifndef PBR_SOURCE_HASH
#define PBR_SOURCE_HASH "unknown"
#endif

extern "C" const char *pbr_build_id(void) { return PBR_SOURCE_HASH; }
