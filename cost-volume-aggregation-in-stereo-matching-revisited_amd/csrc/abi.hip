// ABI version of libdca_hip.so: bumped whenever an entry point of include/dca_hip.h changes its argument list, so a
// stale binary (the .so is git-ignored and shipped separately) is refused by the ctypes loader instead of being called
// with the wrong arguments.
#include "../../include/dca_hip.h"

extern "C" int dca_abi_version(void) { return DCA_ABI_VERSION; }
