// Build fingerprint of libwtpse_hip.so (see wtpse_hip/build.py::source_hash and include/wtpse_hip.h).
#ifndef WTPSE_SRC_HASH
#define WTPSE_SRC_HASH "unstamped"
#endif
extern "C" const char* wtpse_source_hash(void) { return WTPSE_SRC_HASH; }
