// build_id.cpp -- which build of the library this is.  The Makefile hashes every source and header of the library and
// the compiler flags into ASORA_BUILD_ID; counter summaries under profiles/ carry the id of the library they were
// collected on, and bench.py pairs its timings only with a summary whose id matches (include/asora_hip.h).
#ifndef ASORA_BUILD_ID
#error "ASORA_BUILD_ID must come from the Makefile"
#endif
#ifndef ASORA_BUILD_FLAGS
#define ASORA_BUILD_FLAGS ""
#endif
extern "C" {
const char *asora_build_id(void) { return ASORA_BUILD_ID; }
const char *asora_build_flags(void) { return ASORA_BUILD_FLAGS; }
}
