/*
 * hip_backend.h -- the product backend: binds the C ABI of
 * libcompairr_hip.so (include/compairr_hip.h) at run time and runs the
 * per-query loop on the MI355X.  There is no CPU fallback: if the library or
 * a HIP device is missing, construction fails and the program exits with the
 * reference's error convention.
 */
#ifndef COMPAIRR_AMD_HIP_BACKEND_H
#define COMPAIRR_AMD_HIP_BACKEND_H

#include "overlap_host.h"

namespace cmprhost {

/* argv0 is used to find <dir of executable>/../compairr_amd/lib/; the
   environment variable COMPAIRR_HIP_LIB overrides the search. */
OverlapBackend *make_hip_backend(const char *argv0, std::string &error);

}  // namespace cmprhost
#endif
