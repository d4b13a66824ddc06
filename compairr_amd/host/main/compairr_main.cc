/*
 * compairr_main.cc -- the `compairr` program of the MI355X build: the
 * reference's command line and AIRR-TSV I/O for --matrix
 * (/root/reference/src/compairr.cc:743-798) with the per-query loop on the GPU.
 */
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <memory>

#include "hip_backend.h"

int main(int argc, char **argv)
{
  std::string error;
  std::unique_ptr<cmprhost::OverlapBackend> backend(
      cmprhost::make_hip_backend(argv[0], error));
  if (!backend)
    cmprhost::fatal(error.c_str());
  const int rc = cmprhost::compairr_main(argc, argv, *backend);
  /* Everything is written and closed (compairr_main closes its files).  Leaving through _exit skips
     what only costs time now: unloading the HIP runtime and its device context (~0.15 s), freeing
     gigabytes of vectors page by page. */
  fflush(NULL);
  if (getenv("COMPAIRR_HOST_TIMING"))
    fprintf(stderr, "[host] leaving\n");
  _exit(rc);
}
