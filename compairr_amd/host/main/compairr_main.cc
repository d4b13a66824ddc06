/*
 * compairr_main.cc -- the `compairr` program of the MI355X build: the
 * reference's command line and AIRR-TSV I/O for --matrix
 * (/root/reference/src/compairr.cc:743-798) with the per-query loop on the GPU.
 */
#include <memory>

#include "hip_backend.h"

int main(int argc, char **argv)
{
  std::string error;
  std::unique_ptr<cmprhost::OverlapBackend> backend(
      cmprhost::make_hip_backend(argv[0], error));
  if (!backend)
    cmprhost::fatal(error.c_str());
  return cmprhost::compairr_main(argc, argv, *backend);
}
