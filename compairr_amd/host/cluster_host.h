/*
 * cluster_host.h -- the -c / --cluster command: single-linkage clustering of
 * one repertoire file.  The neighbour search ("Building network",
 * /root/reference/src/cluster.cc:79-163, 225-274) is the same per-query loop
 * as --matrix with the set compared against itself and seed != hit
 * (cluster.cc:105), so it runs on the overlap backend in pairs mode; the
 * sweep over the network (cluster.cc:276-417) and the output
 * (cluster.cc:419-452) are host code.
 */
#ifndef COMPAIRR_AMD_CLUSTER_HOST_H
#define COMPAIRR_AMD_CLUSTER_HOST_H

#include <cstdio>

#include "overlap_host.h"

namespace cmprhost {

/* Runs the command; returns the exit status. */
int run_cluster(const Options &o, OverlapBackend &backend, FILE *log, FILE *out);

}  // namespace cmprhost
#endif
