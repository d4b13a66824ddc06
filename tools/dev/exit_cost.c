/* tools/dev/exit_cost.c -- what leaving a process costs after it has touched N GiB of anonymous memory, with 4-KiB
 * pages and with transparent huge pages (madvise): usage: exit_cost <GiB> <0|1 huge> <threads>
 * (the host program of bin/compairr holds ~2.5 GiB of file text and parsed vectors when it leaves) */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
struct part { char *p; size_t n; };
static void *touch(void *a) { struct part *q = a; memset(q->p, 1, q->n); return 0; }
int main(int argc, char **argv)
{
  const double gib = argc > 1 ? atof(argv[1]) : 2.5;
  const int huge = argc > 2 ? atoi(argv[2]) : 0, nt = argc > 3 ? atoi(argv[3]) : 16;
  const size_t n = (size_t)(gib * (1u << 30)) & ~((size_t)(2u << 20) - 1);
  char *p = mmap(0, n + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) return 2;
  p = (char *)(((size_t)p + (2u << 20) - 1) & ~((size_t)(2u << 20) - 1));
  if (huge) madvise(p, n, MADV_HUGEPAGE);
  double t0 = now();
  pthread_t th[256]; struct part parts[256];
  for (int i = 0; i < nt; i++) { parts[i].p = p + n / nt * i; parts[i].n = n / nt; pthread_create(&th[i], 0, touch, &parts[i]); }
  for (int i = 0; i < nt; i++) pthread_join(th[i], 0);
  fprintf(stderr, "touched %.2f GiB huge=%d threads=%d in %.1f ms; leaving at %.3f\n", gib, huge, nt, (now() - t0) * 1e3, now());
  _exit(0);
}
