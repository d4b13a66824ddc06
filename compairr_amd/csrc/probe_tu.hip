/*
 * probe_tu.hip -- one translation unit per (kernel variant, workgroup size):
 * compiled with -DTU_VARIANT=0|1|2|9, -DTU_NW=4|8|16 (variants 1, 2) and
 * -DTU_INLINE=0|1 (variant 2: the fast form / the form that resolves inline); -DTU_WIDE: variant 2 for
 * four amino-acid class residues (fast and inline forms).
 * TU_VARIANT 9 = resolve_kernel, 3 = probe_pairs2_kernel.
 */
#include "select.h"
#if TU_VARIANT == 3
#include "kernels_pairs2.h"
#elif TU_VARIANT == 2
#include "kernels_rows.h"
#elif TU_VARIANT == 1
#include "kernels_sliced.h"
#else
#include "kernels.h"
#endif

namespace cmpr {

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#if TU_VARIANT == 0
#define KERNEL(A_, D_, I_, G_) probe_kernel<A_, D_, I_, G_>
#define SELECT_NAME select_probe_v0
#elif TU_VARIANT == 1
#define KERNEL(A_, D_, I_, G_) probe_sliced_kernel<A_, D_, I_, G_, TU_NW>
#define SELECT_NAME CAT(select_probe_v1_nw, TU_NW)
#elif TU_VARIANT == 2 && defined(TU_WIDE) && TU_INLINE
#define KERNEL(A_, D_, I_, G_) probe_rows_kernel<A_, D_, I_, G_, TU_NW, true, true>
#define SELECT_NAME CAT(select_probe_v2_wide_inline_nw, TU_NW)
#elif TU_VARIANT == 2 && defined(TU_WIDE)
#define KERNEL(A_, D_, I_, G_) probe_rows_kernel<A_, D_, I_, G_, TU_NW, false, true>
#define SELECT_NAME CAT(select_probe_v2_wide_nw, TU_NW)
#elif TU_VARIANT == 2 && TU_INLINE
#define KERNEL(A_, D_, I_, G_) probe_rows_kernel<A_, D_, I_, G_, TU_NW, true>
#define SELECT_NAME CAT(select_probe_v2_inline_nw, TU_NW)
#elif TU_VARIANT == 2
#define KERNEL(A_, D_, I_, G_) probe_rows_kernel<A_, D_, I_, G_, TU_NW, false>
#define SELECT_NAME CAT(select_probe_v2_nw, TU_NW)
#endif

#if TU_VARIANT == 9
ProbeFn select_resolve(bool genes)
{
  return genes ? (ProbeFn)resolve_kernel<true> : (ProbeFn)resolve_kernel<false>;
}
#elif TU_VARIANT == 3
/* nucleotides, d = 2, pair rows (kernels_pairs2.h): 16 waves per workgroup */
ProbeFn select_probe_pairs2(bool genes)
{
  return genes ? (ProbeFn)probe_pairs2_kernel<true, 16> : (ProbeFn)probe_pairs2_kernel<false, 16>;
}
#else
ProbeFn SELECT_NAME(int A, int D, bool indels, bool genes)
{
#define PICK(A_, D_, I_) (genes ? (ProbeFn)KERNEL(A_, D_, I_, true) : (ProbeFn)KERNEL(A_, D_, I_, false))
#if defined(TU_WIDE)
  /* four class residues: amino acids at d = 1 only (layout.h kernel_class_res, ref_index.hip) */
  if (A != 20 || D != 1)
    return nullptr;
  return indels ? PICK(20, 1, true) : PICK(20, 1, false);
#else
  if (A == 20) {
    if (D == 0) return PICK(20, 0, false);
    if (D == 1) return indels ? PICK(20, 1, true) : PICK(20, 1, false);
    return PICK(20, 2, false);
  }
  if (D == 0) return PICK(4, 0, false);
  if (D == 1) return indels ? PICK(4, 1, true) : PICK(4, 1, false);
  return PICK(4, 2, false);
#endif
#undef PICK
}
#endif

}  // namespace cmpr
