/*
 * select.h -- the probe kernels are instantiated in their own translation
 * units (probe_tu.hip, one per kernel variant and workgroup size, so that
 * `make -j` compiles them in parallel); these functions hand out the entry
 * points.
 */
#ifndef COMPAIRR_AMD_SELECT_H
#define COMPAIRR_AMD_SELECT_H

#include "layout.h"

namespace cmpr {

using ProbeFn = void (*)(const ProbeParams);

/* variant 0 (kernels.h probe_kernel) */
ProbeFn select_probe_v0(int A, int D, bool indels, bool genes);
/* variant 1 (kernels_sliced.h probe_sliced_kernel), NW waves per workgroup */
ProbeFn select_probe_v1_nw4(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v1_nw8(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v1_nw16(int A, int D, bool indels, bool genes);
/* variant 2 (kernels_rows.h probe_rows_kernel): the fast form, and the form that
   resolves its Bloom positives inline (deferred_resolve = 0, redo pass) */
ProbeFn select_probe_v2_nw4(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v2_nw8(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v2_nw16(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v2_inline_nw4(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v2_inline_nw8(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v2_inline_nw16(int A, int D, bool indels, bool genes);
/* ... for layouts with four amino-acid class residues (nullptr for anything else) */
ProbeFn select_probe_v2_wide_nw4(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v2_wide_nw8(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v2_wide_nw16(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v2_wide_inline_nw4(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v2_wide_inline_nw8(int A, int D, bool indels, bool genes);
ProbeFn select_probe_v2_wide_inline_nw16(int A, int D, bool indels, bool genes);
/* nucleotides, d = 2 on pair rows (kernels_pairs2.h) */
ProbeFn select_probe_pairs2(bool genes);
/* resolve_kernel (kernels.h) */
ProbeFn select_resolve(bool genes);

}  // namespace cmpr
#endif
