// The KZG SRS in Lagrange form over a domain (lagrange.hip): what lets plonk.Prove commit l, r, o from wire values instead of coefficients.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace zkmi {
// d_srs: at least 2^logn + 2 affine G1 points [tau^j] in HBM -> *handle: 2^logn + 2 registered bases: [L_i(tau)] for i < n, then [tau^n - 1], [tau^(n+1) - tau]
int lagrange_srs_build(const void* d_srs, size_t srs_n, unsigned logn, uint64_t* handle);
}  // namespace zkmi
