// Fixed-base scalar multiplication by the group generators (util.hip): [k_i] G1 / [k_i] G2 for many scalars -- kzg.NewSRS, groth16.Setup, synthetic bases.
#pragma once
#include "ctx.hpp"
#include "curve.hpp"

namespace zkmi {

// out[i] = [scalars[i]] G1 (is_g2 = 0, 64-byte affine points) or G2 (128-byte); scalars: Montgomery fr.Elements in HBM
int fixed_base_mul_scalars(Slot* s, hipStream_t st, int is_g2, const Fr* d_scalars, size_t n, void* d_out);
Affine<Fp> generator_g1();
Affine<Fp2> generator_g2();

}  // namespace zkmi
