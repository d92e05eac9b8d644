// Internal interface of the NTT module (ntt.hip) used by the Groth16 prover (groth16.hip).
#pragma once
#include "ctx.hpp"
#include "ff.hpp"
#include "host_ff.hpp"

namespace zkmi {

enum : unsigned {
    DOM_TW = 1,               // Twiddles:  w^i, i < N/2
    DOM_TW_INV = 2,           // TwiddlesInv
    DOM_COSET = 4,            // CosetTable[i] = g^i
    DOM_COSET_REV = 8,        // CosetTableReversed[i] = g^bitrev(i)
    DOM_COSET_INV_N = 16,     // CosetTableInv[i] * CardinalityInv
    DOM_COSET_INV_N_REV = 32, // CosetTableInvReversed[i] * CardinalityInv
    DOM_COSET_REV_N = 64      // CosetTableReversed[i] * CardinalityInv  (computeH: inverse-then-coset-forward fusion)
};

// Device-resident restatement of gnark-crypto's fft.Domain (tables are built lazily, once per size, and shared).
struct Domain {
    unsigned logn = 0;
    HFr gen, gen_inv, card_inv, coset, coset_inv;
    Fr *tw29 = nullptr, *tw29_inv = nullptr;  // the twiddle tables again as w * 2^261 mod r: multiplier form of the 29-bit-limb butterflies
    Fr *tw = nullptr, *tw_inv = nullptr, *coset_tab = nullptr, *coset_rev = nullptr, *coset_inv_n = nullptr,
       *coset_inv_n_rev = nullptr, *coset_rev_n = nullptr;
};

int get_domain(Slot* s, hipStream_t st, unsigned logn, unsigned need, Domain** out);
int ntt_dev(Slot* s, hipStream_t st, Fr* d_a, unsigned logn, int inverse, int decimation, int coset);
int bit_reverse_dev(Slot* s, hipStream_t st, Fr* d_a, unsigned logn);
int compute_h_inplace(Slot* s, hipStream_t st, Fr* a, Fr* b, Fr* c, unsigned logN, const Fr* const* src = nullptr, const hipStream_t* side = nullptr);
int fr_mul_dev(Slot* s, hipStream_t st, Fr* out, const Fr* a, const Fr* b, size_t n);

}  // namespace zkmi
