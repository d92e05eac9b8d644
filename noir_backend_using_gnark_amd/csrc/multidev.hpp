// Internal interface of multidev.hip: the entry points of msm.hip / groth16.hip / ntt.hip hand over to these when a call names (or defaults to) several
// device entries, or when a handle is a COMPOSITE one (top byte 0xff, first entry in the next byte: ctx.hpp hentry()).
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/zkmi.h"

namespace zkmi {

static inline bool md_is_composite(uint64_t handle) { return (handle >> 56) == 0xff; }
void md_set_default_mask(uint32_t mask);
uint32_t md_default_mask();
// requested: a device_mask field (0 = the process default).  Implicit masks only spread work of at least min_units_per_entry per entry.
int md_entries_for(uint32_t requested, size_t units, size_t min_units_per_entry, std::vector<int>* out);

int bases_register_on_this_entry(const void* points, size_t n, int is_g2, int on_device, int table_c, uint64_t* handle);  // msm.hip: no spreading
int md_msm_host(int g2, const void* points, const zk_fr* scalars, size_t n, const zk_msm_cfg* cfg, void* out, const std::vector<int>& entries);
int md_bases_register(const void* points, size_t n, int is_g2, int on_device, int table_bits, const std::vector<int>& entries, uint64_t* handle);
int md_bases_build_table(uint64_t h, int table_bits);
int md_bases_info(uint64_t h, size_t* n, int* is_g2);
int md_bases_free(uint64_t h);
int md_msm_bases(uint64_t h, size_t offset, const void* scalars, size_t n, const zk_msm_cfg* cfg, void* out, int on_device);

int md_groth16_pk_load(const zk_groth16_pk* pk, const std::vector<int>& entries, uint64_t* handle);
int md_groth16_pk_free(uint64_t h);
int md_groth16_pk_info(uint64_t h, size_t* n_wires, size_t* n_public, uint32_t* log_domain, int* has_tables, int* n_entries_out);
int md_groth16_finalize(uint64_t h, const uint64_t* partials, size_t n_partials, const zk_fr* r, const zk_fr* s, uint8_t proof_out[128]);
int md_groth16_prove(uint64_t h, const void* a, const void* b, const void* c, size_t n_constraints, const void* w, size_t n_wires, const zk_fr* r, const zk_fr* s,
                     int on_device, uint8_t proof_out[128]);
int md_ntt_host(zk_fr* a, uint32_t log_n, int inverse, int decimation, int coset, const std::vector<int>& entries);

}  // namespace zkmi
