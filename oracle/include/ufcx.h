/* Minimal UFCx declarations for the ORACLE's gcc builds of FFCx-layout files (test infrastructure, never shipped with
 * the product: the product strips the descriptor objects and needs no header, csrc/mpcx_ufcx.cpp).
 *
 * FFCx-generated files `#include <ufcx.h>` and end with `ufcx_integral` / `ufcx_form` objects whose members point at the
 * tabulate_tensor functions; DOLFINx -- and through it the reference -- reaches a kernel ONLY through these objects
 * (cpp/assemble_matrix.cpp:438-439 takes `a.kernel(IntegralType::cell, i, 0)`, which DOLFINx filled from
 * `form->form_integrals[k]->tabulate_tensor_float64`).  The oracle does the same with ctypes (oracle/pyoracle.py), so the
 * declarations below only have to agree with the objects the files under tests/ufcx/ffcx_layout_*.c define: member
 * names and order follow the public UFCx interface of FFCx 0.8 - 0.10 as far as those files use it.  Written for this
 * repository; FFCx is not present in this image. */
#ifndef MPCX_ORACLE_UFCX_H
#define MPCX_ORACLE_UFCX_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#define UFCX_VERSION_MAJOR 0
#define UFCX_VERSION_MINOR 10

#if defined(__cplusplus)
#define restrict __restrict__
extern "C" {
#endif

typedef enum { cell = 0, exterior_facet = 1, interior_facet = 2, vertex = 3 } ufcx_integral_type;

typedef void(ufcx_tabulate_tensor_float32)(float* restrict A, const float* restrict w, const float* restrict c,
                                           const float* restrict coordinate_dofs, const int* restrict entity_local_index,
                                           const uint8_t* restrict quadrature_permutation, void* custom_data);
typedef void(ufcx_tabulate_tensor_float64)(double* restrict A, const double* restrict w, const double* restrict c,
                                           const double* restrict coordinate_dofs, const int* restrict entity_local_index,
                                           const uint8_t* restrict quadrature_permutation, void* custom_data);
#ifndef __STDC_NO_COMPLEX__
typedef void(ufcx_tabulate_tensor_complex64)(float _Complex* restrict A, const float _Complex* restrict w,
                                             const float _Complex* restrict c, const float* restrict coordinate_dofs,
                                             const int* restrict entity_local_index,
                                             const uint8_t* restrict quadrature_permutation, void* custom_data);
typedef void(ufcx_tabulate_tensor_complex128)(double _Complex* restrict A, const double _Complex* restrict w,
                                              const double _Complex* restrict c, const double* restrict coordinate_dofs,
                                              const int* restrict entity_local_index,
                                              const uint8_t* restrict quadrature_permutation, void* custom_data);
#endif

typedef struct ufcx_integral
{
  const bool* enabled_coefficients;
  ufcx_tabulate_tensor_float32* tabulate_tensor_float32;
  ufcx_tabulate_tensor_float64* tabulate_tensor_float64;
#ifndef __STDC_NO_COMPLEX__
  ufcx_tabulate_tensor_complex64* tabulate_tensor_complex64;
  ufcx_tabulate_tensor_complex128* tabulate_tensor_complex128;
#endif
  bool needs_facet_permutations;
  uint64_t coordinate_element_hash;
  uint8_t domain;
} ufcx_integral;

typedef struct ufcx_form
{
  const char* signature;
  int rank;
  int num_coefficients;
  int num_constants;
  int* original_coefficient_positions;
  const char** coefficient_name_map;
  const char** constant_name_map;
  uint64_t* finite_element_hashes;
  ufcx_integral** form_integrals;
  int* form_integral_ids;
  int* form_integral_offsets;
} ufcx_form;

#if defined(__cplusplus)
}
#undef restrict
#endif
#endif
