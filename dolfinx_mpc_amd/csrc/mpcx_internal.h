// Internal helpers shared by the host and device translation units of libmpcx.
#pragma once
#include <string>

void mpcx_set_error(const std::string& msg);

#include "mpcx.h"
namespace mpcx
{
// cluster ("Kuhn fan") kernels, mpcx_cubes.hip
int launch_matrix_cubes(const mpcx_matrix_args_t& a);
int launch_vector_cubes(const mpcx_vector_args_t& a);
// pair records + cached entity contexts (plan.row_pairs == 2), mpcx_pairs.hip
int launch_matrix_pairs(const mpcx_matrix_args_t& a);
// scalar types other than fp64 real (mpcx_kernel_t::scalar_type != 0), mpcx_scalar.hip
int launch_matrix_scalar(const mpcx_matrix_args_t& a);
int launch_vector_scalar(const mpcx_vector_args_t& a);
int launch_lifting_scalar(const mpcx_lifting_args_t& a);
// imported UFCx kernels, mpcx_ufcx.cpp
int launch_matrix_ufcx(const mpcx_matrix_args_t& a);
int launch_vector_ufcx(const mpcx_vector_args_t& a);
int launch_lifting_ufcx(const mpcx_lifting_args_t& a);
// second half of the owner-computes vector path (mpcx_kernels.hip), shared with the imported kernels
int launch_vector_spill_reduce(const mpcx_vector_args_t& a, int bs);
// slave rows of a.slave_entities through the built-in operator of a.kernel (vector_mpc_kernel, mpcx_kernels.hip)
int launch_vector_slave_rows(const mpcx_vector_args_t& a);
} // namespace mpcx
