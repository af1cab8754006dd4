"""ctypes binding of libmpcx.so (C ABI in include/mpcx.h).

This is the only door to the compute path: there is no Python or CPU fallback.
If the library is missing or was not built, importing the assembly functions
fails loudly.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

# MPCX_LIBRARY: alternative build of the same ABI (kernel experiments, tools/ablate.sh)
_LIB_PATH = os.environ.get("MPCX_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmpcx.so")


class KernelT(C.Structure):
    _fields_ = [
        ("form", C.c_int32),
        ("celltype", C.c_int32),
        ("degree", C.c_int32),
        ("bs", C.c_int32),
        ("degree1", C.c_int32),
        ("bs1", C.c_int32),
        ("fn_id", C.c_int32),
        ("coeff_degree", C.c_int32),
        ("nq", C.c_int32),
        ("nqf", C.c_int32),
        ("qpts", C.c_void_p),
        ("qwts", C.c_void_p),
        ("fqpts", C.c_void_p),
        ("fqwts", C.c_void_p),
        ("ufcx", C.c_void_p),
        ("qphi", C.c_void_p),
        ("scalar_type", C.c_int32),
        ("vphi", C.c_void_p),
    ]


# mpcx_kernel_t::scalar_type (include/mpcx.h MPCX_SCALAR_*): the reference's four instantiations
SCALAR_TYPES = {"float64": 0, "float32": 1, "complex128": 2, "complex64": 3}


def scalar_id(dtype) -> int:
    import numpy as np

    name = np.dtype(dtype).name
    if name not in SCALAR_TYPES:
        raise NotImplementedError(f"scalar type {name}: float32, float64, complex64, complex128 (python/src/dolfinx_mpc/multipointconstraint.py:55-64)")
    return SCALAR_TYPES[name]


class UfcxDescT(C.Structure):
    _fields_ = [
        ("source", C.c_char_p),
        ("function_name", C.c_char_p),
        ("rank", C.c_int32),
        ("nd0", C.c_int32),
        ("bs0", C.c_int32),
        ("nd1", C.c_int32),
        ("bs1", C.c_int32),
        ("nv", C.c_int32),
        ("transform0_name", C.c_char_p),
        ("transform1_name", C.c_char_p),
    ]


class MpcT(C.Structure):
    _fields_ = [
        ("is_slave", C.c_void_p),
        ("masters_offsets", C.c_void_p),
        ("masters", C.c_void_p),
        ("coeffs", C.c_void_p),
    ]


class RowBlockPlanT(C.Structure):
    _fields_ = [
        ("num_blocks", C.c_int32),
        ("max_rows", C.c_int32),
        ("max_nnz", C.c_int32),
        ("row_pairs", C.c_int32),
        ("block_row0", C.c_void_p),
        ("block_ent_off", C.c_void_p),
        ("block_ents", C.c_void_p),
        ("ent_offs", C.c_void_p),
        ("ent_pattern", C.c_void_p),
    ]


class MatrixArgs(C.Structure):
    _fields_ = [
        ("nrows", C.c_int32),
        ("rowptr", C.c_void_p),
        ("cols", C.c_void_p),
        ("vals", C.c_void_p),
        ("kernel", KernelT),
        ("x", C.c_void_p),
        ("x_dofmap", C.c_void_p),
        ("nv", C.c_int32),
        ("estride", C.c_int32),
        ("n_entities", C.c_int64),
        ("entities", C.c_void_p),
        ("entities0", C.c_void_p),
        ("entities1", C.c_void_p),
        ("coeffs", C.c_void_p),
        ("cstride", C.c_int32),
        ("constants", C.c_void_p),
        ("dofmap0", C.c_void_p),
        ("nd0", C.c_int32),
        ("bs0", C.c_int32),
        ("dofmap1", C.c_void_p),
        ("nd1", C.c_int32),
        ("bs1", C.c_int32),
        ("bc0", C.c_void_p),
        ("bc1", C.c_void_p),
        ("mpc0", MpcT),
        ("mpc1", MpcT),
        ("slave_entities", C.c_void_p),
        ("n_slave_entities", C.c_int64),
        ("algorithm", C.c_int32),
        ("store_mode", C.c_int32),
        ("plan", RowBlockPlanT),
        ("mdofmap0", C.c_void_p),
        ("mdofmap1", C.c_void_p),
        ("lean", C.c_int32),
        ("cube_recs", C.c_void_p),
        ("cube_rec_bytes", C.c_int32),
        ("cube_flags", C.c_int32),
        ("cube_block_ids", C.c_void_p),
        ("slot_mask", C.c_void_p),
        ("mpc_plan_targets", C.c_int64),
        ("mpc_plan_tgt", C.c_void_p),
        ("mpc_plan_off", C.c_void_p),
        ("mpc_plan_ent", C.c_void_p),
        ("mpc_plan_pq", C.c_void_p),
        ("mpc_plan_coef", C.c_void_p),
        ("mpc_plan_group", C.c_int32),
        ("block_vals", C.c_void_p),
        ("mpc_plan_out", C.c_void_p),
        ("slave_tensors", C.c_void_p),
        ("mpc_plan_slot", C.c_void_p),
        ("pair_recs", C.c_void_p),
        ("pair_ctx", C.c_void_p),
        ("pair_dict", C.c_void_p),
        ("cube_rec_index", C.c_void_p),
        ("cube_cells", C.c_void_p),
        ("cell_info0", C.c_void_p),
        ("cell_info1", C.c_void_p),
        ("val_map", C.c_void_p),
        ("val_map_wide", C.c_int32),
        ("out_map", C.c_void_p),
        ("out_delta", C.c_void_p),
        ("lds_floor", C.c_int32),
        ("stream", C.c_void_p),
    ]


class VectorArgs(C.Structure):
    _fields_ = [
        ("b", C.c_void_p),
        ("num_dofs", C.c_int32),
        ("kernel", KernelT),
        ("x", C.c_void_p),
        ("x_dofmap", C.c_void_p),
        ("nv", C.c_int32),
        ("estride", C.c_int32),
        ("n_entities", C.c_int64),
        ("entities", C.c_void_p),
        ("entities0", C.c_void_p),
        ("coeffs", C.c_void_p),
        ("cstride", C.c_int32),
        ("constants", C.c_void_p),
        ("dofmap", C.c_void_p),
        ("nd", C.c_int32),
        ("bs", C.c_int32),
        ("mpc", MpcT),
        ("algorithm", C.c_int32),
        ("plan", RowBlockPlanT),
        ("mdofmap", C.c_void_p),
        ("slave_entities", C.c_void_p),
        ("n_slave_entities", C.c_int64),
        ("cube_verts", C.c_void_p),
        ("n_cubes", C.c_int64),
        ("own_lmap", C.c_void_p),
        ("own_hoff", C.c_void_p),
        ("own_spill", C.c_void_p),
        ("own_src", C.c_void_p),
        ("own_rows", C.c_void_p),
        ("own_seg", C.c_void_p),
        ("n_own_rows", C.c_int64),
        ("cube_cells", C.c_void_p),
        ("cell_info0", C.c_void_p),
        ("row_map", C.c_void_p),
        ("cube_boxes", C.c_int32),
        ("grid_idx", C.c_void_p),
        ("grid_iv", C.c_void_p),
        ("grid_tab", C.c_void_p),
        ("grid_n", C.c_int32 * 3),
        ("grid_block_rows", C.c_void_p),
        ("grid_block_rows_max", C.c_int32),
        ("grid_eta", C.c_void_p),
        ("grid_J", C.c_void_p),
        ("grid_ng", C.c_int32),
        ("grid_ntypes", C.c_int32),
        ("lds_floor", C.c_int32),
        ("stream", C.c_void_p),
    ]


class LiftingArgs(C.Structure):
    _fields_ = [
        ("b", C.c_void_p),
        ("num_dofs", C.c_int32),
        ("kernel", KernelT),
        ("x", C.c_void_p),
        ("x_dofmap", C.c_void_p),
        ("nv", C.c_int32),
        ("estride", C.c_int32),
        ("n_entities", C.c_int64),
        ("entities", C.c_void_p),
        ("entities0", C.c_void_p),
        ("entities1", C.c_void_p),
        ("coeffs", C.c_void_p),
        ("cstride", C.c_int32),
        ("constants", C.c_void_p),
        ("dofmap0", C.c_void_p),
        ("nd0", C.c_int32),
        ("bs0", C.c_int32),
        ("dofmap1", C.c_void_p),
        ("nd1", C.c_int32),
        ("bs1", C.c_int32),
        ("bc_markers1", C.c_void_p),
        ("bc_values1", C.c_void_p),
        ("x0", C.c_void_p),
        ("scale", C.c_double),
        ("lift_entities", C.c_void_p),
        ("n_lift_entities", C.c_int64),
        ("mpc0", MpcT),
        ("cell_info0", C.c_void_p),
        ("cell_info1", C.c_void_p),
        ("row_map", C.c_void_p),
        ("stream", C.c_void_p),
    ]


# every symbol include/mpcx.h declares
EXPORTS = [
    "mpcx_assemble_matrix",
    "mpcx_assemble_fused",
    "mpcx_mask_dofmap",
    "mpcx_scatter_offsets",
    "mpcx_cube_records",
    "mpcx_hex_records",
    "mpcx_hex_slot_shapes",
    "mpcx_cell_shapes",
    "mpcx_p2_cluster_dofs",
    "mpcx_p2_cluster_records",
    "mpcx_p2_cluster_tables",
    "mpcx_p1_cluster_tables",
    "mpcx_cluster_plan_create",
    "mpcx_cluster_plan_num_parts",
    "mpcx_cluster_plan_num_clusters",
    "mpcx_cluster_plan_num_slots",
    "mpcx_cluster_plan_verts",
    "mpcx_cluster_plan_leftover",
    "mpcx_cluster_plan_part",
    "mpcx_cluster_plan_destroy",
    "mpcx_ufcx_big_tensor",
    "mpcx_ufcx_rowwise",
    "mpcx_cell_plan_create",
    "mpcx_cell_plan_fill",
    "mpcx_cell_plan_num_slots",
    "mpcx_cell_plan_num_blocks",
    "mpcx_cell_plan_destroy",
    "mpcx_pairs_plan_create",
    "mpcx_pairs_plan_update_geometry",
    "mpcx_pairs_plan_fill",
    "mpcx_pairs_plan_num_pairs",
    "mpcx_pairs_plan_num_blocks",
    "mpcx_pairs_plan_destroy",
    "mpcx_nodeblock_plan_create",
    "mpcx_nodeblock_plan_fill",
    "mpcx_nodeblock_plan_destroy",
    "mpcx_master_plan_create",
    "mpcx_master_plan_fill",
    "mpcx_master_plan_num_targets",
    "mpcx_master_plan_num_tuples",
    "mpcx_master_plan_destroy",
    "mpcx_grid_plan_create",
    "mpcx_grid_plan_fill",
    "mpcx_grid_plan_num_intervals",
    "mpcx_grid_plan_block_rows",
    "mpcx_grid_plan_destroy",
    "mpcx_cell_grid_plan_create",
    "mpcx_cell_grid_plan_fill",
    "mpcx_cell_grid_plan_destroy",
    "mpcx_owner_plan_create",
    "mpcx_owner_plan_fill",
    "mpcx_owner_plan_destroy",
    "mpcx_cube_detect",
    "mpcx_cube_slot_width",
    "mpcx_cube_pack_narrow",
    "mpcx_cluster_keys",
    "mpcx_cluster_build",
    "mpcx_cluster_canonical",
    "mpcx_cluster_ordered",
    "mpcx_renumber_mesh",
    "mpcx_dof_permutation",
    "mpcx_rowblock_pairs_device",
    "mpcx_hbm_probe",
    "mpcx_add_diagonal_scalar",
    "mpcx_backsubstitution_scalar",
    "mpcx_homogenize_scalar",
    "mpcx_csr_permutation",
    "mpcx_permute_values",
    "mpcx_invert_permutation",
    "mpcx_write_out_order",
    "mpcx_pair_words",
    "mpcx_pair_records",
    "mpcx_pair_dict_stride",
    "mpcx_pair_compress_workspace",
    "mpcx_pair_compress",
    "mpcx_pair_context_size",
    "mpcx_pair_context",
    "mpcx_diag_slot_mask",
    "mpcx_add_diagonal",
    "mpcx_add_diagonal_mapped",
    "mpcx_assemble_vector",
    "mpcx_apply_lifting",
    "mpcx_backsubstitution",
    "mpcx_homogenize",
    "mpcx_mpc_finalize",
    "mpcx_cell_to_slaves",
    "mpcx_mpc_finalize_device",
    "mpcx_cell_to_slaves_device",
    "mpcx_scan_exclusive_i32_i64",
    "mpcx_scan_exclusive_i32",
    "mpcx_scan_exclusive_i64",
    "mpcx_segment_offsets",
    "mpcx_sort_pairs_i64_i32",
    "mpcx_sort_pairs_i64_i64",
    "mpcx_run_heads",
    "mpcx_run_fill",
    "mpcx_block_ranges",
    "mpcx_owner_plan_count",
    "mpcx_owner_plan_keys",
    "mpcx_owner_plan_halo",
    "mpcx_low_word_iota",
    "mpcx_pattern_build",
    "mpcx_pattern_nnz",
    "mpcx_pattern_nrows",
    "mpcx_pattern_copy",
    "mpcx_pattern_free",
    "mpcx_pattern_device_adjacency",
    "mpcx_pattern_device_rows",
    "mpcx_rowblock_plan_build",
    "mpcx_rowblock_plan_num_blocks",
    "mpcx_rowblock_plan_num_ents",
    "mpcx_rowblock_plan_copy",
    "mpcx_rowblock_plan_free",
    "mpcx_mpc_plan_build",
    "mpcx_mpc_plan_size",
    "mpcx_mpc_plan_num_targets",
    "mpcx_mpc_plan_copy",
    "mpcx_mpc_plan_free",
    "mpcx_mpc_plan_device",
    "mpcx_compress_offsets",
    "mpcx_ufcx_compile",
    "mpcx_ufcx_resolve",
    "mpcx_ufcx_code_size",
    "mpcx_ufcx_code",
    "mpcx_ufcx_free",
    "mpcx_gather_f64",
    "mpcx_scatter_add_f64",
    "mpcx_spmv",
    "mpcx_block_expand",
    "mpcx_spmv_blockscalar",
    "mpcx_csr_positions",
    "mpcx_spmv_coo_add",
    "mpcx_inverse_diagonal",
    "mpcx_cg_start",
    "mpcx_cg_step",
    "mpcx_last_error",
    "mpcx_version",
    "mpcx_preload",
    "mpcx_device_count",
]

_lib = None


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def lib() -> C.CDLL:
    """Load libmpcx.so; raise if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} not found: the HIP backend has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    # One HIP runtime per process: torch ships its own libamdhip64.so and owns
    # the device allocations/streams we are handed, so make sure that copy is
    # the one libmpcx.so's NEEDED libamdhip64.so.7 resolves to (same SONAME).
    import torch

    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
    L = C.CDLL(_LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.mpcx_assemble_matrix.argtypes = [C.POINTER(MatrixArgs)]
    L.mpcx_assemble_matrix.restype = C.c_int
    L.mpcx_assemble_vector.argtypes = [C.POINTER(VectorArgs)]
    L.mpcx_assemble_vector.restype = C.c_int
    L.mpcx_assemble_fused.argtypes = [C.POINTER(MatrixArgs), C.POINTER(VectorArgs), vp]
    L.mpcx_assemble_fused.restype = C.c_int
    L.mpcx_apply_lifting.argtypes = [C.POINTER(LiftingArgs)]
    L.mpcx_apply_lifting.restype = C.c_int
    L.mpcx_add_diagonal.argtypes = [i32, vp, vp, vp, vp, i64, dbl, vp]
    L.mpcx_add_diagonal.restype = C.c_int
    L.mpcx_add_diagonal_mapped.argtypes = [i32, vp, vp, vp, vp, i64, dbl, vp, i32, vp]
    L.mpcx_add_diagonal_mapped.restype = C.c_int
    L.mpcx_backsubstitution.argtypes = [vp, vp, i64, C.POINTER(MpcT), vp]
    L.mpcx_backsubstitution.restype = C.c_int
    L.mpcx_homogenize.argtypes = [vp, vp, i64, vp]
    L.mpcx_homogenize.restype = C.c_int
    L.mpcx_mpc_finalize.argtypes = [i32, i32, i32] + [vp] * 12
    L.mpcx_mpc_finalize.restype = C.c_int
    L.mpcx_cell_to_slaves.argtypes = [i64, i32, i32, vp, vp, vp, vp]
    L.mpcx_cell_to_slaves.restype = i64
    szp = C.POINTER(C.c_size_t)
    L.mpcx_mpc_finalize_device.argtypes = [i32, i32, i32] + [vp] * 14 + [vp, szp, vp]
    L.mpcx_mpc_finalize_device.restype = C.c_int
    L.mpcx_cell_to_slaves_device.argtypes = [i64, i32, i32, vp, vp, vp, vp, vp, vp]
    L.mpcx_cell_to_slaves_device.restype = C.c_int
    L.mpcx_scan_exclusive_i32_i64.argtypes = [vp, i64, vp, vp, szp, vp]
    L.mpcx_scan_exclusive_i32_i64.restype = C.c_int
    L.mpcx_scan_exclusive_i32.argtypes = [vp, i64, vp, vp, szp, vp]
    L.mpcx_scan_exclusive_i32.restype = C.c_int
    L.mpcx_scan_exclusive_i64.argtypes = [vp, i64, vp, vp, szp, vp]
    L.mpcx_scan_exclusive_i64.restype = C.c_int
    L.mpcx_segment_offsets.argtypes = [vp, i64, i32, i64, vp, vp]
    L.mpcx_segment_offsets.restype = C.c_int
    L.mpcx_sort_pairs_i64_i32.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, szp, vp]
    L.mpcx_sort_pairs_i64_i32.restype = C.c_int
    L.mpcx_sort_pairs_i64_i64.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, szp, vp]
    L.mpcx_sort_pairs_i64_i64.restype = C.c_int
    L.mpcx_run_heads.argtypes = [vp, i64, vp, vp]
    L.mpcx_run_heads.restype = C.c_int
    L.mpcx_run_fill.argtypes = [vp, vp, vp, i64, vp, vp, vp]
    L.mpcx_run_fill.restype = C.c_int
    L.mpcx_block_ranges.argtypes = [i32, vp, i32, i32, i32, vp, i32, vp, i64]
    L.mpcx_block_ranges.restype = i64
    L.mpcx_owner_plan_count.argtypes = [i64, i32, vp, i32, i32, vp, vp, vp, vp, vp]
    L.mpcx_owner_plan_count.restype = C.c_int
    L.mpcx_owner_plan_keys.argtypes = [i64, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.mpcx_owner_plan_keys.restype = C.c_int
    L.mpcx_owner_plan_halo.argtypes = [i64, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp]
    L.mpcx_owner_plan_halo.restype = C.c_int
    L.mpcx_low_word_iota.argtypes = [i64, vp, vp, vp, vp]
    L.mpcx_low_word_iota.restype = C.c_int
    L.mpcx_pattern_build.argtypes = [i64, vp, i32, i32, i32, vp, i32, i32, i32] + [vp] * 8 + [i32]
    L.mpcx_pattern_build.restype = vp
    L.mpcx_pattern_nnz.argtypes = [vp]
    L.mpcx_pattern_nnz.restype = i64
    L.mpcx_pattern_nrows.argtypes = [vp]
    L.mpcx_pattern_nrows.restype = i32
    L.mpcx_pattern_copy.argtypes = [vp, vp, vp]
    L.mpcx_pattern_copy.restype = C.c_int
    L.mpcx_pattern_free.argtypes = [vp]
    L.mpcx_pattern_free.restype = None
    L.mpcx_pattern_device_adjacency.argtypes = [i64, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpcx_pattern_device_adjacency.restype = C.c_int
    L.mpcx_pattern_device_rows.argtypes = [i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    L.mpcx_pattern_device_rows.restype = C.c_int
    L.mpcx_mask_dofmap.argtypes = [vp, i64, i32, i32, vp, vp, i32, vp, vp]
    L.mpcx_mask_dofmap.restype = C.c_int
    L.mpcx_scatter_offsets.argtypes = [vp, vp, i32, i64, vp, vp, vp, i32, i32, vp, i32, i32, i32, vp, vp, vp]
    L.mpcx_scatter_offsets.restype = C.c_int
    L.mpcx_cube_records.argtypes = [i64, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.mpcx_cube_records.restype = C.c_int
    L.mpcx_hex_records.argtypes = [i64, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.mpcx_hex_records.restype = C.c_int
    L.mpcx_hex_slot_shapes.argtypes = [i64, vp, vp, vp, vp]
    L.mpcx_hex_slot_shapes.restype = C.c_int
    L.mpcx_cluster_canonical.argtypes = [i64, vp, vp, vp, vp]
    L.mpcx_cluster_canonical.restype = C.c_int
    L.mpcx_cluster_ordered.argtypes = [i64, vp, vp, vp, vp, vp]
    L.mpcx_cluster_ordered.restype = C.c_int
    L.mpcx_renumber_mesh.argtypes = [vp, i64, vp, i64, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.mpcx_renumber_mesh.restype = C.c_int
    L.mpcx_dof_permutation.argtypes = [vp, vp, vp, i64, C.c_int32, vp, vp]
    L.mpcx_dof_permutation.restype = C.c_int
    L.mpcx_cell_shapes.argtypes = [i64, vp, vp, vp, vp]
    L.mpcx_cell_shapes.restype = C.c_int
    L.mpcx_p2_cluster_dofs.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp]
    L.mpcx_p2_cluster_dofs.restype = C.c_int
    L.mpcx_p2_cluster_records.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpcx_p2_cluster_records.restype = C.c_int
    L.mpcx_p2_cluster_tables.argtypes = [vp, vp, vp, vp]
    L.mpcx_p2_cluster_tables.restype = C.c_int
    L.mpcx_p1_cluster_tables.argtypes = [vp, vp, vp]
    L.mpcx_p1_cluster_tables.restype = C.c_int
    L.mpcx_cluster_plan_create.argtypes = [i64, vp, i64, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, i32, vp, C.POINTER(C.c_void_p)]
    L.mpcx_cluster_plan_create.restype = C.c_int
    for name, rt in (("num_parts", C.c_int32), ("num_clusters", C.c_int64), ("num_slots", C.c_int64), ("verts", C.c_void_p)):
        f = getattr(L, "mpcx_cluster_plan_" + name)
        f.argtypes, f.restype = [vp], rt
    L.mpcx_cluster_plan_leftover.argtypes = [vp, C.POINTER(C.c_void_p)]
    L.mpcx_cluster_plan_leftover.restype = C.c_int64
    L.mpcx_cluster_plan_part.argtypes = [vp, i32, C.POINTER(MatrixArgs)]
    L.mpcx_cluster_plan_part.restype = C.c_int
    L.mpcx_cluster_plan_destroy.argtypes = [vp]
    L.mpcx_cluster_plan_destroy.restype = None
    L.mpcx_cell_plan_create.argtypes = [i32, vp, vp, vp, i64, i32, vp, i64, vp, i32, i32, vp, vp, vp, i32, i32, vp, vp, i32, i32, vp, i32, i32, vp,
                                        C.POINTER(C.c_void_p)]
    L.mpcx_cell_plan_create.restype = C.c_int
    L.mpcx_ufcx_big_tensor.argtypes = [vp]
    L.mpcx_ufcx_big_tensor.restype = C.c_int
    L.mpcx_ufcx_rowwise.argtypes = [vp]
    L.mpcx_ufcx_rowwise.restype = C.c_int
    L.mpcx_cell_plan_fill.argtypes = [vp, vp]
    L.mpcx_cell_plan_fill.restype = C.c_int
    L.mpcx_cell_plan_num_slots.argtypes = [vp]
    L.mpcx_cell_plan_num_slots.restype = C.c_int64
    L.mpcx_cell_plan_num_blocks.argtypes = [vp]
    L.mpcx_cell_plan_num_blocks.restype = C.c_int32
    L.mpcx_cell_plan_destroy.argtypes = [vp]
    L.mpcx_cell_plan_destroy.restype = None
    L.mpcx_owner_plan_create.argtypes = [i64, i32, vp, i32, i32, i32, vp, i32, i32, vp, C.POINTER(C.c_void_p)]
    L.mpcx_owner_plan_create.restype = C.c_int
    L.mpcx_owner_plan_fill.argtypes = [vp, C.POINTER(VectorArgs)]
    L.mpcx_owner_plan_fill.restype = C.c_int
    L.mpcx_owner_plan_destroy.argtypes = [vp]
    L.mpcx_owner_plan_destroy.restype = None
    L.mpcx_cube_detect.argtypes = [vp, i64, vp, vp, vp]
    L.mpcx_cube_detect.restype = C.c_int
    L.mpcx_cube_slot_width.argtypes = [i64, vp, vp, vp]
    L.mpcx_cube_slot_width.restype = C.c_int
    L.mpcx_cube_pack_narrow.argtypes = [i64, vp, vp, vp, vp]
    L.mpcx_cube_pack_narrow.restype = C.c_int
    L.mpcx_cluster_keys.argtypes = [vp, vp, i64, vp, vp]
    L.mpcx_cluster_keys.restype = C.c_int
    L.mpcx_cluster_build.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp]
    L.mpcx_cluster_build.restype = C.c_int
    L.mpcx_rowblock_pairs_device.argtypes = [i64, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]
    L.mpcx_rowblock_pairs_device.restype = C.c_int
    L.mpcx_add_diagonal_scalar.argtypes = [i32, vp, vp, vp, vp, i64, dbl, dbl, vp]
    L.mpcx_add_diagonal_scalar.restype = C.c_int
    L.mpcx_backsubstitution_scalar.argtypes = [i32, vp, vp, i64, C.POINTER(MpcT), vp]
    L.mpcx_backsubstitution_scalar.restype = C.c_int
    L.mpcx_homogenize_scalar.argtypes = [i32, vp, vp, i64, vp]
    L.mpcx_homogenize_scalar.restype = C.c_int
    L.mpcx_hbm_probe.argtypes = [vp, vp, i64, i32, vp]
    L.mpcx_hbm_probe.restype = C.c_int
    L.mpcx_csr_permutation.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    L.mpcx_csr_permutation.restype = C.c_int
    L.mpcx_permute_values.argtypes = [i64, vp, i32, vp, vp, vp]
    L.mpcx_permute_values.restype = C.c_int
    L.mpcx_invert_permutation.argtypes = [i64, vp, i32, vp, vp]
    L.mpcx_invert_permutation.restype = C.c_int
    L.mpcx_write_out_order.argtypes = [i32, vp, vp, i32, vp, vp, vp, vp]
    L.mpcx_write_out_order.restype = C.c_int
    L.mpcx_pair_words.argtypes = [i32]
    L.mpcx_pair_words.restype = i32
    L.mpcx_pair_records.argtypes = [i64, vp, i32, vp, vp, vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    L.mpcx_pair_records.restype = C.c_int
    L.mpcx_pair_dict_stride.argtypes = [i32]
    L.mpcx_pair_dict_stride.restype = i32
    L.mpcx_pair_compress_workspace.argtypes = [i32]
    L.mpcx_pair_compress_workspace.restype = i64
    L.mpcx_pair_compress.argtypes = [i64, vp, i32, vp, vp, C.POINTER(C.c_int32), vp, vp]
    L.mpcx_pair_compress.restype = C.c_int
    L.mpcx_pair_context_size.argtypes = [C.POINTER(KernelT)]
    L.mpcx_pair_context_size.restype = i32
    L.mpcx_pair_context.argtypes = [C.POINTER(KernelT), i64, i32, vp, vp, vp, i32, vp, vp]
    L.mpcx_pair_context.restype = C.c_int
    L.mpcx_diag_slot_mask.argtypes = [i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.mpcx_diag_slot_mask.restype = C.c_int
    L.mpcx_rowblock_plan_build.argtypes = [i32, vp, i32, i32, i64, i32, vp, vp, i32, i32, vp, i32, i32]
    L.mpcx_rowblock_plan_build.restype = vp
    L.mpcx_rowblock_plan_num_blocks.argtypes = [vp]
    L.mpcx_rowblock_plan_num_blocks.restype = i32
    L.mpcx_rowblock_plan_num_ents.argtypes = [vp]
    L.mpcx_rowblock_plan_num_ents.restype = i64
    L.mpcx_rowblock_plan_copy.argtypes = [vp, vp, vp, vp]
    L.mpcx_rowblock_plan_copy.restype = C.c_int
    L.mpcx_rowblock_plan_free.argtypes = [vp]
    L.mpcx_rowblock_plan_free.restype = None
    L.mpcx_mpc_plan_build.argtypes = [i64, vp, i32, vp, vp, vp, i32, i32, vp, i32, i32] + [vp] * 12
    L.mpcx_mpc_plan_build.restype = vp
    L.mpcx_mpc_plan_size.argtypes = [vp]
    L.mpcx_mpc_plan_size.restype = i64
    L.mpcx_mpc_plan_num_targets.argtypes = [vp]
    L.mpcx_mpc_plan_num_targets.restype = i64
    L.mpcx_mpc_plan_copy.argtypes = [vp, vp, vp, vp, vp, vp]
    L.mpcx_mpc_plan_copy.restype = C.c_int
    L.mpcx_mpc_plan_free.argtypes = [vp]
    L.mpcx_mpc_plan_free.restype = None
    L.mpcx_mpc_plan_device.argtypes = [i64, vp, i32, vp, vp, vp, i32, i32, vp, i32, i32, vp, vp, C.POINTER(MpcT),
                                       C.POINTER(MpcT), vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.mpcx_mpc_plan_device.restype = C.c_int
    L.mpcx_ufcx_resolve.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32]
    L.mpcx_ufcx_resolve.restype = C.c_int
    L.mpcx_ufcx_compile.argtypes = [C.POINTER(UfcxDescT)]
    L.mpcx_ufcx_compile.restype = vp
    L.mpcx_ufcx_code_size.argtypes = [vp]
    L.mpcx_ufcx_code_size.restype = i64
    L.mpcx_ufcx_code.argtypes = [vp, vp]
    L.mpcx_ufcx_code.restype = C.c_int
    L.mpcx_ufcx_free.argtypes = [vp]
    L.mpcx_ufcx_free.restype = None
    L.mpcx_compress_offsets.argtypes = [vp, i64, i32, i32, vp, vp]
    L.mpcx_compress_offsets.restype = i32
    L.mpcx_gather_f64.argtypes = [vp, vp, i64, vp, vp]
    L.mpcx_gather_f64.restype = C.c_int
    L.mpcx_scatter_add_f64.argtypes = [vp, vp, i64, vp, vp]
    L.mpcx_scatter_add_f64.restype = C.c_int
    L.mpcx_spmv.argtypes = [i32, vp, vp, vp, vp, vp, vp]
    L.mpcx_spmv.restype = C.c_int
    L.mpcx_block_expand.argtypes = [i32, vp, i32, vp, vp, vp, vp]
    L.mpcx_block_expand.restype = C.c_int
    L.mpcx_spmv_blockscalar.argtypes = [i32, vp, vp, i32, vp, vp, vp, vp, vp]
    L.mpcx_spmv_blockscalar.restype = C.c_int
    L.mpcx_csr_positions.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.mpcx_csr_positions.restype = C.c_int
    L.mpcx_spmv_coo_add.argtypes = [i64, vp, vp, vp, vp, vp, vp]
    L.mpcx_spmv_coo_add.restype = C.c_int
    L.mpcx_inverse_diagonal.argtypes = [i32, vp, vp, vp, vp, vp]
    L.mpcx_inverse_diagonal.restype = C.c_int
    L.mpcx_cg_start.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpcx_cg_start.restype = C.c_int
    L.mpcx_cg_step.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    L.mpcx_cg_step.restype = C.c_int
    L.mpcx_last_error.argtypes = []
    L.mpcx_last_error.restype = C.c_char_p
    L.mpcx_grid_plan_create.argtypes = [vp, i64, vp, C.POINTER(RowBlockPlanT), vp, C.POINTER(vp)]
    L.mpcx_grid_plan_create.restype = C.c_int
    L.mpcx_grid_plan_fill.argtypes = [vp, C.POINTER(VectorArgs)]
    L.mpcx_grid_plan_fill.restype = C.c_int
    L.mpcx_grid_plan_num_intervals.argtypes = [vp, i32]
    L.mpcx_grid_plan_num_intervals.restype = i32
    L.mpcx_grid_plan_block_rows.argtypes = [vp]
    L.mpcx_grid_plan_block_rows.restype = i32
    L.mpcx_grid_plan_destroy.argtypes = [vp]
    L.mpcx_grid_plan_destroy.restype = None
    L.mpcx_cell_grid_plan_create.argtypes = [vp, i64, vp, C.POINTER(RowBlockPlanT), vp, i32, vp, C.POINTER(vp)]
    L.mpcx_cell_grid_plan_create.restype = C.c_int
    L.mpcx_cell_grid_plan_fill.argtypes = [vp, C.POINTER(VectorArgs)]
    L.mpcx_cell_grid_plan_fill.restype = C.c_int
    L.mpcx_cell_grid_plan_destroy.argtypes = [vp]
    L.mpcx_cell_grid_plan_destroy.restype = None
    L.mpcx_version.argtypes = []
    L.mpcx_version.restype = C.c_int
    L.mpcx_preload.argtypes = [vp]
    L.mpcx_preload.restype = C.c_int
    L.mpcx_device_count.argtypes = []
    L.mpcx_device_count.restype = C.c_int
    _lib = L
    return L


class PlanNotRepresentable(RuntimeError):
    """A row-block / cluster plan cannot express this problem (a scatter offset beyond 8 bits, a block beyond the
    LDS budget, ...): the caller may fall back to the thread-per-entity algorithm.  Device faults, launch errors
    and out-of-memory conditions are NOT of this type and propagate."""


def check(rc: int, what: str):
    """Native return codes -> RuntimeError, like nanobind turns the reference's
    std::runtime_error into RuntimeError."""
    if rc != 0:
        msg = lib().mpcx_last_error().decode()
        raise RuntimeError(f"{what} failed ({rc}): {msg}")


_gpu_seen = False


def require_gpu():
    """the current HIP device; raises without one (there is no CPU fallback).  Called a dozen times per assembly call:
    the availability probe runs once per process, the device index is read every time (a rank may switch devices)."""
    global _gpu_seen
    import torch

    if not _gpu_seen:
        if not torch.cuda.is_available():
            raise RuntimeError(
                "dolfinx_mpc_amd needs a HIP device (MI355X / gfx950): torch.cuda.is_available() is False "
                "and there is no CPU fallback for the assembly path."
            )
        _gpu_seen = True
        if _preload_thread is not None and _preload_thread.is_alive():
            _preload_thread.join()  # (started at import: normally long finished; never run beside the first set-up kernels --
            # measured: a preload that overlaps the pattern build and the first assembly DOUBLES both, 0.10 + 0.22 -> 0.22 + 0.38 s)
    return torch.device("cuda", torch.cuda.current_device())


_preload_thread = None


def start_preload():
    """load the library's code objects -- and those of the torch kernels the plan builders use -- in the background
    (include/mpcx.h mpcx_preload): the first launch from every translation unit of libmpcx.so loads tens of MB of gfx950 code,
    which a cold box used to pay inside the first assembly.  Called once when the package is imported on a
    machine with a device, so that the loads run beside the caller's host-side problem set-up; the device is the one of
    LOCAL_RANK (the launcher's convention: one rank per GPU), else the current one; MPCX_PRELOAD=0 switches it off"""
    global _preload_thread
    if _preload_thread is not None or os.environ.get("MPCX_PRELOAD", "1") == "0" or not os.path.exists(_LIB_PATH):
        return
    multi = any(int(os.environ.get(v, "1") or 1) > 1 for v in ("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS"))
    if multi and "LOCAL_RANK" not in os.environ:
        # a launcher that does not say which GPU is this rank's (mpirun / srun use other variables): every rank would create a
        # context and load code objects on device 0 -- the loads then happen at the first call, on the device the caller selected
        return
    try:
        import torch

        if not torch.cuda.is_available():
            return
        n = torch.cuda.device_count()
        device_index = int(os.environ["LOCAL_RANK"]) % max(n, 1) if "LOCAL_RANK" in os.environ else torch.cuda.current_device()
    except Exception:  # noqa: BLE001
        return
    import threading

    def run():
        try:
            import torch

            torch.cuda.set_device(device_index)  # (the current device is per thread)
            lib().mpcx_preload(None)
            # torch loads the code object of each of ITS kernels at first use as well: the plan builders' small unique /
            # nonzero / cumsum / repeat_interleave / searchsorted / sort calls cost ~50 ms each in a fresh process (0.15 s of a
            # 0.6 s first assembly at config 2, tools/first_call_probe.py) -- touch them here, on tiny tensors
            dev = torch.device("cuda", device_index)
            for dt in (torch.int32, torch.int64):
                t = torch.arange(8, device=dev, dtype=dt)
                torch.unique(t)
                torch.nonzero(t > 3)
                torch.cumsum(t, 0)
                t.sum().item()
                torch.sort(t)
                torch.searchsorted(t, t)
                t[t % 2 == 0]
                t.max().item()
            t = torch.arange(8, device=dev, dtype=torch.int64)
            torch.repeat_interleave(t, t)
            torch.zeros(8, dtype=torch.uint8, device=dev).to(torch.int64)
            torch.zeros(8, dtype=torch.float64, device=dev).abs().max().item()
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001  (an optimisation only)
            pass

    _preload_thread = threading.Thread(target=run, name="mpcx-preload", daemon=True)
    _preload_thread.start()
