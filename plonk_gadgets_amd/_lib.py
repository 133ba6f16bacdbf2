"""ctypes loader for libplonk_gadgets_hip.so.  Fails loudly when the library is missing: there is no CPU path."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libplonk_gadgets_hip.so")


class Scalar(C.Structure):
    """pg_scalar == BlsScalar's inner [u64; 4]"""
    _fields_ = [("l", C.c_uint64 * 4)]

    @staticmethod
    def of(limbs) -> "Scalar":
        s = Scalar()
        for i in range(4):
            s.l[i] = int(limbs[i])
        return s

    def limbs(self):
        return [int(self.l[i]) for i in range(4)]


class AllocatedScalarC(C.Structure):
    _fields_ = [("var", C.c_uint64), ("scalar", Scalar)]


class ColumnsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")]


class FullColumnsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
                                          "q_variable_group_add", "w_4", "w_l_value", "w_r_value", "w_o_value",
                                          "w_4_value")]


class ShardC(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32)] + [(n, C.c_uint64) for n in
                                                                 ("lo", "hi", "gate_base", "var_base", "n_gates", "n_vars")]


class PackedC(C.Structure):
    _fields_ = [("q_words", C.c_uint64 * 5), ("w_words", C.c_uint64 * 3), ("var_words", C.c_uint64),
                ("total_words", C.c_uint64), ("n_gates", C.c_uint64), ("n_vars", C.c_uint64)]


class LayoutC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("num_bits", "gates_per_item", "vars_per_item", "n_gates", "n_vars")]


# every symbol include/plonk_gadgets_hip.h declares: name -> (restype, argtypes)
_P = C.POINTER
SIGNATURES = {
    "pg_engine_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "pg_engine_destroy": (None, [C.c_void_p]),
    "pg_engine_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pg_status_string": (C.c_char_p, [C.c_int]),
    "pg_last_error": (C.c_char_p, []),
    "pg_build_arch": (C.c_char_p, []),
    "pg_scalar_from_u64": (None, [C.c_uint64, _P(Scalar)]),
    "pg_scalar_from_canonical": (None, [_P(C.c_uint64), _P(Scalar)]),
    "pg_scalar_to_canonical": (None, [_P(Scalar), _P(C.c_uint64)]),
    "pg_scalar_add": (None, [_P(Scalar), _P(Scalar), _P(Scalar)]),
    "pg_scalar_sub": (None, [_P(Scalar), _P(Scalar), _P(Scalar)]),
    "pg_scalar_neg": (None, [_P(Scalar), _P(Scalar)]),
    "pg_scalar_mul": (None, [_P(Scalar), _P(Scalar), _P(Scalar)]),
    "pg_scalar_invert": (C.c_int, [_P(Scalar), _P(Scalar)]),
    "pg_scalar_invert_fermat": (C.c_int, [_P(Scalar), _P(Scalar)]),
    "pg_bits_count": (C.c_uint64, [_P(Scalar)]),
    "pg_num_bits_closest_power_of_two": (C.c_uint64, [_P(Scalar)]),
    "pg_range_check_layout": (C.c_int, [_P(Scalar), _P(Scalar), C.c_uint64, _P(LayoutC)]),
    "pg_range_check_batch": (C.c_int, [C.c_void_p, _P(Scalar), _P(Scalar), C.c_void_p, C.c_uint64, C.c_uint64,
                                       C.c_uint64, _P(ColumnsC), C.c_void_p, C.c_void_p]),
    "pg_scalars_from_canonical_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, _P(C.c_uint64), C.c_void_p]),
    "pg_scalars_to_canonical_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "pg_range_check_structure_batch": (C.c_int, [C.c_void_p, _P(Scalar), _P(Scalar), C.c_uint64, C.c_uint64, C.c_uint64, _P(ColumnsC),
                                                 C.c_void_p]),
    "pg_range_check_values_batch": (C.c_int, [C.c_void_p, _P(Scalar), _P(Scalar), C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "pg_max_bound_values_batch": (C.c_int, [C.c_void_p, _P(Scalar), C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "pg_max_bound_ragged_values_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]),
    "pg_scalar_mix_values_batch": (C.c_int, [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pg_range_check_allocated_batch": (C.c_int, [C.c_void_p, _P(Scalar), _P(Scalar), C.c_void_p, C.c_void_p, C.c_uint64,
                                                 C.c_uint64, C.c_uint64, _P(ColumnsC), C.c_void_p, C.c_void_p]),
    "pg_max_bound_allocated_batch": (C.c_int, [C.c_void_p, _P(Scalar), C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                               C.c_uint64, _P(ColumnsC), C.c_void_p, C.c_void_p]),
    "pg_scalar_decomposition_layout": (C.c_int, [C.c_uint64, C.c_uint64, _P(LayoutC)]),
    "pg_scalar_decomposition_batch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                                C.c_uint64, _P(ColumnsC), C.c_void_p, C.c_void_p]),
    "pg_scalar_decomposition_gadget": (C.c_int, [C.c_void_p, C.c_uint64, _P(AllocatedScalarC), _P(C.c_uint64),
                                                 _P(C.c_uint64)]),
    "pg_max_bound_layout": (C.c_int, [_P(Scalar), C.c_uint64, _P(LayoutC)]),
    "pg_max_bound_batch": (C.c_int, [C.c_void_p, _P(Scalar), C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                     _P(ColumnsC), C.c_void_p, C.c_void_p]),
    "pg_max_bound_ragged_plan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           _P(LayoutC), C.c_void_p]),
    "pg_max_bound_ragged_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_uint64, C.c_uint64, _P(ColumnsC), C.c_void_p, C.c_void_p]),
    "pg_conditionally_select_zero_batch": (C.c_int, [C.c_void_p] * 5 + [C.c_uint64] * 3 + [_P(ColumnsC), C.c_void_p,
                                                                                            C.c_void_p]),
    "pg_conditionally_select_one_batch": (C.c_int, [C.c_void_p] * 5 + [C.c_uint64] * 3 + [_P(ColumnsC), C.c_void_p,
                                                                                           C.c_void_p]),
    "pg_maybe_equal_batch": (C.c_int, [C.c_void_p] * 5 + [C.c_uint64] * 3 + [_P(ColumnsC), C.c_void_p, C.c_void_p]),
    "pg_is_non_zero_plan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                      _P(LayoutC), _P(C.c_uint64), C.c_void_p]),
    "pg_is_non_zero_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                       C.c_uint64, C.c_uint64, C.c_uint64, _P(ColumnsC), C.c_void_p]),
    "pg_max_bound_ragged_plan_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]),
    "pg_scalar_mix_plan_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "pg_plan_result": (C.c_int, [C.c_void_p, _P(LayoutC), _P(C.c_uint64)]),
    "pg_scalar_mix_plan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     _P(LayoutC), _P(C.c_uint64), C.c_void_p]),
    "pg_composer_create": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, _P(C.c_void_p)]),
    "pg_composer_destroy": (None, [C.c_void_p]),
    "pg_composer_circuit_size": (C.c_uint64, [C.c_void_p]),
    "pg_composer_num_variables": (C.c_uint64, [C.c_void_p]),
    "pg_composer_zero_var": (C.c_uint64, [C.c_void_p]),
    "pg_composer_columns": (C.c_int, [C.c_void_p, _P(ColumnsC)]),
    "pg_composer_reserve": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "pg_composer_auto_grow": (C.c_int, [C.c_void_p, C.c_int]),
    "pg_composer_spread_columns": (C.c_int, [C.c_void_p, C.c_uint64]),
    "pg_composer_gate_capacity": (C.c_uint64, [C.c_void_p]),
    "pg_composer_var_capacity": (C.c_uint64, [C.c_void_p]),
    "pg_composer_sync": (C.c_int, [C.c_void_p]),
    "pg_composer_clear_witness": (C.c_int, [C.c_void_p]),
    "pg_composer_refresh_stats": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64), _P(C.c_int)]),
    "pg_composer_queue": (C.c_int, [C.c_void_p, C.c_int]),
    "pg_composer_flush": (C.c_int, [C.c_void_p]),
    "pg_composer_queue_stats": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64), _P(C.c_uint64)]),
    "pg_composer_add_input": (C.c_int, [C.c_void_p, _P(Scalar), _P(C.c_uint64)]),
    "pg_composer_add_witness_to_circuit_description": (C.c_int, [C.c_void_p, _P(Scalar), _P(C.c_uint64)]),
    "pg_composer_constrain_to_constant": (C.c_int, [C.c_void_p, C.c_uint64, _P(Scalar), _P(Scalar)]),
    "pg_composer_assert_equal": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "pg_composer_poly_gate": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64] + [_P(Scalar)] * 6),
    "pg_composer_add": (C.c_int, [C.c_void_p, _P(Scalar), C.c_uint64, _P(Scalar), C.c_uint64, _P(Scalar), _P(Scalar),
                                  _P(C.c_uint64)]),
    "pg_composer_mul": (C.c_int, [C.c_void_p, _P(Scalar), C.c_uint64, C.c_uint64, _P(Scalar), _P(Scalar), _P(C.c_uint64)]),
    "pg_composer_mul_gate": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64] + [_P(Scalar)] * 4),
    "pg_composer_boolean_gate": (C.c_int, [C.c_void_p, C.c_uint64]),
    "pg_allocated_scalar_allocate": (C.c_int, [C.c_void_p, _P(Scalar), _P(AllocatedScalarC)]),
    "pg_range_check": (C.c_int, [C.c_void_p, _P(Scalar), _P(Scalar), _P(AllocatedScalarC), _P(C.c_uint64)]),
    "pg_max_bound": (C.c_int, [C.c_void_p, _P(Scalar), _P(AllocatedScalarC), _P(C.c_uint64), _P(C.c_uint64)]),
    "pg_conditionally_select_zero": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, _P(C.c_uint64)]),
    "pg_conditionally_select_one": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, _P(C.c_uint64)]),
    "pg_is_non_zero": (C.c_int, [C.c_void_p, C.c_uint64, _P(Scalar)]),
    "pg_maybe_equal": (C.c_int, [C.c_void_p, _P(AllocatedScalarC), _P(AllocatedScalarC), _P(C.c_uint64)]),
    "pg_composer_range_check_batch": (C.c_int, [C.c_void_p, _P(Scalar), _P(Scalar), C.c_void_p, C.c_uint64, C.c_void_p]),
    "pg_composer_add_input_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, _P(C.c_uint64)]),
    "pg_composer_range_check_allocated_batch": (C.c_int, [C.c_void_p, _P(Scalar), _P(Scalar), C.c_void_p, C.c_void_p, C.c_uint64,
                                                          C.c_void_p]),
    "pg_composer_max_bound_batch": (C.c_int, [C.c_void_p, _P(Scalar), C.c_void_p, C.c_uint64, C.c_void_p, _P(C.c_uint64)]),
    "pg_composer_max_bound_allocated_batch": (C.c_int, [C.c_void_p, _P(Scalar), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                                        _P(C.c_uint64)]),
    "pg_composer_scalar_decomposition_batch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pg_composer_conditionally_select_zero_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pg_composer_conditionally_select_one_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pg_composer_maybe_equal_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pg_composer_max_bound_ragged_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "pg_composer_is_non_zero_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, _P(C.c_uint64)]),
    "pg_composer_scalar_mix_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                               C.c_void_p, C.c_void_p, _P(C.c_uint64)]),
    "pg_composer_poly_gate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _P(Scalar), _P(Scalar), _P(Scalar),
                                              _P(Scalar), _P(Scalar), C.c_uint64]),
    "pg_composer_add_batch": (C.c_int, [C.c_void_p, _P(Scalar), C.c_void_p, _P(Scalar), C.c_void_p, _P(Scalar), C.c_uint64, C.c_void_p]),
    "pg_composer_mul_batch": (C.c_int, [C.c_void_p, _P(Scalar), C.c_void_p, C.c_void_p, _P(Scalar), C.c_uint64, C.c_void_p]),
    "pg_composer_constrain_to_constant_batch": (C.c_int, [C.c_void_p, C.c_void_p, _P(Scalar), C.c_uint64]),
    "pg_composer_boolean_gate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "pg_composer_copy_out": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _P(ColumnsC)]),
    "pg_composer_read_value": (C.c_int, [C.c_void_p, C.c_uint64, _P(Scalar)]),
    "pg_composer_check": (C.c_int, [C.c_void_p, _P(C.c_int64)]),
    "pg_composer_dense_pi": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pg_composer_materialize": (C.c_int, [C.c_void_p, _P(FullColumnsC)]),
    "pg_composer_permutation": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "pg_composer_permutation_reserve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "pg_check_rows": (C.c_int, [C.c_void_p, _P(ColumnsC), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _P(C.c_int64),
                                C.c_void_p]),
    "pg_shard_range": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, _P(C.c_uint64), _P(C.c_uint64)]),
    "pg_range_check_shard_layout": (C.c_int, [_P(Scalar), _P(Scalar), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                              _P(ShardC)]),
    "pg_range_check_sharded_batch": (C.c_int, [C.c_void_p, _P(Scalar), _P(Scalar), C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                               C.c_uint64, C.c_uint64, _P(ColumnsC), C.c_void_p, _P(ShardC), C.c_void_p]),
    "pg_packed_layout": (C.c_int, [C.c_uint64, C.c_uint64, _P(PackedC)]),
    "pg_columns_in_packed": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, _P(ColumnsC)]),
    "pg_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pg_comm_library": (C.c_char_p, []),
    "pg_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _P(C.c_void_p)]),
    "pg_comm_adopt": (C.c_int, [C.c_void_p, C.c_void_p, _P(C.c_void_p)]),
    "pg_comm_destroy": (None, [C.c_void_p]),
    "pg_comm_rank": (C.c_uint32, [C.c_void_p]),
    "pg_comm_world": (C.c_uint32, [C.c_void_p]),
    "pg_allgather_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pg_allgather_columns": (C.c_int, [C.c_void_p, _P(ColumnsC), C.c_uint64, C.c_uint64, _P(ColumnsC), C.c_void_p]),
    "pg_fill_columns": (C.c_int, [C.c_void_p, C.POINTER(ColumnsC), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "pg_fill_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]),
    "pg_columns_slab_layout": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pg_scalar_mix_batch": (C.c_int, [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                                          C.c_uint64, _P(ColumnsC), C.c_void_p, C.c_void_p]),
    "pg_scalar_mix_planned_batch": (C.c_int, [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                                                  C.c_uint64, C.c_uint64, _P(ColumnsC), C.c_void_p, C.c_void_p]),
}

# the gather pipeline's consumer callback (pg_chunk_consumer)
CHUNK_CONSUMER = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint32, _P(ColumnsC), C.c_uint64, C.c_uint64, C.c_void_p)
SIGNATURES.update({
    "pg_range_check_gather_pipeline_create": (C.c_int, [C.c_void_p, _P(Scalar), _P(Scalar), C.c_uint64, C.c_uint32, _P(C.c_void_p)]),
    "pg_range_check_gather_pipeline_bytes_per_chunk": (C.c_uint64, [C.c_void_p]),
    "pg_range_check_gather_pipeline_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, CHUNK_CONSUMER,
                                                     C.c_void_p, C.c_void_p]),
    "pg_range_check_gather_pipeline_destroy": (None, [C.c_void_p]),
    "pg_max_bound_ragged_sharded_plan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                                   C.c_uint64, _P(ShardC), _P(C.c_uint64), _P(C.c_uint64), C.c_void_p]),
    "pg_max_bound_ragged_sharded_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_uint64, C.c_uint64, _P(ColumnsC), C.c_void_p, _P(ShardC), C.c_void_p]),
})

_lib = None


def load():
    """Load the library (importing torch first so that its bundled HIP runtime -- the one that owns the
    tensors' device memory -- is the libamdhip64.so.7 the library binds to)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m plonk_gadgets_amd.build` "
            "(or __graft_entry__.build()). plonk_gadgets_amd has no CPU fallback.")
    import torch  # noqa: F401  (loads libamdhip64.so.7)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(lib, name)  # AttributeError here == header/library mismatch
        f.restype, f.argtypes = res, args
    _lib = lib
    return lib
