// experiment.hpp -- the fence around this tree's build options.
//
// Every -DPG_... option below is an A/B knob or a timing build of tools/ab_emit.py (what each one measured: DESIGN.md section 3,
// profiles/NOTES_r0*.md).  All of them leave the OUTPUT as it is; the builds that do not (a launch stripped of its memory
// traffic, an emitter without its multiplications, this round's rejected schedules) are not in the sources at all but patches
// under tools/patches/, applied to a copy of csrc/ by tools/ab_emit.py.  The shipped library is built with NONE of them
// (plonk_gadgets_amd/build.py: tests/test_product_boundaries.py checks its command line, and that this list names every option
// the sources test), and defining one without -DPG_EXPERIMENT does not compile.
#pragma once

#if !defined(PG_EXPERIMENT) &&                                                                                                  \
    (defined(PG_NT_STORES) || defined(PG_FULL_BARRIERS) || defined(PG_EMIT_WAVES_PER_SIMD) || defined(PG_XCD_REMAP) ||          \
     defined(PG_UNALIGNED_SWEEPS) || defined(PG_ROWS_WAVES_PER_SIMD) || defined(PG_ROWS_SETPRIO) || defined(PG_INVERT_FERMAT) || \
     defined(PG_INV_GRP) || defined(PG_RC_W) || defined(PG_MB_W) || defined(PG_MIX_VARS_VGPRS) || defined(PG_MIX_STAMPS) ||     \
     defined(PG_QUEUE_MAX) || defined(PG_PERM_LDS_PAD) || defined(PG_GRID_BLOCKS_PER_CU) || defined(PG_PLAN_TWO_LAUNCHES) ||    \
     defined(PG_INV_LANES_PER_CU) || defined(PG_INV_MAX_PER_LANE) || defined(PG_SIDE_STREAM_NORMAL_PRIORITY) || defined(PG_VAR_SWEEP_SINGLE) || \
     defined(PG_TILE_SPREAD) || defined(PG_MIX_EARLY_TILES) || defined(PG_MAT_THREADS) || defined(PG_MAT_UNROLL) || defined(PG_MAT_LOADS) || defined(PG_PERM_LADDER_ROWS) || defined(PG_PERM_LADDER_LDS) || defined(PG_MAT_GRID_PER_CU) || defined(PG_EMIT_GRID_BLOCKS_PER_CU))
#error "a PG_... build option is defined without -DPG_EXPERIMENT: the shipped library is built with none of them (csrc/experiment.hpp)"
#endif
