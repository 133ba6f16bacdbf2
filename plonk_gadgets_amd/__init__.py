"""plonk_gadgets_amd -- MI355X (gfx950) batched constraint-evaluation engine for the gadget hot path of
dusk-network/plonk_gadgets (range_check / max_bound ladders and the scalar gadgets).

The compute path is libplonk_gadgets_hip.so (hand-written HIP behind a C ABI, include/plonk_gadgets_hip.h);
this package is the thin host side: scalar helpers, buffer plumbing on torch device tensors, and sharding."""
from .scalar import BlsScalar, bits_count, num_bits_closest_power_of_two  # noqa: F401
from .engine import Columns, Engine, Layout, NonExistingInverse, PgError  # noqa: F401
from .composer import (AllocatedScalar, StandardComposer, Variable, conditionally_select_one,  # noqa: F401
                       conditionally_select_zero, is_non_zero, max_bound, maybe_equal, range_check,
                       scalar_decomposition_gadget)
