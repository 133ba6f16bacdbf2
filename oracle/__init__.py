"""CPU oracle for the gadget path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (plonk_gadgets_amd) never does.
"""
