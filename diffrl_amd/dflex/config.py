"""Global switches with the reference's names (dflex/dflex/config.py:10-12)."""
no_grad = False    # True: in-place forward-only stepping, no checkpoints (sim.py:2201-2207)
check_grad = False  # accepted for compatibility; the adjoint is verified by tests/, not at run time
verify_fp = False   # True: raise on non-finite state after every step
