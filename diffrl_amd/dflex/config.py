"""Global switches with the reference's names (dflex/dflex/config.py:10-12)."""
no_grad = False    # True: in-place forward-only stepping, no checkpoints (sim.py:2201-2207)
check_grad = False  # accepted for compatibility; the adjoint is verified by tests/, not at run time
verify_fp = False   # True: raise on non-finite state after every step
# Not in the reference (it has only one form): True -- SemiImplicitIntegrator.forward's backward returns the cotangent of joint_q's
# quaternion coordinates WITH the component along the quaternion that the reference's literal adjoint has (dflex/dflex/quat.h:232-288,
# spatial.h:740-798; include/dsim.h: dsim_step_backward_literal).  Default False: the wrench form (no such component), which every
# rollout gradient is independent of.  $DSIM_GQ_LITERAL=1 sets the default.
import os as _os
literal_quat_grad = _os.environ.get("DSIM_GQ_LITERAL", "0") not in ("", "0")
