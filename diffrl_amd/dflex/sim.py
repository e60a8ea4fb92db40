"""`SemiImplicitIntegrator`: the operator boundary of the reference (dflex/dflex/sim.py:2157-2221).

`forward(model, state, dt, substeps, mass_matrix_freq)` advances all environments by one control step
with ONE fused HIP launch (and registers ONE autograd node whose backward is one fused adjoint
launch), instead of ~100 kernel launches + ~280 tensor allocations per step in the reference
(SURVEY.md section 3.2)."""
import torch

from ..engine import SimStep
from . import config
from .model import Model, ModelBuilder, State  # noqa: F401  (re-exported like dflex.sim)


class SemiImplicitIntegrator:
    def __init__(self):
        pass

    def forward(self, model: Model, state_in: State, dt: float, substeps: int, mass_matrix_freq: int) -> State:
        eng = model.engine()
        mact = model.muscle_activation if model.muscle_count else None
        out = State(act_like=model.joint_qd, model=model)
        if config.no_grad:
            with torch.no_grad():
                q, qd, _ = eng.forward(state_in.joint_q.contiguous(), state_in.joint_qd.contiguous(),
                                       state_in.joint_act.contiguous(), mact.contiguous() if mact is not None else None,
                                       float(dt), int(substeps), int(mass_matrix_freq), False)
            out.joint_q, out.joint_qd = q, qd
        else:
            out.joint_q, out.joint_qd = SimStep.apply(eng, float(dt), int(substeps), int(mass_matrix_freq),
                                                      state_in.joint_q, state_in.joint_qd, state_in.joint_act, mact)
        out._xf_q = eng.last_q_in if not config.no_grad else None   # None in no-grad mode (State.body_X_sc then derives from out.joint_q)
        if config.verify_fp and not (torch.isfinite(out.joint_q).all() and torch.isfinite(out.joint_qd).all()):
            raise FloatingPointError("non-finite state after SemiImplicitIntegrator.forward")
        return out
