"""`ti.ad` for the demos (demo_pour.py:157, 171; demo_pour_vel.py:81, 95).

clear_all_gradients()      Taichi zeroes the `.grad` of every field: here every live simulator's adjoint frames, primitive adjoints and action
                           buffers (`MPMSimulator.clear_grads`).
Tape(loss=env.loss.loss)   Taichi's tape clears the gradients and the loss on entry, records the kernels launched inside and, on exit, sets
                           loss.grad = 1 and replays their adjoints in reverse - for the demos those are the `compute_loss(f)` calls, whose adjoint seeds
                           `x.grad[f]` and the controlled primitive's pose / velocity adjoints (loss_pour.py:130-140).  The losses of this build add
                           those seeds while they compute the value when they are `recording` (losses.*.tape()); the loss is a plain sum over
                           frames, so the result is the same.  `loss` is the field-like number the loss object hands out; it knows its owner."""
import contextlib


def _simulators():
    from softmac_amd.engine import mpm_simulator
    return list(mpm_simulator.LIVE_SIMULATORS)


def clear_all_gradients():
    for sim in _simulators():
        sim.clear_grads()


@contextlib.contextmanager
def Tape(loss, clear_gradients=True, validation=False, grad_check=None):
    owner = getattr(loss, "owner", None)
    if owner is None:
        raise TypeError("ti.ad.Tape(loss=...): pass the loss object's `.loss` (e.g. env.loss.loss)")
    if clear_gradients:
        clear_all_gradients()
    owner.clear()
    with owner.tape():
        yield
