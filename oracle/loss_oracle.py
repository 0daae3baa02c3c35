"""TEST INFRASTRUCTURE ONLY - never imported by the product (softmac_amd/).

CPU restatement (numpy) of the reference's chamfer loss and of what its Taichi tape leaves in x.grad
(/root/reference/softmac/engine/losses/loss_pour.py:44-70 `chamfer_closest` + `compute_chamfer_loss_kernel`, and the
generated adjoint run by `compute_loss_kernel_grad` :130-140; loss_grip.py:45-68 is the same code):

    for i: nn_cur[i] = first j minimising |x_i - t_j|^2          (strict `<` while j runs upwards)
    for j: nn_tar[j] = first i minimising |x_i - t_j|^2
    L = sum_i |x_i - t_nn_cur[i]|^2 + sum_j |x_nn_tar[j] - t_j|^2
    dL/dx_i = 2 (x_i - t_nn_cur[i]) + sum_{j : nn_tar[j] = i} 2 (x_i - t_j)       (the indices are constants of the adjoint)

PARITY UNPINNED: the reference ships no golden value for its losses; this file is checked by finite
differences and against scipy's k-d tree in tests/test_losses.py."""
import numpy as np


def chamfer(x, target, chunk=1024):
    """Brute force, exactly the reference's double loop (first minimum).  Returns loss, grad (n,3), nn_cur, nn_tar."""
    x = np.asarray(x, dtype=np.float64)
    t = np.asarray(target, dtype=np.float64)
    nn_cur = np.empty(len(x), dtype=np.int64)
    d_cur = np.empty(len(x))
    for s in range(0, len(x), chunk):
        d = ((x[s:s + chunk, None, :] - t[None, :, :]) ** 2).sum(-1)
        nn_cur[s:s + chunk] = d.argmin(1)                       # argmin returns the first minimum
        d_cur[s:s + chunk] = d.min(1)
    nn_tar = np.empty(len(t), dtype=np.int64)
    d_tar = np.empty(len(t))
    for s in range(0, len(t), chunk):
        d = ((t[s:s + chunk, None, :] - x[None, :, :]) ** 2).sum(-1)
        nn_tar[s:s + chunk] = d.argmin(1)
        d_tar[s:s + chunk] = d.min(1)
    loss = d_cur.sum() + d_tar.sum()
    grad = 2.0 * (x - t[nn_cur])
    np.add.at(grad, nn_tar, 2.0 * (x[nn_tar] - t))
    return loss, grad, nn_cur, nn_tar


def chamfer_kdtree(x, target):
    """Same quantity through scipy's k-d tree (for sizes where the double loop is too slow; ties are not ordered)."""
    from scipy.spatial import cKDTree
    x = np.asarray(x, dtype=np.float64)
    t = np.asarray(target, dtype=np.float64)
    d_cur, nn_cur = cKDTree(t).query(x)
    d_tar, nn_tar = cKDTree(x).query(t)
    loss = (d_cur ** 2).sum() + (d_tar ** 2).sum()
    grad = 2.0 * (x - t[nn_cur])
    np.add.at(grad, nn_tar, 2.0 * (x[nn_tar] - t))
    return loss, grad, nn_cur, nn_tar
