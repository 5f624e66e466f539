"""Gradient parity helpers shared by the GPU test modules (test infrastructure: uses the oracle)."""
import torch


def oracle_grads(step, sd, batch, dtype, names):
    """Gradients of the oracle's training step in ``dtype`` (fp64 = ground truth)."""
    sd2 = {k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()}
    for n in names:
        sd2[n] = sd2[n].clone().requires_grad_(True)
    cast = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t  # noqa: E731
    (x, m), y, info = batch
    loss, _ = step(sd2, ((cast(x), cast(m)), cast(y), tuple(cast(i) for i in info)))
    return [g.double() for g in torch.autograd.grad(loss, [sd2[n] for n in names])]


def check_grads_vs_fp64(model, step, sd, batch, names, total_bound=3e-3, cpu32=True, truth=torch.float64):
    """The training step has ~10^6 kinks (ReLU / LeakyReLU signs, clamp, floor
    in the warps).  Any fp32 implementation lands a handful of pre-activations
    that are within rounding of 0 on the other side than exact arithmetic does,
    and at the tiny resolution of the small test clips (16x16 trunk) ONE such flip
    moves that layer's gradient by O(1e-2).  So: ground truth = the oracle in
    fp64.  Measured on the small clips (relative L2 over ALL parameters): DFPN step
    CPU-fp32 2.0e-3, HIP 1.9e-3; CHN step CPU-fp32 0.9e-4, HIP fp32-MFMA 1.0e-3,
    bf16x6 1.0e-3, bf16x3 1.6e-3 (a fixed-order MFMA accumulation chain carries
    ~2.5x the rounding noise of the CPU's blocked summation, i.e. a few more
    flips).  Bounds: over all parameters err <= max(total_bound, 3 * err_cpu32); per
    tensor err <= max(2e-2, 3 * err_cpu32) except for at most 2 % of the tensors
    (isolated kink flips), none of which may exceed 0.2.
    ``cpu32=False`` skips the fp32 run of the oracle (half the CPU time): the bounds are then the absolute ones.
    ``truth``: dtype of the oracle run that serves as ground truth (fp32 where fp64 is unaffordable and the fp32 oracle's own
    deviation from fp64 has been measured to be far below the bound).
    Returns (relative L2 of the HIP gradient, of the fp32-CPU oracle's) against fp64."""
    g64 = oracle_grads(step, sd, batch, truth, names)
    g32 = oracle_grads(step, sd, batch, torch.float32, names) if cpu32 else g64
    params = dict(model.named_parameters())
    bad, num_g, num_c, den = [], 0.0, 0.0, 0.0
    for n, a32, a64 in zip(names, g32, g64):
        assert params[n].grad is not None, n
        gg = params[n].grad.double().cpu()
        ref = float(a64.norm()) + 1e-30
        e_gpu, e_cpu = float((gg - a64).norm()) / ref, float((a32 - a64).norm()) / ref
        num_g += float((gg - a64).norm()) ** 2
        num_c += float((a32 - a64).norm()) ** 2
        den += ref ** 2
        assert e_gpu < 0.2, (n, e_gpu, e_cpu)
        if e_gpu > max(2e-2, 3.0 * e_cpu):
            bad.append((n, e_gpu, e_cpu))
    assert len(bad) <= max(2, len(names) // 50), bad[:8]
    tot_g, tot_c = (num_g / den) ** 0.5, (num_c / den) ** 0.5
    assert tot_g <= max(total_bound, 3.0 * tot_c), (tot_g, tot_c)
    return tot_g, tot_c
