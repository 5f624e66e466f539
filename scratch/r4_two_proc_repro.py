"""Round 4: tests/test_gpu_dist.py failed twice (of ~6 runs) with the exchanged gradient 2 - 6 % of max|g| away from the mean
of the ranks' own gradients.  No exchange here at all: two processes share the card, each repeats the backward of ITS clip
and compares every pass with its first one, per parameter window."""
import os, sys, socket
import torch, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)


def _rank(rank, world, passes, inplace):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
    import seeded as S
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    import master_thesis_amd as mt
    from master_thesis_amd import engine as E
    E.INPLACE_GRADS = inplace
    model = mt.DFPN(mt.VGGFeatures(device=dev), **KW)
    model.load_state_dict(S.add_buffers(S.seeded_state_dict(S.dfpn_shapes(), seed=0)))
    model = model.to(dev)
    opt = model.configure_optimizers()['optimizer']
    (x, m), y, info = S.synthetic_batch(2, 2, 64, 64, seed=31)
    sl = slice(rank % 2, rank % 2 + 1)
    batch = ((x[sl].to(dev), m[sl].to(dev)), y[sl].to(dev), tuple(i[sl].to(dev) if torch.is_tensor(i) else i for i in info))
    names = [n for n, p in model.named_parameters() if 'model_vgg' not in n]
    first, bad = None, 0
    for it in range(passes):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        torch.cuda.synchronize()
        g = opt.flat_g.clone()
        if first is None:
            first, l0 = g, float(loss)
            continue
        err = float((g - first).abs().max() / first.abs().max())
        if err > 1e-3 or abs(float(loss) - l0) > 1e-5 * abs(l0):
            bad += 1
            print(f'rank {rank} pass {it}: max-normalised {err:.3e}, loss {float(loss):.7f} vs {l0:.7f}', flush=True)
            for n, (off, k) in zip(names, opt._slices):
                e = float((g[off:off + k] - first[off:off + k]).abs().max() / (first[off:off + k].abs().max() + 1e-30))
                if e > 1e-3:
                    print(f'      {n}: {e:.2e} of its own max', flush=True)
    print(f'rank {rank}: {bad} of {passes - 1} passes differ from the first', flush=True)


if __name__ == '__main__':
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    inplace = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
    mp.spawn(_rank, args=(world, passes, inplace), nprocs=world, join=True)
