"""Round 4: which FORWARD tensor of the DFPN differs between passes when several processes share the card?"""
import os, sys
import torch, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)


def _rank(rank, world, passes):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
    import seeded as S
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    import master_thesis_amd as mt
    from master_thesis_amd import engine as E
    model = mt.DFPN(mt.VGGFeatures(device=dev), **KW)
    model.load_state_dict(S.add_buffers(S.seeded_state_dict(S.dfpn_shapes(), seed=0)))
    model = model.to(dev)
    (x, m), y, info = S.synthetic_batch(2, 2, 64, 64, seed=31)
    sl = slice(rank % 2, rank % 2 + 1)
    x, m = x[sl].to(dev), m[sl].to(dev)
    t, r = model.get_indexes(2)
    # record every conv-engine launch's output (name, tensor clone) through the profiling hook of the engine
    outs_first, bad = None, 0
    orig = E.conv_forward
    log = []

    def spy(spec, xv, x_shape, w, b, out=None, **kw):
        src = getattr(xv, '_src', None)
        if src is not None:
            log.append((f'IN of {spec.kind}{spec.cin}->{spec.cout} k{spec.k} {tuple(src.shape)}', src.detach().clone()))
        o = orig(spec, xv, x_shape, w, b, out=out, **kw)
        log.append((f'{spec.kind}{spec.cin}->{spec.cout} k{spec.k} s{spec.stride} {tuple(o.shape)}', o.detach().clone()))
        return o
    E.conv_forward = spy
    for it in range(passes):
        log.clear()
        with torch.no_grad():
            corr, f16, f64, f256 = model(x[:, :, t].contiguous(), m[:, :, t].contiguous(), x[:, :, r], m[:, :, r])
        torch.cuda.synchronize()
        cur = [(n, o) for n, o in log] + [('corr', corr.clone()), ('flow_16', f16.clone()), ('flow_64', f64.clone()), ('flow_256', f256.clone())]
        if outs_first is None:
            outs_first = cur
            continue
        diffs = [(i, n, float((a - b).abs().max()), float(b.abs().max())) for i, ((n, a), (_, b)) in enumerate(zip(cur, outs_first)) if not torch.equal(a, b)]
        if diffs:
            bad += 1
            i, n, e, mx = diffs[0]
            a, b = cur[i][1], outs_first[i][1]
            idx = (a != b).nonzero()
            lo, hi = idx.min(dim=0).values.tolist(), idx.max(dim=0).values.tolist()
            print(f'rank {rank} pass {it}: first differing tensor #{i} {n}: {idx.shape[0]} of {a.numel()} elements differ, index box {lo} .. {hi}', flush=True)
            print(f'rank {rank} pass {it}: {len(diffs)} of {len(cur)} tensors differ; first: #{i} {n}: max abs diff {e:.3e} (max |.| {mx:.3e}); '
                  f'last: {diffs[-1][1]} {diffs[-1][2]:.3e}', flush=True)
    print(f'rank {rank}: {bad} of {passes - 1} forward passes differ from the first ({len(outs_first)} tensors compared)', flush=True)


if __name__ == '__main__':
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    mp.spawn(_rank, args=(world, passes), nprocs=world, join=True)
