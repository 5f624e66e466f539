"""Two gloo ranks on one card: how much do two identical backward passes differ, and does the armed pass differ more?"""
import os, sys, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)


def _rank(rank, world, port):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
    import seeded as S
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import master_thesis_amd as mt
    from master_thesis_amd import dist as mdist
    model = mt.DFPN(mt.VGGFeatures(device=dev), **KW)
    model.load_state_dict(S.add_buffers(S.seeded_state_dict(S.dfpn_shapes(), seed=0)))
    model = model.to(dev)
    opt = model.configure_optimizers()['optimizer']
    (x, m), y, info = S.synthetic_batch(2, 2, 64, 64, seed=31)
    sl = slice(rank, rank + 1)
    batch = ((x[sl].to(dev), m[sl].to(dev)), y[sl].to(dev), tuple(i[sl].to(dev) if torch.is_tensor(i) else i for i in info))

    def grads(armed):
        opt.zero_grad()
        if armed:
            mdist.arm_overlap(opt)
        loss = model.training_step(batch, 0)
        loss.backward()
        torch.cuda.synchronize()
        return opt.flat_g.clone()
    g1, g2 = grads(False), grads(False)
    names = [n for n, p in model.named_parameters() if 'model_vgg' not in n]
    if rank == 0:
        print('unarmed vs unarmed: max-normalised', float((g1 - g2).abs().max() / g1.abs().max()), flush=True)
    g3 = grads(True)
    ranges = list(mdist._pend.ranges)
    mdist.sync_gradients(opt)
    tot = opt.flat_g.clone()
    both = [torch.empty_like(g1) for _ in range(world)]
    dist.all_gather(both, g1)
    want = both[0] + both[1]
    if rank == 0:
        print('ranges sent early', ranges)
        print('armed exchange vs sum of unarmed: max-normalised', float((tot - want).abs().max() / want.abs().max()))
        for p, (off, k) in zip(opt._plist, opt._slices):
            e = float((tot[off:off + k] - want[off:off + k]).abs().max() / (want[off:off + k].abs().max() + 1e-30))
            if e > 1e-5:
                print('   window', off, k, 'err %.2e' % e, [n for n, q in model.named_parameters() if q is p])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    mp.spawn(_rank, args=(2, port), nprocs=2, join=True)
