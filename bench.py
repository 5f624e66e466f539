#!/usr/bin/env python
"""Throughput of the DFPN / CHN training step on synthetic 256x256x5 clips.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed / RCCL for N > 1: launched by
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``);
clips are sharded over ranks (weak scaling: 8 clips per GPU), one flat gradient
all-reduce per step.  A step = forward + losses + backward + all-reduce + Adam
on one batch already resident in HBM.  Rank 0 prints ONE JSON line.

``value`` is BASELINE config 2 (DFPN train step, frames_n=5, 256x256, batch 8
per GPU); the CHN train step with the DFPN aligner (config 3) is timed in the
same run and reported under ``chn_train_step``.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md 8d / BASELINE.md section 4: algorithmic conv+GEMM work per step
TFLOP_PER_STEP = {'dfpn': 43.31, 'chn': 30.71}
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32-input MFMA peak
PEAK_16BIT_MFMA_TFLOPS = 2500.0      # dense bf16 / f16 MFMA peak (same guide)
PEAK_HBM_GBS = 8000.0


def cfg_tag(args, name):
    return name + ': ' if (args.batch, args.frames, args.size) == (8, 5, 256) else ''


def arithmetic_label():
    from master_thesis_amd import engine as E
    return (f'fp32 tensors; conv MFMA arithmetic {E.PRECISION} on the halo-tile kernels '
            f'({"2-term fp16 split of power-of-two scaled operands, 3 f16 MFMA per product, ~7e-8 rel" if E.PRECISION == "f16x3" else E.PRECISION}), '
            f'bf16x6 (3-term bf16 split, fp32-equivalent) on the strided/transposed layers, '
            f'{E.vgg_precision()} on the never-differentiated VGG passes; fp32 accumulation everywhere')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8, help='clips per GPU')
    ap.add_argument('--frames', type=int, default=5)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-chn', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of a captured hipGraph')
    ap.add_argument('--cpu-seconds', type=float, default=30.0)
    return ap.parse_args()


def timed_steps(step_fn, steps, warmup, world):
    import torch.distributed as dist
    for _ in range(warmup):
        step_fn()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    verbose = os.environ.get('MT355_BENCH_VERBOSE') == '1'
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn()
        if verbose:     # (perturbs the measurement: debugging aid only)
            torch.cuda.synchronize()
            print(f'[rank {os.environ.get("RANK", "0")}] step {i}: {time.perf_counter() - t0:.3f} s', file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt


def kernel_profile(step_fn):
    """One extra step with HIP events around every conv-engine launch (on the
    stream they run on): per-kernel launches, total ms and algorithmic FLOP."""
    from master_thesis_amd import engine as E
    E.PROFILE = []
    step_fn()
    torch.cuda.synchronize()
    agg = {}
    for name, flops, e0, e1 in E.PROFILE:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += flops
    E.PROFILE = None
    return agg


def corr_roofline(b_pairs=2048):
    """Stand-alone correlation (K5) micro-benchmark on >= 2048 pairs in flight
    (SURVEY.md 8d): algorithmic bytes = 1.31 MB per pair."""
    from master_thesis_amd import ops
    b, r, p, c = b_pairs // 4, 4, 256, 512
    ft = torch.randn((b, 16, 16, c), device='cuda')
    fr = torch.randn((b * r, 16, 16, c), device='cuda')
    vt = (torch.rand((b, p), device='cuda') > 0.1).float()
    vr = (torch.rand((b * r, p), device='cuda') > 0.1).float()
    for _ in range(20):      # reach steady-state clocks: the launch is < 1 ms
        ops.correlation(ft, fr, vt, vr, b, r)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        ops.correlation(ft, fr, vt, vr, b, r)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    bytes_ = (b * c * p + b * r * c * p + (b + b * r) * p + b * r * p * p) * 4.0
    gbs = bytes_ / (ms * 1e-3) / 1e9
    flops = 2.0 * b * r * p * p * c
    return {'kernel': 'corr_fused_kernel (mt_corr_masked)', 'pairs': b * r, 'ms': ms,
            'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS,
            'tflops': flops / (ms * 1e-3) / 1e12}


PMC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_hbm_traffic.txt')


def pmc_traffic(label):
    """HBM bytes per launch of the kernel ``label`` from the committed rocprofv3 PMC
    passes (profiles/: FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 correction
    applied there); None when the profile holds no row for it.  The counters cannot be
    collected inside this process, so this is the profile of the same workload."""
    none = {'traffic': None}
    try:
        rows = open(PMC_FILE).read().splitlines()
    except OSError:
        return none
    want = label.replace(',bf16', ',0').replace(',f16', ',1').replace(' ', '')
    for row in rows:
        if row.startswith('#') or '>' not in row:
            continue
        name, rest = row.rsplit('>', 1)
        if (name + '>').replace(' ', '').startswith(want.rstrip('>')):   # (the label may omit trailing template arguments)
            f = rest.strip(',').split(',')
            return {'traffic': float(f[3]) * 1e6, 'traffic_unit': 'bytes/launch (PMC, (2*FETCH_SIZE+WRITE_SIZE)*1024)',
                    'traffic_source': 'profiles/r01_pmc_hbm_traffic.txt', 'traffic_launches': int(f[0])}
    return none


def host_cores():
    """CPU cores this process may really use: affinity mask and cgroup quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(frames, size, budget_s):
    """The CPU oracle (port of the reference path) timed on this box's host
    cores on a bounded sample: TWO clips of the 8-clip batch, one DFPN training
    step (forward + loss + backward), fp32, all cores of the cgroup share."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import seeded as S
    from oracle import ref_torch as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = S.seeded_state_dict(S.dfpn_shapes())
    S.add_buffers(sd)
    names = [k for k in sd if 'model_vgg' not in k and k not in ('mean', 'std')]
    for n in names:
        sd[n] = sd[n].clone().requires_grad_(True)
    clips = 2
    batch = S.synthetic_batch(clips, frames, size, size, seed=1)
    t0 = time.perf_counter()
    loss, _ = O.dfpn_training_step(sd, batch)
    torch.autograd.grad(loss, [sd[n] for n in names])
    dt = time.perf_counter() - t0
    return {'value': clips * frames / dt, 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'{clips} clips x {frames} frames x {size}x{size}, one DFPN training step (fwd+loss+bwd) on the '
                      f'oracle (oracle/ref_torch.py, torch fp32 CPU ops), {dt:.1f} s'}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # MT355_BENCH_BACKEND=gloo rehearses the N > 1 code path with several ranks on ONE card
    # (ranks share device local % device_count; gloo stages CUDA tensors through the host)
    backend = os.environ.get('MT355_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    import master_thesis_amd as mt
    from master_thesis_amd import dist as mdist
    from master_thesis_amd.synthetic import make_batch
    mt._lib.load()

    kw = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
    torch.manual_seed(0)
    vgg = mt.VGGFeatures(device=dev)
    dfpn = mt.DFPN(vgg, **kw).to(dev)
    batch = make_batch(args.batch, args.frames, args.size, args.size, dev, seed=1234 + rank)
    frames_per_step = args.batch * args.frames * world

    opt = dfpn.configure_optimizers()['optimizer']
    mdist.broadcast_parameters(opt)

    def dfpn_step():
        opt.zero_grad()
        loss = dfpn.training_step(batch, 0)
        loss.backward()
        mdist.sync_gradients(opt)
        opt.step()
        return loss

    eager_dfpn_step = dfpn_step
    if not args.no_graph:
        from master_thesis_amd.graph import GraphedTrainStep
        # N == 1: the whole step is one graph; N > 1: forward+backward are the
        # graph, the RCCL all-reduce and the Adam launch follow it
        gstep = GraphedTrainStep(dfpn, opt, batch, include_optimizer=(world == 1))

        def dfpn_step():   # noqa: F811
            verbose = os.environ.get('MT355_BENCH_VERBOSE') == '1'
            t0 = time.perf_counter()
            loss = gstep.step()
            if verbose:
                torch.cuda.synchronize(); t1 = time.perf_counter()
            if world > 1:
                mdist.sync_gradients(opt)
                if verbose:
                    torch.cuda.synchronize(); t2 = time.perf_counter()
                opt.step()
                if verbose:
                    torch.cuda.synchronize()
                    print(f'  replay {t1 - t0:.3f} allreduce {t2 - t1:.3f} adam {time.perf_counter() - t2:.3f}', file=sys.stderr, flush=True)
            return loss

    dt = timed_steps(dfpn_step, args.steps, args.warmup, world)
    ms = dt / args.steps * 1e3
    value = frames_per_step * args.steps / dt
    out = {
        'metric': 'training frames/sec, 256x256x5 DFPN+CHN step', 'value': value, 'unit': 'frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'arithmetic': arithmetic_label(),
        'config': {'workload': f'{cfg_tag(args, "cfg2")}DFPN train step (fwd+loss+bwd+allreduce+Adam), frames_n={args.frames}, '
                               f'{args.size}x{args.size}, batch_size={args.batch}/GPU',
                   'global_batch': args.batch * world, 'frames_n': args.frames, 'parallelism': f'dp{world}'},
        'launch': 'eager' if args.no_graph else 'hipGraph replay of the captured step',
        'step_tflops': TFLOP_PER_STEP['dfpn'] * world / (ms * 1e-3) if (args.batch, args.frames, args.size) == (8, 5, 256) else None,
    }

    if rank == 0:
        agg = kernel_profile(eager_dfpn_step)
    else:
        eager_dfpn_step()
    if rank == 0:
        tot_ms = sum(a[1] for a in agg.values())
        name, (launches, kms, kfl) = max(agg.items(), key=lambda kv: kv[1][1])
        ach = kfl / (kms * 1e-3) / 1e12
        # MFMA products issued per fp32 multiply-add by the kernel's arithmetic
        per = 3 if (',f16' in name or ',2,bf16' in name or name.endswith(',2>') or '_rb_' in name) else (6 if 'split' in name or 'halo' in name else 1)
        peak = PEAK_FP32_MFMA_TFLOPS if per == 1 else PEAK_16BIT_MFMA_TFLOPS / per
        out['roofline'] = {'kernel': name, 'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                           'frac': ach / peak, **pmc_traffic(name),
                           'note': f'achieved = algorithmic conv FLOP / kernel time; peak = dense 16-bit MFMA peak '
                                   f'{PEAK_16BIT_MFMA_TFLOPS:.0f} TFLOP/s / {per} MFMA products per fp32 multiply-add '
                                   f'of this arithmetic (executed MFMA rate = {per} x achieved)' if per > 1 else
                                   'fp32-input MFMA',
                           'launches_per_step': launches, 'avg_launch_ms': kms / launches,
                           'share_of_conv_engine_ms': kms / tot_ms}
        out['kernels'] = {k: {'launches': v[0], 'ms': round(v[1], 3), 'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2)}
                          for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
        out['conv_engine_ms_per_step'] = tot_ms

    if not args.no_chn:
        del opt
        torch.cuda.empty_cache()
        chn = mt.CHN(vgg, None, dfpn, **kw).to(dev)
        opt_c = chn.configure_optimizers()['optimizer']
        mdist.broadcast_parameters(opt_c)

        def chn_step():
            opt_c.zero_grad()
            loss = chn.training_step(batch, 0)
            loss.backward()
            mdist.sync_gradients(opt_c)
            opt_c.step()
            return loss

        eager_chn_step = chn_step
        if not args.no_graph:
            gstep_c = GraphedTrainStep(chn, opt_c, batch, include_optimizer=(world == 1))

            def chn_step():   # noqa: F811
                loss = gstep_c.step()
                if world > 1:
                    mdist.sync_gradients(opt_c)
                    opt_c.step()
                return loss

        dtc = timed_steps(chn_step, args.steps, args.warmup, world)
        out['chn_train_step'] = {
            'workload': f'{cfg_tag(args, "cfg3")}CHN train step (DFPN aligner), frames_n={args.frames}, {args.size}x{args.size}, '
                        f'batch_size={args.batch}/GPU',
            'value': frames_per_step * args.steps / dtc, 'unit': 'frames/s', 'ms_per_step': dtc / args.steps * 1e3}
        if rank == 0:
            aggc = kernel_profile(eager_chn_step)
            out['chn_train_step']['conv_engine_ms_per_step'] = sum(a[1] for a in aggc.values())
            out['chn_train_step']['kernels'] = {
                k: {'launches': v[0], 'ms': round(v[1], 3), 'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2)}
                for k, v in sorted(aggc.items(), key=lambda kv: -kv[1][1])}
        else:
            eager_chn_step()

    if rank == 0:
        out['roofline_corr'] = corr_roofline()
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.frames, args.size, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
