#!/usr/bin/env python
"""Throughput of the DFPN / CHN training step on synthetic clips.

    python bench.py --gpus N --steps K --warmup W [--config 2|5]

* ``--gpus 1`` (default): BASELINE config 2 (DFPN train step, frames_n=5, 256x256, batch 8) is
  ``value``; config 3 (CHN train step, DFPN aligner) is timed in the same run (``chn_train_step``).
* ``--gpus N`` (N > 1): one process per GPU over RCCL.  Started bare, this script launches the N
  ranks itself (``python -m torch.distributed.run --nproc-per-node N ...``, before the parent touches
  a GPU) and exits with their code; started by ``torch.distributed.run`` it is one of the ranks.
  Clips are sharded over ranks (weak scaling: ``--batch`` clips per GPU), one flat gradient
  all-reduce per step (config 4 = ``--gpus 8``).
* ``--gpus 0``: BASELINE config 1, the CPU path: DFPN forward, frames_n=2, 128x128, batch 1 on the
  oracle (oracle/ref_torch.py); no GPU is touched.
* ``--config 5``: frames_n=7, 512x512, batch 4 per GPU; ``value`` is the CHN train step.

A step = forward + losses + backward + all-reduce + Adam on one batch already resident in HBM.
Rank 0 prints ONE JSON line.  At N = 1 it also (a) checks the loss items the GPU path produces on
the benchmarked batch against the CPU oracle (``loss_check``; exit code 3 beyond 1e-3) and (b) times
the oracle on the host cores (``cpu_baseline``).
"""
import argparse
import glob
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md 8d / BASELINE.md section 4: algorithmic conv+GEMM work per step at config 2 / 3
TFLOP_PER_STEP = {'dfpn': 43.31, 'chn': 30.71}
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32-input MFMA peak
PEAK_16BIT_MFMA_TFLOPS = 2500.0      # dense bf16 / f16 MFMA peak (same guide)
PEAK_HBM_GBS = 8000.0
LOSS_TOL = 1e-3                      # north star: within 1e-3 relative of the reference CPU path
CONFIGS = {2: dict(batch=8, frames=5, size=256), 5: dict(batch=4, frames=7, size=512)}


def arithmetic_label():
    from master_thesis_amd import engine as E
    return (f'fp32 tensors; conv MFMA arithmetic {E.PRECISION} '
            f'({"2-term fp16 split of power-of-two scaled operands, 3 f16 MFMA per product, ~7e-8 rel" if E.PRECISION == "f16x3" else E.PRECISION}), '
            f'{E.vgg_precision()} on the never-differentiated VGG passes; weight gradients of the stride-1 3x3 / 5x5 layers: '
            f'{"dY with its leading fp16 term, 2 products (gradient vs the fp64 oracle unchanged: tests/test_gpu_fullsize.py)" if E.WGRAD_D_TERMS == 1 else "both terms of dY, 3 products"}; '
            'fp32 accumulation everywhere')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1, help='0 = BASELINE config 1 on the CPU oracle')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=None, help='clips per GPU (default: the config\'s)')
    ap.add_argument('--frames', type=int, default=None)
    ap.add_argument('--size', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-check', action='store_true', help='skip the GPU-vs-oracle loss check')
    ap.add_argument('--no-chn', action='store_true')
    ap.add_argument('--no-dfpn', action='store_true', help='do not time the DFPN step (profiling runs of the CHN step alone)')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of a captured hipGraph')
    ap.add_argument('--cpu-clips', type=int, default=1, help='clips of the batch the CPU baseline steps on')
    ap.add_argument('--cpu-steps', type=int, default=3, help='timed CPU steps (after one warm-up)')
    args = ap.parse_args()
    for k, v in CONFIGS[args.config].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


def is_named_config(args):
    return CONFIGS[args.config] == dict(batch=args.batch, frames=args.frames, size=args.size)


def timed_steps(step_fn, steps, warmup, world):
    import torch.distributed as dist
    for _ in range(warmup):
        step_fn()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt


def kernel_profile(step_fn):
    """Two extra eager steps, the second with HIP events around every conv-engine launch (on the
    stream they run on): per-kernel launches, total ms and algorithmic FLOP."""
    from master_thesis_amd import engine as E
    step_fn()                     # untimed: the eager path's allocations and clocks settle after the graph replays
    torch.cuda.synchronize()
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


def kernels_table(agg):
    return {k: {'launches': v[0], 'ms': round(v[1], 3), 'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2)}
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}


def corr_roofline(b_pairs=2048):
    """Stand-alone correlation (K5) micro-benchmark on >= 2048 pairs in flight
    (SURVEY.md 8d): algorithmic bytes = 1.31 MB per pair (0.92 MB with the target slab shared)."""
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


def warp_roofline(pairs=1024, size=256):
    """Stand-alone warp (K9, FlowsUtils.align_set, utils.py:77-104) micro-benchmark: bilinear RGB + nearest
    visibility in one launch on ``pairs`` (clip, reference) pairs at ``size``^2 -- 40*h*w bytes per pair
    (read (3+1+2)*h*w*4, write (3+1)*h*w*4), >= 2 GB per launch so that the Infinity Cache
    cannot hold it.  Flows are small random affine grids plus sub-pixel noise (the training distribution)."""
    from master_thesis_amd import ops
    import torch.nn.functional as F
    b, r = pairs // 4, 4
    g = torch.Generator(device='cpu').manual_seed(7)
    x = torch.rand((b, 3, r, size, size), generator=g).cuda()
    v = (torch.rand((b, 1, r, size, size), generator=g) > 0.1).float().cuda()
    p = torch.rand((b * r, 4), generator=g) * 2 - 1
    ang, sc = p[:, 0] * 0.2, 1 + p[:, 1] * 0.1
    theta = torch.zeros((b * r, 2, 3))
    theta[:, 0, 0], theta[:, 0, 1] = sc * torch.cos(ang), -sc * torch.sin(ang)
    theta[:, 1, 0], theta[:, 1, 1] = sc * torch.sin(ang), sc * torch.cos(ang)
    theta[:, :, 2] = p[:, 2:4] * (50.0 / (size / 2))
    flow = F.affine_grid(theta, [b * r, 1, size, size], align_corners=True).reshape(b, r, size, size, 2).cuda()
    flow = (flow + (torch.rand(flow.shape, device='cuda') - 0.5) * (2.0 / size)).contiguous()
    with torch.no_grad():
        for _ in range(5):
            ops.align_set(x, v, flow)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for _ in range(n):
            ops.align_set(x, v, flow)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    bytes_ = 40.0 * size * size * b * r
    gbs = bytes_ / (ms * 1e-3) / 1e9
    return {'kernel': 'warp_fwd_kernel (mt_warp_fwd)', 'pairs': b * r, 'size': size, 'ms': ms, 'bound': 'hbm',
            'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS,
            'bytes_per_launch': bytes_}


def csrc_sha16():
    """Hash of the kernel sources (names + contents, sorted): tools/collect_profiles.sh stamps the PMC summaries with it."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'master_thesis_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h', '.cpp')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(label):
    """HBM bytes per launch of the kernel ``label`` from the newest committed rocprofv3 PMC passes
    (profiles/rNN_pmc_hbm_traffic.txt: FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 correction
    applied there); None when the profile holds no row for it.  The counters cannot be collected
    inside this process, so this is the profile of the same workload (tools/collect_profiles.sh)."""
    none = {'traffic': None}
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_hbm_traffic.txt')))
    if not files:
        return none
    rows = open(files[-1]).read().splitlines()
    want = label.replace('bf16', '0').replace('f16', '1').replace(' ', '')
    for row in rows:
        if row.startswith('#') or '>' not in row:
            continue
        name, rest = row.rsplit('>', 1)
        if (name + '>').replace(' ', '').startswith(want.rstrip('>')):   # (the label may omit trailing template arguments)
            f = rest.strip(',').split(',')
            out = {'traffic': float(f[3]) * 1e6, 'traffic_unit': 'bytes/launch (PMC, (2*FETCH_SIZE+WRITE_SIZE)*1024)',
                   'traffic_source': os.path.relpath(files[-1], ROOT), 'traffic_launches': int(f[0])}
            # a counter file that describes an earlier build of the kernels goes stale silently: the collection script
            # stamps it with a hash of the kernel sources (file times do not survive a checkout), compare
            stamp = [r.split()[-1] for r in rows if r.startswith('# csrc_sha16:')]
            if not stamp or stamp[0] != csrc_sha16():
                out['traffic_stale'] = True
                print(f'bench.py: WARNING {out["traffic_source"]} was collected on other kernel sources (stamp '
                      f'{stamp[0] if stamp else "none"}, now {csrc_sha16()}): roofline.traffic describes an earlier build '
                      '(re-run tools/collect_profiles.sh)', file=sys.stderr)
            return out
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


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from oracle import ref_torch as O
    return O


def _cpu_batch(batch, clips=None):
    (x, m), y, info = batch
    cut = (lambda t: t[:clips]) if clips else (lambda t: t)
    info = tuple(cut(i).cpu() if torch.is_tensor(i) else i for i in info)
    return (cut(x).cpu(), cut(m).cpu()), cut(y).cpu(), info


def cpu_baseline(dfpn, chn, batch, clips, steps):
    """The CPU oracle (port of the reference path, oracle/ref_torch.py) timed on this box's host
    cores: ``clips`` clips of the benchmarked batch, the benchmarked weights, one warm-up step and
    ``steps`` timed training steps (forward + loss + backward; Adam is noise beside them), median."""
    O = _oracle()
    cores = host_cores()
    torch.set_num_threads(cores)
    cb = _cpu_batch(batch, clips)
    frames = cb[0][0].shape[2]
    size = cb[0][0].shape[-1]

    def run(step_fn, sd, trainable):
        for n in trainable:
            sd[n] = sd[n].clone().requires_grad_(True)
        times = []
        for i in range(steps + 1):
            t0 = time.perf_counter()
            loss, _ = step_fn(sd, cb)
            torch.autograd.grad(loss, [sd[n] for n in trainable])
            times.append(time.perf_counter() - t0)
        return statistics.median(times[1:]), times

    sd = {k: v.detach().cpu() for k, v in dfpn.state_dict().items()}
    med, all_t = run(O.dfpn_training_step, sd, [k for k in sd if 'model_vgg' not in k and k not in ('mean', 'std')])
    out = {'value': clips * frames / med, 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
           'sample': f'{clips} clip(s) x {frames} frames x {size}x{size} of the benchmarked batch, DFPN training step '
                     f'(fwd+loss+bwd) on the oracle (oracle/ref_torch.py, torch fp32 CPU ops, {cores} threads): '
                     f'1 warm-up + {steps} timed steps, median {med:.2f} s (all: {[round(t, 2) for t in all_t]})'}
    if chn is not None:
        sd = {k: v.detach().cpu() for k, v in chn.state_dict().items()}
        medc, all_c = run(O.chn_training_step, sd, [k for k in sd if k.startswith('nn.')])
        out['chn_train_step'] = {'value': clips * frames / medc, 'unit': 'frames/s',
                                 'sample': f'CHN training step (DFPN aligner), same clips: 1 warm-up + {steps} timed, '
                                           f'median {medc:.2f} s (all: {[round(t, 2) for t in all_c]})'}
    return out


def loss_check(kind, model, batch):
    """GPU loss items of ONE eager forward on the whole benchmarked batch against the CPU oracle on
    the same batch and weights (forward only, no_grad)."""
    O = _oracle()
    torch.set_num_threads(host_cores())
    with torch.no_grad():
        loss, items = model._step(batch, 4) if kind == 'dfpn' else model._step(batch)
        got = [float(loss)] + [float(i) for i in items]
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        t0 = time.perf_counter()
        step = O.dfpn_training_step if kind == 'dfpn' else O.chn_training_step
        ref_loss, ref_items = step(sd, _cpu_batch(batch))
        dt = time.perf_counter() - t0
    ref = [float(ref_loss)] + [float(i) for i in ref_items]
    names = ['total'] + list(type(model).LOSSES_NAMES)
    scale = max(abs(r) for r in ref)
    # an item is judged relative to itself, with a floor of 1e-3 of the largest item (a term that is
    # ~0 on this batch cannot be held to a relative bound)
    errs = {n: abs(g - r) / max(abs(r), 1e-3 * scale) for n, g, r in zip(names, got, ref)}
    worst = max(errs.values())
    return {'ok': bool(worst <= LOSS_TOL), 'max_rel_err': worst, 'tol': LOSS_TOL,
            'rel_err': {k: float(f'{v:.3e}') for k, v in errs.items()},
            'gpu': dict(zip(names, got)), 'oracle': dict(zip(names, ref)), 'oracle_seconds': round(dt, 1)}


# ------------------------------------------------------------------ --gpus 0 ---
def bench_cpu_config1(args):
    """BASELINE config 1: DFPN forward, frames_n=2, 128x128, batch 1, on the CPU oracle."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import seeded as S
    O = _oracle()
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = S.add_buffers(S.seeded_state_dict(S.dfpn_shapes()))
    (x, m), y, info = S.synthetic_batch(1, 2, 128, 128, seed=5)
    t, r = O.get_indexes(2)
    times = []
    with torch.no_grad():
        for _ in range(args.warmup + args.steps):
            t0 = time.perf_counter()
            O.dfpn_forward(sd, x[:, :, t], m[:, :, t], x[:, :, r], m[:, :, r])
            times.append(time.perf_counter() - t0)
    times = times[args.warmup:]
    sec = sum(times) / len(times)
    print(json.dumps({
        'metric': 'DFPN forward frames/sec, CPU reference path', 'value': 2 / sec, 'unit': 'frames/s', 'n_gpus': 0,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': sec * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'cfg1: DFPN forward, frames_n=2, 128x128, batch_size=1, CPU oracle (oracle/ref_torch.py), no_grad',
                   'global_batch': 1, 'frames_n': 2, 'parallelism': 'cpu'},
        'cores': cores, 'best_ms': min(times) * 1e3,
        'note': 'BASELINE.md section 2: the reference itself ran this config in 1.27 s (best of 3) on 8 threads of the '
                'build container; this line is the oracle (parity-pinned restatement) on this box'}))


# --------------------------------------------------------- --gpus N launcher ---
def visible_gpu_count():
    """GPUs this process could open, counted WITHOUT any HIP / torch.cuda call (the launcher parent must not
    initialise a GPU before it starts the ranks): KFD topology nodes that have SIMDs and whose DRM render node
    this user may open, cut by the *_VISIBLE_DEVICES lists.  Falls back to asking a child process when the KFD
    sysfs tree is absent (the child touches HIP and exits; this process still does not)."""
    nodes = glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties')
    if not nodes:
        try:
            out = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'],
                                 capture_output=True, text=True, timeout=300).stdout.split()
            return int(out[-1]) if out else 0
        except (OSError, ValueError, subprocess.TimeoutExpired):
            return 0
    n = 0
    for path in nodes:
        try:
            props = dict(line.split()[:2] for line in open(path).read().splitlines() if len(line.split()) >= 2)
        except OSError:
            continue            # a node of another container's GPU: not readable = not ours
        if int(props.get('simd_count', '0')) == 0:
            continue            # CPU node
        minor = props.get('drm_render_minor')
        dev = f'/dev/dri/renderD{minor}'
        if minor is None or not os.access(dev, os.R_OK | os.W_OK):
            continue            # present on the host, not handed to this container
        n += 1
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def spawn_ranks(args):
    """Bare ``python bench.py --gpus N``: become the launcher of N ranks.  Nothing here initialises
    a GPU (devices are counted from sysfs: visible_gpu_count); the ranks are fresh child processes."""
    backend = os.environ.get('MT355_BENCH_BACKEND', 'nccl')
    have = visible_gpu_count()
    if backend == 'nccl' and have < args.gpus:
        print(f'bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible', file=sys.stderr)
        sys.exit(2)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    sys.exit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------ one rank ---
def claim_stdout():
    """The contract is ONE JSON line on stdout, but RCCL and gloo print banners to file descriptor 1 when a
    communicator comes up.  From here on fd 1 is stderr for everybody; the returned file is the real stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    return real


def run_rank(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    real_stdout = claim_stdout()
    if args.gpus != world:
        print(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)', file=sys.stderr)
        sys.exit(2)
    # MT355_BENCH_BACKEND=gloo rehearses the N > 1 code path with several ranks on ONE card
    # (ranks share device local % device_count; gloo stages CUDA tensors through the host)
    backend = os.environ.get('MT355_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    if backend == 'nccl' and ndev < world:
        print(f'bench.py: {world} ranks but only {ndev} GPU(s) are visible', file=sys.stderr)
        sys.exit(2)
    local = local % max(ndev, 1)
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
    from master_thesis_amd.graph import GraphedTrainStep
    from master_thesis_amd.synthetic import make_batch
    mt._lib.load()
    if world > 1:
        mdist.init_comm(dev)          # RCCL communicator behind the C ABI (falls back, loudly, to torch.distributed)

    kw = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
    torch.manual_seed(0)
    vgg = mt.VGGFeatures(device=dev)
    dfpn = mt.DFPN(vgg, **kw).to(dev)
    batch = make_batch(args.batch, args.frames, args.size, args.size, dev, seed=1234 + rank)
    frames_per_step = args.batch * args.frames * world
    named = is_named_config(args)

    def make_steps(model):
        """(timed step, eager step, optimizer) of one model."""
        opt = model.configure_optimizers()['optimizer']
        mdist.broadcast_parameters(opt)

        def eager():
            opt.zero_grad()
            mdist.arm_overlap(opt)            # N > 1: finished windows of the flat gradient go on the wire during the backward
            loss = model.training_step(batch, 0)
            loss.backward()
            mdist.sync_gradients(opt)
            opt.step()

        if args.no_graph:
            return eager, eager, opt
        # N == 1: the whole step is one graph; N > 1: forward + backward are the graph, the
        # all-reduce (side stream, event-ordered) and the Adam launch follow it
        g = GraphedTrainStep(model, opt, batch, include_optimizer=(world == 1))

        def graphed(marks=None):
            g.step()
            if world > 1:
                if marks is not None:
                    marks[0].record()
                mdist.sync_gradients(opt)
                if marks is not None:
                    marks[1].record()
                opt.step()
                if marks is not None:
                    marks[2].record()
        return graphed, eager, opt

    def measure(model, kind):
        step, eager, opt = make_steps(model)
        dt = timed_steps(step, args.steps, args.warmup, world)
        res = {'value': frames_per_step * args.steps / dt, 'unit': 'frames/s', 'ms_per_step': dt / args.steps * 1e3}
        if named and args.config == 2:
            res['step_tflops'] = TFLOP_PER_STEP[kind] * world / (res['ms_per_step'] * 1e-3)
        if not args.no_graph:
            # what a plain PyTorch-Lightning loop runs (__main__.py:58 trainer.fit): the same step launched eagerly
            # from Python, no hipGraph -- same protocol (warm-up, barrier + synchronize on both sides, max over ranks)
            dt_e = timed_steps(eager, args.steps, args.warmup, world)
            res['eager_ms_per_step'] = dt_e / args.steps * 1e3
        if world > 1 and not args.no_graph:
            # where an N > 1 step spends its time besides forward + backward: HIP events on the launch stream around the
            # gradient exchange (it runs on the communicator's side stream; the launch stream waits for its last bucket)
            # and around the Adam launch, three extra steps, this rank
            marks = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(3)]
            for mk in marks:
                step(mk)
            torch.cuda.synchronize()
            res['allreduce_ms_per_step'] = statistics.median(mk[0].elapsed_time(mk[1]) for mk in marks)
            res['adam_ms_per_step'] = statistics.median(mk[1].elapsed_time(mk[2]) for mk in marks)
        if rank == 0:
            agg = kernel_profile(eager)
            res['conv_engine_ms_per_step'] = sum(a[1] for a in agg.values())
            res['kernels'] = kernels_table(agg)
            res['_agg'] = agg
        else:
            eager()
            eager()
        return res, opt

    results = {}
    opt_d = None
    if not (args.no_dfpn and not args.no_chn):
        results['dfpn'], opt_d = measure(dfpn, 'dfpn')
    chn = None
    if not args.no_chn:
        del opt_d
        torch.cuda.empty_cache()
        chn = mt.CHN(vgg, None, dfpn, **kw).to(dev)
        results['chn'], opt_c = measure(chn, 'chn')

    head_kind = 'chn' if ((args.config == 5 or 'dfpn' not in results) and 'chn' in results) else 'dfpn'
    head = results[head_kind]
    cfg_name = {('dfpn', 2): 'cfg2', ('chn', 2): 'cfg3', ('chn', 5): 'cfg5', ('dfpn', 5): 'cfg5-shape'}
    if world > 1:
        cfg_name[('chn', 2)] = 'cfg4' if world == 8 else 'cfg4-shape'

    def workload(kind):
        what = 'DFPN train step' if kind == 'dfpn' else 'CHN train step (DFPN aligner)'
        tag = (cfg_name[(kind, args.config)] + ': ') if named else ''
        return (f'{tag}{what} (fwd+loss+bwd+allreduce+Adam), frames_n={args.frames}, {args.size}x{args.size}, '
                f'batch_size={args.batch}/GPU')

    out = {
        'metric': 'training frames/sec, 256x256x5 DFPN+CHN step' if args.config == 2 else
                  f'training frames/sec, {args.size}x{args.size}x{args.frames} DFPN+CHN step',
        'value': head['value'], 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic', 'arithmetic': arithmetic_label(),
        'config': {'workload': workload(head_kind), 'global_batch': args.batch * world, 'frames_n': args.frames,
                   'parallelism': f'dp{world}'},
        'launch': 'eager' if args.no_graph else 'hipGraph replay of the captured step',
        'collective': {'transport': mdist.transport(), 'world_size_seen': world if world == 1 else
                       (mdist._comm.world if mdist._comm is not None else torch.distributed.get_world_size()),
                       'bytes_per_step': 4 * sum(p.numel() for p in (chn if head_kind == 'chn' else dfpn).trainable_parameters())},
    }
    if 'allreduce_ms_per_step' in head:
        out['collective']['ms_per_step'] = head['allreduce_ms_per_step']
        out['collective']['adam_ms_per_step'] = head['adam_ms_per_step']
        out['collective']['note'] = ('rank 0, median of 3 steps: HIP events on the launch stream around sync_gradients '
                                     '(buckets on the side stream, event-ordered) and around the Adam launch that follows the graph')
    if 'step_tflops' in head:
        out['step_tflops'] = head['step_tflops']
    if 'eager_ms_per_step' in head:
        out['eager_ms_per_step'] = head['eager_ms_per_step']

    if rank == 0:
        agg = head['_agg']
        tot_ms = sum(a[1] for a in agg.values())
        name, (launches, kms, kfl) = max(agg.items(), key=lambda kv: kv[1][1])
        ach = kfl / (kms * 1e-3) / 1e12
        # MFMA products issued per fp32 multiply-add by the kernel's arithmetic, and the share of the direct form's products a
        # 1-D Winograd kernel executes (F(2,3) 2/3, F(2,5) 3/5, F(4,3) 1/2): both reported by the library with the plan
        # the launch was made from (mt_conv_plan.products / wino_num / wino_den) -- nothing is parsed out of the name.
        # The ceiling in ALGORITHMIC FLOP/s is the MFMA peak / products / that share, and frac against THAT is the fraction
        # of the matrix pipe's peak the executed MFMAs use (frac must not be able to pass 1)
        from master_thesis_amd import engine as E
        per, executed_share = E.KERNEL_INFO.get(name, (1, 1.0))        # (no entry: an fp32 VALU kernel)
        saving = 1.0 / executed_share
        peak = PEAK_FP32_MFMA_TFLOPS if per == 1 else PEAK_16BIT_MFMA_TFLOPS / per * saving
        note = ('achieved = ALGORITHMIC conv FLOP (2*N*H*W*Cin*Cout*k*k per launch) / kernel time (HIP events around every '
                f'launch of one eager step); peak = dense 16-bit MFMA peak {PEAK_16BIT_MFMA_TFLOPS:.0f} TFLOP/s / {per} MFMA '
                'products per fp32 multiply-add of this arithmetic' +
                (f' x {saving:.3g} (1-D Winograd executes 1/{saving:.3g} of the direct form\'s products): frac = executed MFMA '
                 'FLOP/s over the dense peak' if saving != 1.0 else '')) if per > 1 else 'fp32-input MFMA'
        out['roofline'] = {'kernel': name, 'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                           'frac': ach / peak, **pmc_traffic(name), 'note': note,
                           'executed_mfma_tflops': ach * per / saving,
                           'frac_of_direct_form_ceiling': ach / (peak / saving),
                           'launches_per_step': launches, 'avg_launch_ms': kms / launches,
                           'share_of_conv_engine_ms': kms / tot_ms}
        out['kernels'] = head['kernels']
        out['conv_engine_ms_per_step'] = head['conv_engine_ms_per_step']
        for kind, res in results.items():
            if kind != head_kind:
                res = {k: v for k, v in res.items() if k != '_agg'}
                out['chn_train_step' if kind == 'chn' else 'dfpn_train_step'] = {'workload': workload(kind), **res}
        out['roofline_corr'] = corr_roofline()
        out['roofline_warp'] = warp_roofline()

    failed = False
    if rank == 0 and world == 1:
        if not args.no_check:
            out['loss_check'] = {'dfpn': loss_check('dfpn', dfpn, batch)}
            if chn is not None:
                out['loss_check']['chn'] = loss_check('chn', chn, batch)
            failed = not all(v['ok'] for v in out['loss_check'].values())
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(dfpn, chn, batch, min(args.cpu_clips, args.batch), args.cpu_steps)
    if rank == 0:
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        mdist.shutdown()
        dist.destroy_process_group()
    if failed:
        print('bench.py: GPU loss items differ from the CPU oracle by more than 1e-3 (see loss_check)', file=sys.stderr)
        sys.exit(3)


def main():
    args = parse()
    if args.gpus == 0:
        return bench_cpu_config1(args)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return spawn_ranks(args)
    run_rank(args)


if __name__ == '__main__':
    main()
