"""CPU-only checks: the C-ABI library loads and exports every symbol that
include/mt355.h declares (no compute call is made without a GPU), the ctypes
binding covers the whole header, and the host-side mirror keeps the reference's
names, state-dict keys and layer geometry."""
import ctypes
import os
import re

import pytest
import torch

import seeded as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'mt355.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mt_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from master_thesis_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f'{n} declared in mt355.h but not exported'
    assert lib.mt_version() == 2


def test_library_exports_nothing_but_the_header():
    """Built with -fvisibility=hidden and a linker version script: the dynamic symbol table of libmt355.so holds the
    entry points mt355.h declares and nothing else (no internal C++ helper, no hipcc artefact) -- VERDICT round 3, #12."""
    import subprocess
    import __graft_entry__ as g
    g.build()
    from master_thesis_amd import _lib
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == header_symbols(), sorted(set(exported) ^ set(header_symbols()))


def test_binding_covers_the_header():
    from master_thesis_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_struct_layouts_match_the_header():
    """sizeof() of the ctypes mirrors vs. the C structs (compiled with gcc)."""
    import subprocess
    import tempfile
    from master_thesis_amd import _lib
    src = '#include <stdio.h>\n#include "mt355.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mt_view), ' \
          'sizeof(mt_conv), sizeof(mt_wgrad), sizeof(mt_warp_layout), sizeof(mt_pack_src), sizeof(mt_amax_seg), ' \
          'sizeof(mt_pack_job), sizeof(mt_reduce_sub), sizeof(mt_conv_plan), sizeof(mt_wgrad_plan));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 's.c')
        open(c, 'w').write(src)
        exe = os.path.join(d, 's')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    mine = [ctypes.sizeof(t) for t in (_lib.View, _lib.Conv, _lib.Wgrad, _lib.WarpLayout, _lib.PackSrc, _lib.AmaxSeg,
                                       _lib.PackJob, _lib.ReduceSub, _lib.ConvPlan, _lib.WgradPlan)]
    assert sizes == mine, (sizes, mine)


def test_missing_library_fails_loudly(monkeypatch):
    from master_thesis_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libmt355.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_package_surface_and_state_dict_keys():
    import master_thesis_amd as mt
    for name in ('DFPN', 'CHN', 'CPN', 'VGGFeatures', 'FlowsUtils', 'LossesUtils', 'TransformsUtils'):
        assert hasattr(mt, name)
    kw = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
    vgg = mt.VGGFeatures()
    dfpn = mt.DFPN(vgg, **kw)
    want = S.add_buffers({k: None for k in S.dfpn_shapes()})
    assert set(dfpn.state_dict()) == set(want)
    for k, shape in S.dfpn_shapes().items():
        assert tuple(dfpn.state_dict()[k].shape) == tuple(shape), k
    chn = mt.CHN(vgg, None, dfpn, **kw)
    want = S.add_buffers({k: None for k in S.chn_shapes()}, ('', 'model_aligner.'))
    assert set(chn.state_dict()) == set(want)
    assert mt.DFPN.LOSSES_NAMES == ['corr_loss', 'flow_16', 'flow_64', 'flow_256', 'alignment_recons_64',
                                    'alignment_recons_256']
    assert mt.CHN.LOSSES_NAMES == ['loss_nh', 'loss_vh', 'loss_nvh', 'loss_perceptual', 'loss_grad']
    assert mt.DFPN.get_indexes(5) == (2, [0, 1, 3, 4]) and mt.CHN.get_indexes(2) == (1, [0])
    assert sum(p.numel() for p in dfpn.trainable_parameters()) == 13922927
    assert sum(p.numel() for p in chn.trainable_parameters()) == 14690883


def test_layer_geometry():
    from master_thesis_amd import engine as E
    assert E.LayerSpec('c', 128, 128, 5, 2).out_hw(256, 256) == (128, 128)
    assert E.LayerSpec('c', 128, 128, 3, 2).out_hw(33, 21) == (17, 11)
    assert E.LayerSpec('t', 128, 128, 3, 2).out_hw(8, 12) == (16, 24)
    assert E.LayerSpec('t', 128, 128, 5, 2).out_hw(8, 8) == (16, 16)
    assert E.LayerSpec('c', 64, 64, 3, 1, ups=1).out_hw(16, 16) == (32, 32)
    assert E.cpad(1) == 8 and E.cpad(10) == 16 and E.cpad(128) == 128
    from master_thesis_amd.model_dfpn import FlowEstimator, _specs_from
    specs, params = _specs_from(FlowEstimator().nn)
    assert len(specs) == 21 and len(params) == 42
    assert [s.kind for s in specs].count('t') == 3 and specs[-1].act == 0 and specs[0].act == 1
    h = 64
    for s in specs:
        h, _ = s.out_hw(h, h)
    assert h == 64


def test_synthetic_batch_layout():
    from master_thesis_amd.synthetic import make_batch
    (x, m), y, info = make_batch(2, 3, 32, 32, 'cpu', seed=5)
    assert x.shape == (2, 3, 3, 32, 32) and m.shape == (2, 1, 3, 32, 32) and info[4].shape == (2, 3, 32, 32, 2)
    assert set(m.unique().tolist()) <= {0.0, 1.0} and 0.02 < float(m.mean()) < 0.5
    fill = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1)
    assert torch.equal(x, (1 - m) * y + m * fill)
    assert info[2].tolist() == [True, False] and float(info[4][1].abs().max()) == 0.0


def test_bench_launcher_refuses_missing_gpus():
    """`bench.py --gpus N` launches its own ranks; with fewer than N devices visible it must stop with
    exit code 2 before anything else happens (here: no GPU at all), never report a 1-GPU number."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a box with fewer than 2 GPUs')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and 'GPU(s) are visible' in r.stderr, (r.returncode, r.stderr[-300:])
    # a rank whose launcher disagrees with --gpus stops too
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env2, timeout=300)
    assert r.returncode == 2 and 'launcher started 1 rank' in r.stderr, (r.returncode, r.stderr[-300:])
