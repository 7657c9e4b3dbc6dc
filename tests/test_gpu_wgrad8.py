"""GPU tests of the eight-wave weight-gradient program (csrc/conv_wgrad.hip: conv_wgrad_octo_body - 512-thread workgroups that own a
64 x 64 block as 2 x 2 pairs x 2 pixel slices, or a 32 x 32 block x 8 pixel slices; the 3x3 stride-1 bf16 layers of the BasicBlocks,
reference HRnet.py:35-58) through the C ABI's batched launch, against torch's conv2d weight gradient in float64 on the same bf16
operands."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _problem(nv, N, H, W, cin, cout, seed, x3=False, k=3, stride=1):
    """H x W: the OUTPUT map (dY); the input is stride times as large"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    dt = torch.float32 if x3 else torch.bfloat16
    x = (torch.randn(N, H * stride, W * stride, cin, generator=g)).to(dt).to(DEV).contiguous()
    dy = (torch.randn(N, H, W, cout, generator=g) / 8).to(dt).to(DEV).contiguous()
    d = nv.WgradDesc()
    d.x, d.dy, d.dtype = x.data_ptr(), dy.data_ptr(), nv.HRP_F32X3 if x3 else nv.HRP_BF16
    d.N, d.H, d.W, d.Cin, d.x_pitch = N, H * stride, W * stride, cin, cin
    d.Ho, d.Wo, d.Cout, d.dy_pitch = H, W, cout, cout
    d.in_stride, d.ntaps = stride, k * k
    for i, (a, b) in enumerate([(ky - k // 2, kx - k // 2) for ky in range(k) for kx in range(k)]):
        d.dy_t[i], d.dx_t[i] = a, b
    d.dw_cin = cin
    return d, x, dy


def _reference(x, dy, cin, cout, k=3, stride=1):
    xr = x.double().permute(0, 3, 1, 2)
    gr = dy.double().permute(0, 3, 1, 2)
    return torch.nn.grad.conv2d_weight(xr, (cout, cin, k, k), gr, stride=stride, padding=k // 2).reshape(cout, cin, k * k)


SHAPES = [
    # N, H, W, cin, cout
    (6, 32, 32, 64, 64),       # one 64 x 64 block, 128-pixel tiles
    (6, 16, 16, 128, 128),     # 2 x 2 blocks of 64 x 64
    (5, 8, 8, 256, 256),       # two images per tile, odd image count (idle tile slots)
    (3, 20, 24, 64, 128),      # border tiles that leave the image in both directions, cin != cout
    (3, 64, 64, 32, 32),       # 32 x 32 block x 8 pixel slices, 512-pixel tiles
    (2, 40, 24, 32, 32),       # ... with border tiles
    (2, 16, 16, 32, 64),       # not eligible: the 32 x 32 program in the same batch (its own launch)
    (3, 12, 12, 192, 64),      # three cin blocks of 64, one cout block
]


@pytest.mark.parametrize("x3", [False, True], ids=["bf16", "fp32x3"])
@pytest.mark.parametrize("phase", [0, 1])
def test_eight_wave_program_in_a_mixed_batch(phase, x3):
    """bf16: exact products, fp32 accumulation.  fp32x3 (fp32 tensors, three bf16 products per term; its eight-wave program takes
    the multiples of 64 channels, 128- or 64-pixel tiles): 2^-16 per product."""
    from hrpe_amd import _native as nv
    L = nv.lib()
    probs = [_problem(nv, *s, seed=70 + i, x3=x3) for i, s in enumerate(SHAPES)]
    n = len(probs)
    arr = (nv.WgradDesc * n)(*[p[0] for p in probs])
    dws = []
    for d in arr:
        d.phase, d.accumulate = phase, 0
        dws.append(torch.full((d.Cout * d.dw_cin * 9,), 3.0, device=DEV))
        d.dw = dws[-1].data_ptr()
    info = nv.BatchInfo()
    nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD, arr, n, None, C.byref(info)), "query")
    wss = []
    for i, d in enumerate(arr):
        ws = torch.zeros(int(info.ws_bytes[i]) // 4 + 4, device=DEV)
        d.workspace, d.workspace_bytes = ws.data_ptr(), int(info.ws_bytes[i])
        wss.append(ws)
    host = (C.c_char * int(L.hrp_batch_table_bytes(nv.BATCH_WGRAD, n)))()
    nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD, arr, n, host, C.byref(info)), "prepare")
    assert info.grid3 > 0 and info.grid > 0, "the batch holds problems of both programs"
    tab = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(DEV)
    nv.check(L.hrp_batch_launch(tab.data_ptr(), C.byref(info), None), "launch")
    if phase == 1:
        folds = (nv.WgradFoldDesc * n)()
        nv.check(L.hrp_batch_wgrad_fold_descs(host, C.byref(info), folds), "fold descs")
        torch.cuda.synchronize()
        assert all(float(dw.min()) == 3.0 == float(dw.max()) for dw in dws), "phase 1 must not touch dw"
        finfo = nv.BatchInfo()
        fhost = (C.c_char * int(L.hrp_batch_table_bytes(nv.BATCH_WGRAD_FOLD, n)))()
        nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD_FOLD, folds, n, fhost, C.byref(finfo)), "fold prepare")
        ftab = torch.frombuffer(bytearray(bytes(fhost)), dtype=torch.uint8).to(DEV)
        nv.check(L.hrp_batch_launch(ftab.data_ptr(), C.byref(finfo), None), "fold launch")
    torch.cuda.synchronize()
    for (d, x, dy), dw, s in zip(probs, dws, SHAPES):
        ref = _reference(x, dy, s[3], s[4]).to(DEV)
        got = dw.view(s[4], s[3], 9).double()
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, f"{s}: weight gradient off by {err:.2e} of its range"      # fp32 accumulation of exact bf16 products


def test_eight_wave_program_alone_accumulates():
    """A batch without a problem of the 32 x 32 program (grid == 0) launches, and accumulate = 1 adds to dw."""
    from hrpe_amd import _native as nv
    L = nv.lib()
    probs = [_problem(nv, 4, 16, 16, 128, 128, 91), _problem(nv, 4, 32, 32, 64, 64, 92)]
    arr = (nv.WgradDesc * 2)(*[p[0] for p in probs])
    dws = []
    for d in arr:
        d.phase, d.accumulate = 0, 1
        dws.append(torch.full((d.Cout * d.dw_cin * 9,), 0.5, device=DEV))
        d.dw = dws[-1].data_ptr()
    info = nv.BatchInfo()
    nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD, arr, 2, None, C.byref(info)), "query")
    wss = []
    for i, d in enumerate(arr):
        ws = torch.zeros(int(info.ws_bytes[i]) // 4 + 4, device=DEV)
        d.workspace, d.workspace_bytes = ws.data_ptr(), int(info.ws_bytes[i])
        wss.append(ws)
    host = (C.c_char * int(L.hrp_batch_table_bytes(nv.BATCH_WGRAD, 2)))()
    nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD, arr, 2, host, C.byref(info)), "prepare")
    assert info.grid == 0 and info.grid3 > 0
    tab = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(DEV)
    nv.check(L.hrp_batch_launch(tab.data_ptr(), C.byref(info), None), "launch")
    torch.cuda.synchronize()
    for (d, x, dy), dw in zip(probs, dws):
        ref = _reference(x, dy, d.Cin, d.Cout).to(DEV) + 0.5
        err = float((dw.view(d.Cout, d.Cin, 9).double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, err


def test_eight_wave_program_fp32x3_pointwise_layers():
    """The 1x1 layers of an fp32x3 plan (the Bottleneck convolutions of layer1, 256 <-> 64 channels at 64 x 64: 1 KB of x per
    pixel) run the eight-wave program without a halo."""
    from hrpe_amd import _native as nv
    L = nv.lib()
    shapes = [(4, 32, 32, 256, 64), (3, 16, 16, 64, 256), (2, 20, 12, 128, 128), (2, 16, 16, 32, 64)]
    probs = [_problem(nv, *s, seed=120 + i, x3=True, k=1) for i, s in enumerate(shapes)]
    n = len(probs)
    arr = (nv.WgradDesc * n)(*[p[0] for p in probs])
    dws = []
    for d in arr:
        d.phase, d.accumulate = 0, 0
        dws.append(torch.full((d.Cout * d.dw_cin,), 3.0, device=DEV))
        d.dw = dws[-1].data_ptr()
    info = nv.BatchInfo()
    nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD, arr, n, None, C.byref(info)), "query")
    wss = []
    for i, d in enumerate(arr):
        ws = torch.zeros(int(info.ws_bytes[i]) // 4 + 4, device=DEV)
        d.workspace, d.workspace_bytes = ws.data_ptr(), int(info.ws_bytes[i])
        wss.append(ws)
    host = (C.c_char * int(L.hrp_batch_table_bytes(nv.BATCH_WGRAD, n)))()
    nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD, arr, n, host, C.byref(info)), "prepare")
    assert info.grid3 > 0 and info.grid > 0
    tab = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(DEV)
    nv.check(L.hrp_batch_launch(tab.data_ptr(), C.byref(info), None), "launch")
    torch.cuda.synchronize()
    for (d, x, dy), dw, s in zip(probs, dws, shapes):
        ref = _reference(x, dy, s[3], s[4], k=1).to(DEV)
        err = float((dw.view(s[4], s[3], 1).double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, f"{s}: {err:.2e}"


@pytest.mark.parametrize("x3", [False, True], ids=["bf16", "fp32x3"])
@pytest.mark.parametrize("phase", [0, 1])
def test_eight_wave_program_stride_2_layers(phase, x3):
    """The stride-2 3x3 layers (fuse-layer down paths, transitions, the cls head's downsamp_modules: reference HRnet.py:195-235,
    383-405) with multiples of 64 channels run the 64 x 64 arrangement on 64-pixel tiles (the halo of a 128-pixel tile does not fit);
    32 -> 64 and odd shapes keep the four-wave program.  fp32x3: 32-pixel tiles (RAW tile + planes)."""
    from hrpe_amd import _native as nv
    L = nv.lib()
    shapes = [(3, 16, 16, 128, 256), (4, 8, 8, 256, 512), (2, 32, 32, 64, 64), (3, 12, 20, 64, 128), (2, 16, 16, 32, 64)]
    probs = [_problem(nv, *s, seed=140 + i, stride=2, x3=x3) for i, s in enumerate(shapes)]
    n = len(probs)
    arr = (nv.WgradDesc * n)(*[p[0] for p in probs])
    dws = []
    for d in arr:
        d.phase, d.accumulate = phase, 0
        dws.append(torch.full((d.Cout * d.dw_cin * 9,), 3.0, device=DEV))
        d.dw = dws[-1].data_ptr()
    info = nv.BatchInfo()
    nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD, arr, n, None, C.byref(info)), "query")
    wss = []
    for i, d in enumerate(arr):
        ws = torch.zeros(int(info.ws_bytes[i]) // 4 + 4, device=DEV)
        d.workspace, d.workspace_bytes = ws.data_ptr(), int(info.ws_bytes[i])
        wss.append(ws)
    host = (C.c_char * int(L.hrp_batch_table_bytes(nv.BATCH_WGRAD, n)))()
    nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD, arr, n, host, C.byref(info)), "prepare")
    assert info.grid3 > 0 and info.grid > 0
    tab = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(DEV)
    nv.check(L.hrp_batch_launch(tab.data_ptr(), C.byref(info), None), "launch")
    if phase == 1:
        folds = (nv.WgradFoldDesc * n)()
        nv.check(L.hrp_batch_wgrad_fold_descs(host, C.byref(info), folds), "fold descs")
        finfo = nv.BatchInfo()
        fhost = (C.c_char * int(L.hrp_batch_table_bytes(nv.BATCH_WGRAD_FOLD, n)))()
        nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD_FOLD, folds, n, fhost, C.byref(finfo)), "fold prepare")
        ftab = torch.frombuffer(bytearray(bytes(fhost)), dtype=torch.uint8).to(DEV)
        nv.check(L.hrp_batch_launch(ftab.data_ptr(), C.byref(finfo), None), "fold launch")
    torch.cuda.synchronize()
    for (d, x, dy), dw, s in zip(probs, dws, shapes):
        ref = _reference(x, dy, s[3], s[4], stride=2).to(DEV)
        err = float((dw.view(s[4], s[3], 9).double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, f"{s}: {err:.2e}"
