"""csrc/conv3d_fl.hip - the software-pipelined flat-tile 3x3x3 convolution of the V-Net levels below full resolution
(vnetWithArgs.py:5-31: Conv3d(k=3, pad=1) of ConvBlock at 32 / 64 / 128 / 256 channels) - against igemm_kernel<9,..,FLAT,DEPTH=3>
(bit-identical: the same products in the same order) and an fp64 torch convolution (-m gpu)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(x):
    return x.cuda().contiguous(memory_format=torch.channels_last_3d)


# LA-patch levels (planes 56x40, 28x20, 14x10, 7x5), one and several volumes, several tiles per workgroup (nv=6 at 56x40: 336 planes),
# a 16-multiple width (48) and a one-chunk-per-slice K (16 -> 32), two BatchNorm groups, every tile shape the cost model picks
SHAPES = [dict(nv=1, ci=32, co=32, sp=(6, 56, 40)), dict(nv=6, ci=32, co=32, sp=(56, 56, 40), groups=2),
          dict(nv=2, ci=64, co=64, sp=(28, 28, 20), groups=2), dict(nv=4, ci=128, co=128, sp=(14, 14, 10)),
          dict(nv=4, ci=256, co=256, sp=(7, 7, 5)), dict(nv=1, ci=16, co=32, sp=(3, 9, 48)), dict(nv=2, ci=48, co=96, sp=(5, 12, 61)),
          dict(nv=1, ci=64, co=32, sp=(2, 20, 20)), dict(nv=3, ci=32, co=64, sp=(2, 5, 3))]


@pytest.mark.parametrize("shape", SHAPES)
def test_pipelined_3x3x3_kernel_equals_igemm_kernel(shape):
    from arco_amd import _lib as L, ops
    nv, ci, co, (d3, h, w) = shape["nv"], shape["ci"], shape["co"], shape["sp"]
    groups = shape.get("groups", 1)
    g = torch.Generator().manual_seed(ci * 1000 + co + h)
    x = _cl(torch.randn(nv, ci, d3, h, w, generator=g))
    wt = (torch.randn(co, ci, 3, 3, 3, generator=g) / (5.2 * ci ** 0.5)).cuda()
    bias = torch.randn(co, generator=g).cuda()
    prev_mma = ops.CONV_MMA
    ops.CONV_MMA = 3
    prev = ops.conv3d_fl_set(1)
    res = {}
    try:
        wp = ops.pack_weight(wt, 27, 0)
        xr, ldx = ops.rows_view(x)
        for on in (0, 1):
            ops.conv3d_fl_set(on)
            cfg = L.query("arco_conv_config_mma", 27, nv * d3, h, w, ci, co, ldx, 3)
            assert (9270000 <= cfg < 9300000) == bool(on), cfg      # conv3d_fl_kernel 9.27e6, conv3d_dw_kernel 9.28e6, conv3d_fc_kernel 9.29e6
            out, (ssum, ssq, nmb) = ops.conv_raw(xr, ldx, ci, wp, co, nv, h, w, 27, bias=bias, stats=True, d3=d3, stat_groups=groups)
            assert ssum.shape == (co, nmb) and nmb % groups == 0
            res[on] = (out.clone(), ssum.double().view(co, groups, -1).sum(2), ssq.double().view(co, groups, -1).sum(2))
    finally:
        ops.conv3d_fl_set(prev)
        ops.CONV_MMA = prev_mma
    assert torch.equal(res[0][0], res[1][0])
    ref = F.conv3d(x.double(), wt.double(), bias.double(), padding=1)
    assert float((res[1][0].double() - ref).abs().max() / ref.abs().max()) < 3e-6
    for k in (1, 2):
        assert torch.allclose(res[0][k], res[1][k], rtol=1e-5, atol=1e-5)
    refg = ref.view(groups, nv // groups, co, -1)
    assert torch.allclose(res[1][1].t(), refg.sum((1, 3)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(res[1][2].t(), (refg * refg).sum((1, 3)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("cfg", [44, 34, 24, 14, 42, 32, 22, 12, 92, 93, 124, 114, 152, 142, 132, 122, 112, 222, 232, 212, 214])
def test_every_tile_shape_of_the_pipelined_3x3x3_kernel(cfg, monkeypatch):
    """ARCO_CONV3D_FL_CFG is read once per process: the tile shapes are forced through a child interpreter."""
    import os, subprocess, sys
    code = r'''
import torch, torch.nn.functional as F
from arco_amd import ops
ops.CONV_MMA = 3
g = torch.Generator().manual_seed(5)
for (nv, ci, co, sp) in ((2, 64, 64, (9, 28, 20)), (1, 32, 128, (4, 14, 10)), (3, 16, 64, (3, 7, 5))):
    x = torch.randn(nv, ci, *sp, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    wt = (torch.randn(co, ci, 3, 3, 3, generator=g) / (5.2 * ci ** 0.5)).cuda()
    b = torch.randn(co, generator=g).cuda()
    xg = x.clone().requires_grad_(True)
    y = ops.conv(xg, wt, b)
    gy = torch.randn(y.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    y.backward(gy)
    ops.conv3d_fl_set(0)
    xg0 = x.clone().requires_grad_(True)
    y0 = ops.conv(xg0, wt, b); y0.backward(gy)
    ops.conv3d_fl_set(1)
    assert torch.equal(y, y0) and torch.equal(xg.grad, xg0.grad)
    ref = F.conv3d(x.double(), wt.double(), b.double(), padding=1)
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 3e-6
print("OK")
'''
    env = dict(os.environ, ARCO_CONV3D_FL_CFG=str(cfg))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


# f16 activation storage (BASELINE.json configs[4]): hconv_fc_kernel against hconv_kernel - LiTS-patch levels (planes 40x24, 20x12, 10x6; the
# 32-channel 80x48 level stays on hconv_kernel's rectangular tiles), the LA ones, one and two volumes, two BatchNorm groups
H_SHAPES = [dict(nv=1, ci=32, co=32, sp=(9, 80, 40)), dict(nv=2, ci=64, co=64, sp=(40, 40, 24), groups=2), dict(nv=2, ci=128, co=128, sp=(20, 20, 12)),
            dict(nv=2, ci=256, co=256, sp=(10, 10, 6), groups=2), dict(nv=1, ci=32, co=64, sp=(5, 56, 40)), dict(nv=3, ci=16, co=32, sp=(3, 7, 5)),
            dict(nv=1, ci=48, co=96, sp=(2, 12, 61))]


@pytest.mark.parametrize("shape", H_SHAPES)
def test_pipelined_f16_3x3x3_kernel_equals_hconv_kernel(shape):
    from arco_amd import _lib as L, ops
    nv, ci, co, (d3, h, w) = shape["nv"], shape["ci"], shape["co"], shape["sp"]
    groups = shape.get("groups", 1)
    g = torch.Generator().manual_seed(ci * 1000 + co + h)
    x = _cl(torch.randn(nv, ci, d3, h, w, generator=g).half())
    wt = (torch.randn(co, ci, 3, 3, 3, generator=g) / (5.2 * ci ** 0.5)).half().float().cuda()
    bias = torch.randn(co, generator=g).cuda()
    prev = ops.conv3d_fl_set(1)
    res = {}
    try:
        wp = ops.pack_weight(wt, 27, 0, half=True)
        xr, ldx = ops.rows_view(x)
        for on in (0, 1):
            ops.conv3d_fl_set(on)
            cfg = L.query("arco_conv_config_mma", 27, nv * d3, h, w, ci, co, ldx, 4)
            assert (9260000 <= cfg < 9270000) == bool(on), cfg
            out, (ssum, ssq, nmb) = ops.conv_raw(xr, ldx, ci, wp, co, nv, h, w, 27, bias=bias, stats=True, d3=d3, stat_groups=groups, half=True)
            assert out.dtype == torch.float16 and ssum.shape == (co, nmb) and nmb % groups == 0
            res[on] = (out.clone(), ssum.double().view(co, groups, -1).sum(2), ssq.double().view(co, groups, -1).sum(2))
    finally:
        ops.conv3d_fl_set(prev)
    assert torch.equal(res[0][0], res[1][0])
    ref = F.conv3d(x.double(), wt.double(), bias.double(), padding=1)
    assert float((res[1][0].double() - ref).abs().max() / ref.abs().max()) < 1e-3          # one f16 rounding of the stored result
    for k in (1, 2):
        assert torch.allclose(res[0][k], res[1][k], rtol=1e-5, atol=1e-4)
    got = res[1][0].double().view(groups, nv // groups, co, -1)
    assert torch.allclose(res[1][1].t(), got.sum((1, 3)), rtol=1e-4, atol=1e-2)             # statistics of the ROUNDED outputs


@pytest.mark.parametrize("cfg", [64, 54, 62, 52, 44, 34, 24, 14, 42, 32, 22, 12])
def test_every_tile_shape_of_the_pipelined_f16_kernel(cfg):
    """ARCO_HCONV_FC_CFG is read once per process: hconv_fc_kernel's tile shapes forced through a child interpreter, outputs and
    BatchNorm partial sums against hconv_kernel (two BatchNorm groups, a ragged last tile, planes narrower than a tile)."""
    import os, subprocess, sys
    code = r'''
import torch
from arco_amd import ops, _lib as L
g = torch.Generator().manual_seed(7)
for (nv, ci, co, sp, groups) in ((2, 64, 64, (9, 40, 24), 2), (1, 64, 128, (4, 20, 12), 1), (3, 16, 64, (3, 7, 5), 1), (2, 32, 64, (2, 33, 45), 1)):
    d3, h, w = sp
    x = torch.randn(nv, ci, *sp, generator=g).half().cuda().contiguous(memory_format=torch.channels_last_3d)
    wt = (torch.randn(co, ci, 3, 3, 3, generator=g) / (5.2 * ci ** 0.5)).half().float().cuda()
    b = torch.randn(co, generator=g).cuda()
    wp = ops.pack_weight(wt, 27, 0, half=True)
    xr, ldx = ops.rows_view(x)
    res = {}
    for on in (0, 1):
        ops.conv3d_fl_set(on)
        cfg = L.query("arco_conv_config_mma", 27, nv * d3, h, w, ci, co, ldx, 4)
        assert (9260000 <= cfg < 9270000) == bool(on), cfg
        out, (ssum, ssq, nmb) = ops.conv_raw(xr, ldx, ci, wp, co, nv, h, w, 27, bias=b, stats=True, d3=d3, stat_groups=groups, half=True)
        res[on] = (out.clone(), ssum.double().view(co, groups, -1).sum(2), ssq.double().view(co, groups, -1).sum(2))
    ops.conv3d_fl_set(1)
    assert torch.equal(res[0][0], res[1][0])
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-4) and torch.allclose(res[0][2], res[1][2], rtol=1e-5, atol=1e-4)
print("OK")
'''
    env = dict(os.environ, ARCO_HCONV_FC_CFG=str(cfg))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("c,sp", [(32, (56, 56, 40)), (64, (28, 28, 20)), (128, (14, 14, 10)), (256, (7, 7, 5))])
def test_la_levels_direct_sums_and_linearity(c, sp):
    """The V-Net's 3x3x3 levels at the LA patch (4 volumes, BASELINE.json configs[2]) held to properties that do not go through any
    other kernel of this library: exact direct sums at the eight corners (zero padding on every face), one edge and one interior voxel,
    linearity, and invariance of a volume's output to what the other volumes of the launch hold (no tile reads across volumes)."""
    from arco_amd import ops
    prev_mma = ops.CONV_MMA
    ops.CONV_MMA = 3
    try:
        g = torch.Generator(device="cuda").manual_seed(c)
        d, h, w = sp
        x1 = torch.randn((4, d, h, w, c), device="cuda", generator=g).permute(0, 4, 1, 2, 3)
        x2 = torch.randn((4, d, h, w, c), device="cuda", generator=g).permute(0, 4, 1, 2, 3)
        wt = torch.randn((c, c, 3, 3, 3), device="cuda", generator=g) / (27 * c) ** 0.5
        with torch.no_grad():
            y1, y2 = ops.conv(x1, wt, None), ops.conv(x2, wt, None)
            y12 = ops.conv((0.5 * x1 + x2).contiguous(memory_format=torch.channels_last_3d), wt, None)
            assert float(((0.5 * y1 + y2) - y12).abs().max()) < 2e-5 * float(y12.abs().max()) + 1e-6
            xm = x1.clone(); xm[1:] = x2[1:]
            ym = ops.conv(xm.contiguous(memory_format=torch.channels_last_3d), wt, None)
            assert torch.equal(ym[0], y1[0]) and torch.equal(ym[1:], y2[1:])
            xp = torch.nn.functional.pad(x1[3:].double(), (1, 1, 1, 1, 1, 1))
            pts = [(a, b, e) for a in (0, d - 1) for b in (0, h - 1) for e in (0, w - 1)] + [(d // 2, 0, w // 2), (d // 2, h // 2, w // 2)]
            for (a, b, e) in pts:
                ref = (wt.double() * xp[0, :, a:a + 3, b:b + 3, e:e + 3].unsqueeze(0)).sum(dim=(1, 2, 3, 4))
                assert float((y1[3, :, a, b, e].double() - ref).abs().max()) < 3e-6 * float(ref.abs().max()) + 1e-6, (a, b, e)
    finally:
        ops.CONV_MMA = prev_mma
