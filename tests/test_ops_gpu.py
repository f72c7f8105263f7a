"""HIP kernels (through the C ABI) vs golden vectors and the oracle.  Needs an MI355X."""
import copy
import math
import os

import numpy as np
import pytest
import torch

from oracle import pavenet_ref as R
from oracle.seeded import seeded_array
from tests.fused_expected import grid_expected, pose_expected

pytestmark = pytest.mark.gpu
LEVELS = [(12, 20), (6, 10), (3, 5), (2, 3)]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _levels(levels, dev='cuda'):
    shapes = torch.as_tensor(levels, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    return shapes, lsi, shapes.to(dev), lsi.to(dev)


def test_mmcv_seed3_known_answer(golden_dir):
    """The reference's own test for this op (test_ms_deformable_attn.py:73-135), with its tolerances."""
    from pavenet_amd.ops import MultiScaleDeformableAttnFunction
    g = np.load(os.path.join(golden_dir, 'op_msda.npz'))
    shapes = _t(g['s3_shapes']).cuda()
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    v, loc, aw = _t(g['s3_value']).cuda(), _t(g['s3_loc']).cuda(), _t(g['s3_aw']).cuda()
    out64 = MultiScaleDeformableAttnFunction.apply(v.double(), shapes, lsi, loc.double(),
                                                   aw.double(), 2).cpu().numpy()
    ref64 = g['s3_out_f64']
    assert np.abs(out64 - ref64).max() < 1e-18
    assert (np.abs(out64 - ref64) / np.abs(ref64)).max() < 1e-15
    out32 = MultiScaleDeformableAttnFunction.apply(v, shapes, lsi, loc, aw, 2).cpu().numpy()
    ref32 = g['s3_out_f32']
    assert np.allclose(out32, ref32, rtol=1e-2, atol=1e-3)
    assert np.abs(out32 - ref32).max() < 1e-9
    assert (np.abs(out32 - ref32) / np.abs(ref32)).max() < 1e-6


class _ExtFunction(torch.autograd.Function):
    """The autograd Function of MO:20-89 written against an `ext_module` exactly as the reference
    calls it (positional tensors, `im2col_step=` keyword): what mmcv runs on top of `mmcv._ext`."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        from pavenet_amd import _ext as ext_module
        ctx.im2col_step = im2col_step
        output = ext_module.ms_deform_attn_forward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations,
            attention_weights, im2col_step=ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        from pavenet_amd import _ext as ext_module
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        grad_value, grad_loc, grad_aw = (torch.zeros_like(t) for t in (value, loc, aw))
        ext_module.ms_deform_attn_backward(
            value, shapes, lsi, loc, aw, grad_output.contiguous(), grad_value, grad_loc, grad_aw,
            im2col_step=ctx.im2col_step)
        return grad_value, None, None, grad_loc, grad_aw, None


def test_pybind_ext_seed3_known_answer_and_gradcheck(golden_dir):
    """The boundary as a built, loaded, executed pybind module (pavenet_amd/_ext*.so, the
    mmcv._ext surface of pybind.cpp:737-748): mmcv's seed-3 known answer at mmcv's tolerances
    (test_ms_deformable_attn.py:73-135), its error behaviour, and torch.autograd.gradcheck through
    an MO-style Function on top of it (test_ms_deformable_attn.py:138-182, fp64)."""
    from pavenet_amd import _ext
    g = np.load(os.path.join(golden_dir, 'op_msda.npz'))
    shapes = _t(g['s3_shapes']).cuda()
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    v, loc, aw = _t(g['s3_value']).cuda(), _t(g['s3_loc']).cuda(), _t(g['s3_aw']).cuda()
    out64 = _ext.ms_deform_attn_forward(v.double(), shapes, lsi, loc.double(), aw.double(),
                                        im2col_step=2).cpu().numpy()
    ref64 = g['s3_out_f64']
    assert np.abs(out64 - ref64).max() < 1e-18
    assert (np.abs(out64 - ref64) / np.abs(ref64)).max() < 1e-15
    out32 = _ext.ms_deform_attn_forward(value=v, value_spatial_shapes=shapes,
                                        value_level_start_index=lsi, sampling_locations=loc,
                                        attention_weights=aw, im2col_step=2).cpu().numpy()
    ref32 = g['s3_out_f32']
    assert np.abs(out32 - ref32).max() < 1e-9
    assert (np.abs(out32 - ref32) / np.abs(ref32)).max() < 1e-6
    with pytest.raises(RuntimeError, match='contiguous'):
        _ext.ms_deform_attn_forward(v.expand(2, -1, -1, -1)[:, ::1].transpose(2, 3), shapes, lsi,
                                    loc, aw, im2col_step=2)
    with pytest.raises(RuntimeError):   # batch 3 % im2col_step 2 (ms_deform_attn_cuda.cu:242-245)
        _ext.ms_deform_attn_forward(v.repeat(3, 1, 1, 1), shapes, lsi, loc.repeat(3, 1, 1, 1, 1, 1),
                                    aw.repeat(3, 1, 1, 1, 1), im2col_step=2)
    # gradcheck, mmcv's own recipe (N, M = 1, 2; Lq, L, P = 2, 2, 2; shapes (3, 2), (2, 1))
    for D in (4, 30, 32):
        gs = torch.as_tensor([(3, 2), (2, 1)], dtype=torch.long).cuda()
        gl = torch.cat((gs.new_zeros((1,)), gs.prod(1).cumsum(0)[:-1]))
        S = int(gs.prod(1).sum())
        gen = torch.Generator().manual_seed(D)
        value = (torch.rand(1, S, 2, D, generator=gen) * 0.01).double().cuda().requires_grad_(True)
        sl = torch.rand(1, 2, 2, 2, 2, 2, generator=gen).double().cuda().requires_grad_(True)
        w = torch.rand(1, 2, 2, 2, 2, generator=gen).double() + 1e-5
        w = (w / w.sum(-1, keepdim=True).sum(-2, keepdim=True)).cuda().requires_grad_(True)
        assert torch.autograd.gradcheck(_ExtFunction.apply, (value, gs, gl, sl, w, 2))


@pytest.mark.parametrize('case', ['enc', 'pose', 'joint', 'odd', 'd71'])
def test_sampler_golden(golden_dir, case):
    from pavenet_amd.ops import ms_deform_attn_forward
    g = np.load(os.path.join(golden_dir, 'op_msda.npz'))
    shapes = _t(g['levels']).cuda()
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    out = ms_deform_attn_forward(_t(g[f'{case}_value']).cuda(), shapes, lsi,
                                 _t(g[f'{case}_loc']).cuda(), _t(g[f'{case}_aw']).cuda(), 64)
    np.testing.assert_allclose(out.cpu().numpy(), g[f'{case}_out'], rtol=1e-5, atol=2e-6)


def test_sampler_errors():
    from pavenet_amd.ops import ms_deform_attn_forward
    shapes, lsi, sd, ld = _levels(LEVELS)
    S = int(shapes.prod(1).sum())
    v = torch.zeros(3, S, 8, 32, device='cuda')
    loc = torch.zeros(3, 5, 8, 4, 4, 2, device='cuda')
    aw = torch.zeros(3, 5, 8, 4, 4, device='cuda')
    with pytest.raises(RuntimeError):  # batch % im2col_step != 0 (ms_deform_attn_cuda.cu:242-245)
        ms_deform_attn_forward(v, sd, ld, loc, aw, 2)
    with pytest.raises(RuntimeError):  # CPU tensor
        ms_deform_attn_forward(v.cpu(), sd, ld, loc, aw, 64)
    with pytest.raises(RuntimeError):  # non-contiguous
        ms_deform_attn_forward(v.transpose(0, 1).contiguous().transpose(0, 1), sd, ld, loc, aw, 64)
    assert ms_deform_attn_forward(v, sd, ld, loc, aw, 3).shape == (3, 5, 256)


def test_sampler_edge_cases():
    """Empty query set, a single query / single point, and locations far outside the maps
    (zero padding: every corner is out of range -> exact zeros), as the PyTorch formulation."""
    from pavenet_amd.ops import ms_deform_attn_forward, oks_nms
    shapes, lsi, sd, ld = _levels(LEVELS)
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(1)
    v = torch.randn(2, S, 8, 32, generator=g).cuda()
    out = ms_deform_attn_forward(v, sd, ld, torch.zeros(2, 0, 8, 4, 4, 2, device='cuda'),
                                 torch.zeros(2, 0, 8, 4, 4, device='cuda'), 64)
    assert out.shape == (2, 0, 256)
    loc = torch.rand(2, 1, 8, 4, 1, 2, generator=g)
    aw = torch.rand(2, 1, 8, 4, 1, generator=g)
    got = ms_deform_attn_forward(v, sd, ld, loc.cuda(), aw.cuda(), 64)
    exp = R.msda_forward_torch(v.cpu(), shapes, loc, aw)
    np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    far = torch.full((2, 7, 8, 4, 4, 2), 7.5)
    far[..., 1] = -3.25
    z = ms_deform_attn_forward(v, sd, ld, far.cuda(), torch.ones(2, 7, 8, 4, 4, device='cuda'), 64)
    assert torch.count_nonzero(z) == 0
    # exactly on the border: pixel = loc * W - 0.5 in (-1, 0) keeps one row / column of corners
    edge = torch.zeros(2, 3, 8, 4, 4, 2)
    w1 = torch.rand(2, 3, 8, 4, 4, generator=g)
    got = ms_deform_attn_forward(v, sd, ld, edge.cuda(), w1.cuda(), 64)
    exp = R.msda_forward_torch(v.cpu(), shapes, edge, w1)
    np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    keep, order = oks_nms(torch.zeros(2, 0, 15, 3, device='cuda'), torch.zeros(2, 0, device='cuda'),
                          _t(R.OKS_SIGMAS_15).cuda(), 0.45)
    assert keep.shape == (2, 0) and order.shape == (2, 0)


@pytest.mark.parametrize('T,U,clips', [(1, 321, 1), (1, 643, 2), (2, 75, 2), (3, 75, 2), (4, 33, 1),
                                       (5, 45, 1), (7, 30, 2)])
def test_grid_fused_vs_oracle(T, U, clips):
    from pavenet_amd.ops import deform_attn_grid_fused
    shapes, lsi, sd, ld = _levels(LEVELS)
    S = int(shapes.prod(1).sum())
    value = _t(seeded_array(f'gf.value.{T}', (clips * T, S, 8, 32)))
    proj = _t(seeded_array(f'gf.proj.{T}', (U, T * 8 * 16 * 3)))
    proj[:, :T * 8 * 16 * 2] *= 2.0       # offsets: a few pixels
    ref = _t(seeded_array(f'gf.ref.{T}', (T, U, 4, 2), 0.35)) + 0.5  # some outside [0,1]
    unit_clip = (torch.arange(U) % clips).to(torch.int32)
    exp = grid_expected(value, shapes, lsi, proj, ref, T, unit_clip.long())
    order = torch.randperm(U, generator=torch.Generator().manual_seed(0)).to(torch.int32)
    for od in (None, order.cuda()):
        out, smax, ssum = deform_attn_grid_fused(
            value.cuda(), sd, ld, proj.cuda(), ref.cuda(), T=T, n_clips=clips,
            units_per_clip=U, unit_clip=unit_clip.cuda(), order=od, return_stats=True)
        np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)
    lg = proj[:, T * 8 * 16 * 2:].view(U, T, 8, 16).permute(0, 2, 1, 3).reshape(U, 8, -1)
    np.testing.assert_allclose(smax.cpu().numpy(), lg.max(-1)[0].numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(ssum.cpu().numpy(),
                               torch.exp(lg - lg.max(-1, keepdim=True)[0]).sum(-1).numpy(),
                               rtol=1e-5)


def test_grid_fused_T1_unaligned_rows_take_the_per_query_kernel():
    """T = 1 projections whose row stride is not a multiple of 4 floats cannot use the head-major
    kernel's 16-byte loads: the C entry point falls back to the per-query form
    (fused_deform_attn_kernel<GRID, 2, 1>).  Reached through the C ABI directly (the torch wrapper
    only hands over dense rows); same numbers as the oracle."""
    from pavenet_amd import native
    lib = native.load()
    shapes, lsi, sd, ld = _levels(LEVELS)
    S = int(shapes.prod(1).sum())
    U, stride = 37, 385
    value = _t(seeded_array('gf1u.value', (1, S, 8, 32)))
    wide = _t(seeded_array('gf1u.proj', (U, stride)))
    wide[:, :256] *= 2.0
    proj = wide[:, :384].contiguous()
    ref = _t(seeded_array('gf1u.ref', (1, U, 4, 2), 0.35)) + 0.5
    exp = grid_expected(value, shapes, lsi, proj, ref, 1, torch.zeros(U, dtype=torch.long))
    vd, pd, rd = value.cuda(), wide.cuda(), ref.cuda()
    out = torch.empty(U, 256, device='cuda')
    st = lib.pave_deform_attn_grid_fused_f32(
        vd.data_ptr(), sd.data_ptr(), ld.data_ptr(), pd.data_ptr(), rd.data_ptr(), None, None,
        out.data_ptr(), None, None, U, U, 1, 1, S, 4, 4, stride, None, 0, 4,
        torch.cuda.current_stream().cuda_stream)
    native.check(st, 'grid_fused (unaligned rows)')
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('T,clips,Q,K', [(1, 2, 9, 17), (2, 2, 10, 15), (3, 2, 10, 15), (5, 1, 7, 15),
                                         (7, 1, 5, 15)])
def test_pose_fused_vs_oracle(T, clips, Q, K):
    from pavenet_amd.ops import deform_attn_pose_fused
    shapes, lsi, sd, ld = _levels(LEVELS)
    S = int(shapes.prod(1).sum())
    L = 4
    value = _t(seeded_array(f'pf.value.{T}', (clips * T, S, 8, 32)))
    proj = _t(seeded_array(f'pf.proj.{T}', (clips * Q, T * 8 * L * K * 3)))
    ref = torch.sigmoid(_t(seeded_array(f'pf.ref.{T}', (clips, T * Q, L, 2 * K), 1.0)))
    exp = pose_expected(value, shapes, lsi, proj, ref, T, clips, Q, K)
    out = deform_attn_pose_fused(value.cuda(), sd, ld, proj.cuda(), ref.cuda(), T=T,
                                 n_clips=clips, num_query=Q, num_keypoints=K)
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)


def test_softmax_is_stabilised():
    """|logit| > 88 overflows the reference's un-stabilised exp (OT:1737-1740, flagged BUG by
    its author); the fused kernel must stay finite and equal the mathematically exact result."""
    from pavenet_amd.ops import deform_attn_grid_fused
    shapes, lsi, sd, ld = _levels(LEVELS)
    S = int(shapes.prod(1).sum())
    T, U = 3, 8
    value = _t(seeded_array('stab.value', (T, S, 8, 32)))
    proj = _t(seeded_array('stab.proj', (U, T * 8 * 16 * 3)))
    base = proj.clone()
    proj[:, T * 8 * 16 * 2:] += 200.0  # shift every logit: softmax is shift-invariant
    ref = _t(seeded_array('stab.ref', (T, U, 4, 2), 0.2)) + 0.5
    kw = dict(T=T, n_clips=1, units_per_clip=U)
    a = deform_attn_grid_fused(value.cuda(), sd, ld, base.cuda(), ref.cuda(), **kw)
    b = deform_attn_grid_fused(value.cuda(), sd, ld, proj.cuda(), ref.cuda(), **kw)
    assert torch.isfinite(b).all()
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_frame_table_entries_outside_the_value_cache_cannot_fault():
    """The fused T-frame kernels address `value` through a frame table of slab indices read on the device
    (streaming: per-frame caches shared by overlapping windows).  The C entry takes the number of slabs and the
    kernels clamp every index into it: a table with entries far outside the cache gives exactly what the
    clamped table gives -- a wrong frame, never an access outside the tensor (advisor finding, round 3)."""
    from pavenet_amd.ops import deform_attn_grid_fused, deform_attn_pose_fused
    shapes, lsi, sd, ld = _levels(LEVELS)
    S = int(shapes.prod(1).sum())
    T, U, n_cached = 3, 8, 5
    value = _t(seeded_array('ft.value', (n_cached, S, 8, 32))).cuda()
    proj = _t(seeded_array('ft.proj', (U, T * 8 * 16 * 3))).cuda()
    ref = (_t(seeded_array('ft.ref', (T, U, 4, 2), 0.2)) + 0.5).cuda()
    bad = torch.tensor([4, 2 ** 30, -7], dtype=torch.int32).cuda()
    ok = torch.tensor([4, n_cached - 1, 0], dtype=torch.int32).cuda()
    kw = dict(T=T, n_clips=1, units_per_clip=U)
    a = deform_attn_grid_fused(value, sd, ld, proj, ref, frame_table=bad, **kw)
    b = deform_attn_grid_fused(value, sd, ld, proj, ref, frame_table=ok, **kw)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    K = 15
    pproj = _t(seeded_array('ft.pproj', (U, T * 8 * 4 * K * 3))).cuda()
    pref = (_t(seeded_array('ft.pref', (1, T * U, 4, 2 * K), 0.2)) + 0.5).cuda()
    pk = dict(T=T, n_clips=1, num_query=U, num_keypoints=K)
    a = deform_attn_pose_fused(value, sd, ld, pproj, pref, frame_table=bad, **pk)
    b = deform_attn_pose_fused(value, sd, ld, pproj, pref, frame_table=ok, **pk)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize('T', [1, 3, 5])
def test_level_broadcast_reference_points_equal_the_materialised_copy(T):
    """Un-padded batches: the decoders pass `reference_points[:, :, None].expand(.., L, ..)` (OT:6712-6720,
    MT:845-856).  The fused kernels read such a tensor as ONE row per entry (ref_levels = 1) -- bit-identical to
    the materialised [.., L, ..] copy, pose and grid form, with and without a frame table."""
    from pavenet_amd.ops import deform_attn_grid_fused, deform_attn_pose_fused
    shapes, lsi, sd, ld = _levels(LEVELS)
    S = int(shapes.prod(1).sum())
    clips, Q, K, L = 2, 11, 15, 4
    value = _t(seeded_array(f'lb.value.{T}', (clips * T, S, 8, 32))).cuda()
    proj = _t(seeded_array(f'lb.proj.{T}', (clips * Q, T * 8 * L * K * 3))).cuda()
    base = torch.sigmoid(_t(seeded_array(f'lb.ref.{T}', (clips, T * Q, 2 * K), 1.0))).cuda()
    ref = base[:, :, None].expand(-1, -1, L, -1)
    assert not ref.is_contiguous()
    kw = dict(T=T, n_clips=clips, num_query=Q, num_keypoints=K)
    a = deform_attn_pose_fused(value, sd, ld, proj, ref, **kw)
    b = deform_attn_pose_fused(value, sd, ld, proj, ref.contiguous(), **kw)
    assert torch.equal(a, b)
    table = torch.arange(clips * T - 1, -1, -1, dtype=torch.int32).cuda()
    a = deform_attn_pose_fused(value, sd, ld, proj, ref, frame_table=table, **kw)
    b = deform_attn_pose_fused(value, sd, ld, proj, ref.contiguous(), frame_table=table, **kw)
    assert torch.equal(a, b)
    U = clips * Q
    gproj = _t(seeded_array(f'lb.gproj.{T}', (U, T * 8 * 16 * 3))).cuda()
    gbase = (_t(seeded_array(f'lb.gref.{T}', (T * U, 1, 2), 0.2)) + 0.5).cuda()
    gref = gbase[:, :, None].expand(-1, -1, L, -1).reshape(T, U, L, 2)     # the joint decoder's view
    assert gref.stride(2) == 0
    uc = torch.arange(U, dtype=torch.int32).cuda() // Q
    gk = dict(T=T, n_clips=clips, units_per_clip=Q, unit_clip=uc)
    a = deform_attn_grid_fused(value, sd, ld, gproj, gref, **gk)
    b = deform_attn_grid_fused(value, sd, ld, gproj, gref.contiguous(), **gk)
    assert torch.equal(a, b)


@pytest.mark.parametrize('B,H,W,heads,shift', [(2, 14, 21, 3, 0), (2, 14, 21, 3, 3), (1, 20, 33, 6, 3),
                                               (3, 7, 7, 12, 0), (1, 5, 9, 3, 3), (1, 29, 48, 6, 3)])
def test_swin_window_attention_kernel_vs_reference_formulation(B, H, W, heads, shift):
    """pave_swin_window_attn_f32 (pad to the window, roll, partition, relative-position bias, the -100 mask
    between roll regions, softmax, PV, reverse, un-roll and crop as index arithmetic of ONE kernel on the
    un-partitioned token map) against ShiftWindowMSA / WindowMSA in their torch formulation (the restatement of
    mmdet/models/backbones/swin.py:22-286 that is bit-identical to the reference on CPU) in fp64: maps that are
    and are not multiples of the window, maps smaller than a window, shifted and un-shifted blocks."""
    from pavenet_amd import ops
    from pavenet_amd.swin import ShiftWindowMSA
    torch.manual_seed(B * 100 + H + heads + shift)
    C = 32 * heads
    m = ShiftWindowMSA(C, heads, 7, shift_size=shift).double()
    with torch.no_grad():
        m.w_msa.relative_position_bias_table.normal_(0, 0.5)
        m.w_msa.qkv.bias.normal_(0, 0.3)
        m.w_msa.proj.weight.copy_(torch.eye(C, dtype=torch.float64))      # the kernel stops in front of `proj`
        m.w_msa.proj.bias.zero_()
    x = torch.randn(B, H * W, C, dtype=torch.float64)
    with torch.no_grad():
        exp = m(x, (H, W))
        qkv = m.w_msa.qkv(x).float().view(B, H, W, 3 * C)
        n = 49
        bt = m.w_msa.relative_position_bias_table[m.w_msa.relative_position_index.view(-1)].view(n, n, heads)
        bt = bt.permute(2, 1, 0).contiguous().float()
    got = ops.swin_window_attn(qkv.cuda(), bt.cuda(), m.w_msa.qkv.bias.float().cuda(), heads, 7, shift, m.w_msa.scale)
    np.testing.assert_allclose(got.view(B, H * W, C).cpu().numpy(), exp.numpy(), rtol=2e-4, atol=2e-5)
    # the shipped form runs the two products on the fp32 matrix pipe; the per-lane form (diag variant 18) must
    # agree with it up to the summation order
    from pavenet_amd import native
    with native.diag_build(18):
        lanes = ops.swin_window_attn(qkv.cuda(), bt.cuda(), m.w_msa.qkv.bias.float().cuda(), heads, 7, shift,
                                     m.w_msa.scale)
    np.testing.assert_allclose(got.cpu().numpy(), lanes.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(lanes.view(B, H * W, C).cpu().numpy(), exp.numpy(), rtol=2e-4, atol=2e-5)


def test_gemm_gelu_epilogue_and_wide_layernorm_rows():
    """What the Swin blocks need beside the window kernel: the exact-GELU epilogue of the split GEMM (tile and
    small-row forms, against fp64 x Phi(x)) and pave_bias_add_layernorm_f32 on 1 536- and 3 072-wide rows."""
    from pavenet_amd import ops
    g = torch.Generator().manual_seed(21)
    for M in (300, 9000):
        a, w, b = torch.randn(M, 192, generator=g), torch.randn(768, 192, generator=g) * 0.1, torch.randn(768, generator=g)
        got = ops.gemm_bf16x3(a.cuda(), ops.split_weight_bf16x3(w.cuda()), b.cuda(), relu='gelu')
        exp = torch.nn.functional.gelu(a.double() @ w.double().t() + b.double())
        np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=2e-6)
        # the sigmoid epilogue (the heads' sigma branches): the fp32 GEMM result through torch.sigmoid's expression
        lin = ops.gemm_bf16x3(a.cuda(), ops.split_weight_bf16x3(w.cuda()), b.cuda())
        got = ops.gemm_bf16x3(a.cuda(), ops.split_weight_bf16x3(w.cuda()), b.cuda(), relu='sigmoid')
        np.testing.assert_allclose(got.cpu().numpy(), lin.sigmoid().cpu().numpy(), rtol=0, atol=1.2e-7)
    # ... and through the heads' helper on a 2-output branch (planes padded to 64 columns, output to 4)
    from pavenet_amd.bricks import mlp_rows
    br = torch.nn.Sequential(torch.nn.Linear(192, 64), torch.nn.ReLU(), torch.nn.Linear(64, 2)).cuda()
    x = torch.randn(3, 50, 192, generator=g).cuda()
    from pavenet_amd.bricks import get_gemm_mode, set_gemm_mode
    mode = get_gemm_mode()
    set_gemm_mode('bf16x3')
    try:
        with torch.no_grad():
            got, exp = mlp_rows(br, x, act='sigmoid'), br(x).sigmoid()
    finally:
        set_gemm_mode(mode)
    assert got.shape == exp.shape and got.stride(-2) == 4         # the padded matrix's column slice
    np.testing.assert_allclose(got.cpu().numpy(), exp.cpu().numpy(), rtol=0, atol=2e-6)
    for C in (1536, 3072, 1028):
        x, ga, be = torch.randn(77, C, generator=g) * 3 + 1, torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
        got = ops.bias_add_layernorm(x.cuda(), None, None, ga.cuda(), be.cuda(), 1e-5)
        exp = torch.nn.functional.layer_norm(x.double(), (C,), ga.double(), be.double(), 1e-5)
        np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)


def test_merge_softmax_partials_kernel_vs_host_formulation():
    """pave_merge_softmax_partials_f32 (the frame-sharded attentions' exact softmax merge as one launch) against
    pavenet_amd.dist.merge_softmax_partials in fp64: dominated ranks, a rank without frames (sum = 0, max = -inf,
    garbage row) and G = 1 ... 8."""
    from pavenet_amd import dist as pd
    from pavenet_amd.ops import merge_softmax_partials
    g = torch.Generator().manual_seed(9)
    for G in (1, 2, 5, 8):
        U, C, H = 300, 256, 8
        rows = torch.randn(G, U, C, generator=g)
        smax = torch.randn(G, U, H, generator=g) * 10
        ssum = torch.rand(G, U, H, generator=g) * 50 + 0.01
        if G > 2:
            smax[1] += 80.0                       # one rank dominates
            ssum[2] = 0.0                         # a rank that owns no frame: dropped, its row not read
            smax[2] = float('-inf')
            rows[2] = float('nan')
        parts = torch.cat([rows, smax, ssum], 2).cuda()
        got = merge_softmax_partials(parts, C, H).cpu()
        exp = pd.merge_softmax_partials(torch.nan_to_num(rows).double(), smax.double(), ssum.double(), H)
        np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=1e-5, atol=1e-6)


def test_full_size_linearity():
    """At BASELINE's full size (S = 22 323, 800x1344) the oracle is too slow to run whole:
    check size-independent properties instead -- linearity in value, a constant map sampled
    at in-range points returns the constant, and a sub-sample of rows against the oracle."""
    from pavenet_amd.ops import deform_attn_grid_fused
    levels = [(100, 168), (50, 84), (25, 42), (13, 21)]
    shapes, lsi, sd, ld = _levels(levels)
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(1)
    value = torch.randn(1, S, 8, 32, generator=g)
    value2 = torch.randn(1, S, 8, 32, generator=g)
    U = S
    proj = torch.randn(U, 8 * 16 * 3, generator=g)
    proj[:, :256] *= 3.0
    ys = torch.cat([((torch.arange(h * w) // w).float() + 0.5) / h for h, w in levels])
    xs = torch.cat([((torch.arange(h * w) % w).float() + 0.5) / w for h, w in levels])
    ref = torch.stack([xs, ys], -1)[None, :, None, :].expand(1, U, 4, 2).contiguous()
    kw = dict(T=1, n_clips=1, units_per_clip=U)
    vd, v2d, pd, rd = value.cuda(), value2.cuda(), proj.cuda(), ref.cuda()
    a = deform_attn_grid_fused(vd, sd, ld, pd, rd, **kw)
    b = deform_attn_grid_fused(v2d, sd, ld, pd, rd, **kw)
    ab = deform_attn_grid_fused(2.0 * vd - 3.0 * v2d, sd, ld, pd, rd, **kw)
    np.testing.assert_allclose(ab.cpu().numpy(), (2.0 * a - 3.0 * b).cpu().numpy(),
                               rtol=1e-4, atol=1e-4)
    # interior queries with zero offsets on a constant map -> the constant
    ones = torch.ones_like(vd)
    p0 = pd.clone()
    p0[:, :256] = 0
    c = deform_attn_grid_fused(ones, sd, ld, p0, rd, **kw).cpu()
    inner = ((xs > 0.1) & (xs < 0.9) & (ys > 0.1) & (ys < 0.9))  # border samples lose weight
    np.testing.assert_allclose(c[inner].numpy(), np.ones_like(c[inner].numpy()), rtol=1e-5)
    assert float(c.max()) <= 1.0 + 1e-5
    # sub-sample against the oracle
    idx = torch.arange(0, U, 97)
    exp = grid_expected(value, shapes, lsi, proj[idx], ref[:, idx], 1,
                        torch.zeros(len(idx), dtype=torch.long))
    np.testing.assert_allclose(a.cpu()[idx].numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('C', [256, 1024, 64, 260])
def test_bias_add_layernorm_vs_torch(C):
    from pavenet_amd.ops import bias_add_layernorm
    g = torch.Generator().manual_seed(C)
    x = torch.randn(37, 5, C, generator=g) * 3
    res = torch.randn(37, 5, C, generator=g)
    bias, gamma, beta = (torch.randn(C, generator=g) for _ in range(3))
    exp = torch.nn.functional.layer_norm(x + bias + res, (C,), gamma, beta, 1e-5)
    out = bias_add_layernorm(x.cuda(), bias.cuda(), res.cuda(), gamma.cuda(), beta.cuda(), 1e-5)
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)
    out = bias_add_layernorm(x.cuda(), None, None, gamma.cuda(), beta.cuda(), 1e-5)
    exp = torch.nn.functional.layer_norm(x, (C,), gamma, beta, 1e-5)
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)
    # second output: y + pos, with a pos table shared by the 37 leading rows ([5, C]) or full
    for pos in (torch.randn(5, C, generator=g), torch.randn(37 * 5, C, generator=g)):
        y, yp = bias_add_layernorm(x.cuda(), bias.cuda(), res.cuda(), gamma.cuda(), beta.cuda(),
                                   1e-5, pos=pos.cuda())
        exp = torch.nn.functional.layer_norm(x + bias + res, (C,), gamma, beta, 1e-5)
        np.testing.assert_allclose(y.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)
        assert torch.equal(yp.cpu(), y.cpu() + pos.view(-1, 5, C).expand(37, 5, C)
                           if pos.shape[0] == 5 else y.cpu() + pos.view(37, 5, C))


def test_bias_act_rows_vs_torch():
    from pavenet_amd.ops import bias_act_rows_
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 64, 9, 11, generator=g)
    res = torch.randn(3, 64, 9, 11, generator=g)
    bias = torch.randn(64, generator=g)
    exp = torch.relu(x + bias.view(1, -1, 1, 1) + res)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    rd = res.cuda().contiguous(memory_format=torch.channels_last)
    out = bias_act_rows_(xd, bias.cuda(), rd, relu=True)
    assert out.data_ptr() == xd.data_ptr()
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=0, atol=0)
    t = torch.randn(50, 128, generator=g)
    out = bias_act_rows_(t.cuda(), bias.repeat(2).cuda(), None, relu=False)
    np.testing.assert_allclose(out.cpu().numpy(), (t + bias.repeat(2)).numpy(), rtol=0, atol=0)


@pytest.mark.parametrize('sigma', [0.7, 3.0, 12.0])
@pytest.mark.parametrize('gr', [False, True])
@pytest.mark.parametrize('variant', [0, 1])
@pytest.mark.parametrize('levels', [[(24, 40), (12, 20), (6, 10), (3, 5)],
                                    [(25, 42), (13, 21), (7, 11), (4, 6)],
                                    [(16, 20), (8, 10), (4, 5), (2, 3)]])
def test_enc_tile_kernel_equals_direct_and_oracle(levels, sigma, gr, variant):
    """LDS-tile encoder kernel == direct-gather kernel == oracle, for small offsets (all corners
    served from LDS), medium and huge offsets (mostly the global second pass), maps whose sizes are
    not multiples of the tile, and reference points scaled by valid ratios (padded frames)."""
    from pavenet_amd.ops import deform_attn_enc_tile, deform_attn_grid_fused, enc_tile_supported
    assert enc_tile_supported(levels)
    shapes, lsi, sd, ld = _levels(levels)
    S = int(shapes.prod(1).sum())
    F = 2
    value = _t(seeded_array(f'win.value.{S}', (F, S, 8, 32)))
    proj = _t(seeded_array(f'win.proj.{S}.{sigma}', (F * S, 384)))
    proj[:, :256] *= sigma
    vr = (torch.ones(F, 2) if gr else torch.tensor([[1.0, 1.0], [0.83, 0.9]])).view(F, 1, 1, 2)
    ys = torch.cat([((torch.arange(h * w) // w).float() + 0.5) / h for h, w in levels])
    xs = torch.cat([((torch.arange(h * w) % w).float() + 0.5) / w for h, w in levels])
    ref = (torch.stack([xs, ys], -1)[None, :, None, :] * vr).expand(F, S, 4, 2)
    ref = ref.reshape(1, F * S, 4, 2).contiguous()
    a = deform_attn_enc_tile(value.cuda(), proj.cuda(), ref.cuda(), levels_hw=levels,
                             variant=variant)
    b = deform_attn_grid_fused(value.cuda(), sd, ld, proj.cuda(), ref.cuda(), T=1, n_clips=F,
                               units_per_clip=S)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-5)
    idx = torch.arange(0, F * S, 7)
    exp = grid_expected(value, shapes, lsi, proj[idx], ref[:, idx], 1, idx // S)
    np.testing.assert_allclose(a.cpu()[idx].numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('variant', [0, 1])
def test_enc_tile_window_shift_changes_nothing_but_speed(variant):
    """Per-(head, level) window shifts (incl. shifts that push whole windows off the map) give
    the same results (up to the summation order of the two passes): the window only decides which
    corners come from LDS."""
    from pavenet_amd.ops import deform_attn_enc_tile, enc_tile_window_shift
    levels = [(24, 40), (12, 20), (6, 10), (3, 5)]
    S = sum(h * w for h, w in levels)
    F = 2
    g = torch.Generator().manual_seed(11)
    value = torch.randn(F, S, 8, 32, generator=g).cuda()
    proj = torch.randn(F * S, 384, generator=g)
    bias = torch.zeros(8, 4, 4, 2)
    for h in range(8):                       # the reference's ray initialisation, MO:227-240
        d = torch.tensor([math.cos(h * math.pi / 4), math.sin(h * math.pi / 4)])
        d = d / d.abs().max()
        for i in range(4):
            bias[h, :, i] = d * (i + 1)
    proj[:, :256] = proj[:, :256] * 0.5 + bias.reshape(1, 256)
    proj = proj.cuda()
    ys = torch.cat([((torch.arange(h * w) // w).float() + 0.5) / h for h, w in levels])
    xs = torch.cat([((torch.arange(h * w) % w).float() + 0.5) / w for h, w in levels])
    ref = torch.stack([xs, ys], -1)[None, :, None, :].expand(F, S, 4, 2).reshape(1, F * S, 4, 2)
    ref = ref.contiguous().cuda()
    base = deform_attn_enc_tile(value, proj, ref, levels_hw=levels, variant=variant)
    ray = enc_tile_window_shift(bias.reshape(-1).cuda())
    assert len(ray) == 64 and 2 <= max(abs(v) for v in ray) <= 3     # mean of 1..4 px on the ray
    for shift in (ray, tuple((-1) ** i * (i % 9) for i in range(64)), (60,) * 64):
        out = deform_attn_enc_tile(value, proj, ref, levels_hw=levels, variant=variant,
                                   window_shift=shift)
        np.testing.assert_allclose(out.cpu().numpy(), base.cpu().numpy(), rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError):
        deform_attn_enc_tile(value, proj, ref, levels_hw=levels, window_shift=(100,) * 64)


def test_enc_tile_kernel_rejects_non_pyramid_and_survives_non_finite():
    """A pyramid that is not a halving one is refused (callers use the direct kernel); NaN / Inf
    offsets, logits or reference points never become out-of-range accesses."""
    from pavenet_amd.ops import deform_attn_enc_tile, enc_tile_supported
    assert not enc_tile_supported([(8, 8), (5, 4), (2, 2), (1, 1)])
    assert not enc_tile_supported([(8, 8), (4, 4), (2, 2)])
    levels = [(16, 20), (8, 10), (4, 5), (2, 3)]
    S = sum(h * w for h, w in levels)
    value = _t(seeded_array('tile.nf.value', (1, S, 8, 32))).cuda()
    proj = _t(seeded_array('tile.nf.proj', (S, 384)))
    proj[::5, :256] = float('nan')
    proj[1::5, 3] = float('inf')
    proj[2::5, 300] = float('inf')
    proj[3::5, :256] = 1e30
    ref = torch.rand(1, S, 4, 2)
    ref[0, ::11] = float('nan')
    with pytest.raises(RuntimeError):
        deform_attn_enc_tile(value, proj.cuda(), ref.cuda(), levels_hw=[(16, 20), (9, 10), (4, 5), (2, 3)])
    out = deform_attn_enc_tile(value, proj.cuda(), ref.cuda(), levels_hw=levels)
    torch.cuda.synchronize()
    assert out.shape == (S, 256)
    clean = torch.arange(S) % 5 == 4
    clean &= torch.arange(S) % 11 != 0
    assert torch.isfinite(out.cpu()[clean]).all()


@pytest.mark.parametrize('D', [4, 30, 32, 64, 71, 1025])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_backward_vs_autograd_of_reference_formulation(D, dtype):
    """ms_deform_attn_backward against autograd through the reference's own PyTorch formulation
    (MO:92-149, restated in the oracle) -- the comparison mmcv's gradcheck makes for
    D in {4, 30, 32, 64, 71, 1025} (test_ms_deformable_attn.py:138-182)."""
    from pavenet_amd.ops import MultiScaleDeformableAttnFunction
    N, M, Lq, L, P = 2, 3, 5, 2, 2
    shapes = torch.as_tensor([(3, 2), (2, 1)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(D)
    value = (torch.rand(N, S, M, D, generator=g) * 0.01).to(dtype)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g).to(dtype) * 1.2 - 0.1  # some outside
    aw = torch.rand(N, Lq, M, L, P, generator=g).to(dtype) + 1e-5
    aw = aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    gout = torch.rand(N, Lq, M * D, generator=g).to(dtype)
    ref_in = [t.double().clone().requires_grad_(True) for t in (value, loc, aw)]
    R.msda_forward_torch(ref_in[0], shapes, ref_in[1], ref_in[2]).backward(gout.double())
    dev_in = [t.cuda().clone().requires_grad_(True) for t in (value, loc, aw)]
    out = MultiScaleDeformableAttnFunction.apply(dev_in[0], shapes.cuda(), lsi.cuda(), dev_in[1],
                                                 dev_in[2], 2)
    out.backward(gout.cuda())
    tol = dict(rtol=1e-9, atol=1e-12) if dtype == torch.float64 else dict(rtol=2e-4, atol=1e-6)
    for a, b in zip(dev_in, ref_in):
        np.testing.assert_allclose(a.grad.double().cpu().numpy(), b.grad.numpy(), **tol)


@pytest.mark.parametrize('kind,T', [('grid', 1), ('grid', 3), ('grid_clips', 2), ('pose', 3),
                                    ('pose', 1), ('tile', 1)])
def test_fused_backward_vs_autograd_of_reference_formulation(kind, T):
    """Gradients of the fused T-frame launches (forward = the fused HIP kernel, backward =
    pavenet_amd/fused_autograd.py on the col2im kernel) against fp64 autograd through the oracle's
    un-fused formulation (per-frame softmax, Z_t re-weighting, grid_sample sampler:
    tests/fused_expected.py, MO:1484-1578 / OT:1737-1858)."""
    from pavenet_amd import ops
    levels = [(8, 12), (4, 6), (2, 3), (1, 2)]
    shapes, lsi, sd, ld = _levels(levels)
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(T + len(kind))
    old_sampler, R.SAMPLER = R.SAMPLER, 'torch'      # differentiable CPU sampler (MO:92-149)
    try:
        if kind == 'pose':
            clips, Q, K = 2, 5, 15
            value = torch.randn(clips * T, S, 8, 32, generator=g)
            proj = torch.randn(clips * Q, T * 8 * 4 * K * 3, generator=g)
            ref = torch.sigmoid(torch.randn(clips, T * Q, 4, 2 * K, generator=g))
            exp_in = [t.double().requires_grad_(True) for t in (value, proj, ref)]
            exp = pose_expected(exp_in[0], shapes, lsi, exp_in[1], exp_in[2], T, clips, Q, K)
            dev_in = [t.cuda().requires_grad_(True) for t in (value, proj, ref)]
            out = ops.deform_attn_pose_fused(dev_in[0], sd, ld, dev_in[1], dev_in[2], T=T,
                                             n_clips=clips, num_query=Q, num_keypoints=K)
        else:
            clips = 2 if kind != 'grid' else 1
            U = (S if kind == 'tile' else 21) * clips
            value = torch.randn(clips * T, S, 8, 32, generator=g)
            proj = torch.randn(U, T * 8 * 16 * 3, generator=g)
            proj[:, :T * 256] *= 1.5
            if kind == 'tile':      # encoder self-attention: the reference grid of every token
                ys = torch.cat([((torch.arange(h * w) // w).float() + 0.5) / h for h, w in levels])
                xs = torch.cat([((torch.arange(h * w) % w).float() + 0.5) / w for h, w in levels])
                ref = torch.stack([xs, ys], -1)[None, :, None, :].expand(clips, -1, 4, 2)
                ref = (ref.reshape(1, U, 4, 2) + 0.01 * torch.randn(1, U, 4, 2, generator=g)).contiguous()
                unit_clip = torch.arange(clips).repeat_interleave(S)
            else:
                ref = torch.rand(T, U, 4, 2, generator=g) * 0.9 + 0.05
                unit_clip = torch.arange(U) % clips if kind == 'grid_clips' else torch.zeros(U).long()
            exp_in = [t.double().requires_grad_(True) for t in (value, proj, ref)]
            exp = grid_expected(exp_in[0], shapes, lsi, exp_in[1], exp_in[2], T, unit_clip.long())
            dev_in = [t.cuda().requires_grad_(True) for t in (value, proj, ref)]
            if kind == 'tile':
                out = ops.deform_attn_enc_tile(dev_in[0], dev_in[1], dev_in[2], levels_hw=levels)
            else:
                out = ops.deform_attn_grid_fused(
                    dev_in[0], sd, ld, dev_in[1], dev_in[2], T=T, n_clips=clips,
                    units_per_clip=U // clips,
                    unit_clip=unit_clip.to(torch.int32).cuda() if kind == 'grid_clips' else None)
        gout = torch.randn(exp.shape, generator=g)
        exp.backward(gout.double())
        out.backward(gout.cuda())
    finally:
        R.SAMPLER = old_sampler
    np.testing.assert_allclose(out.detach().cpu().numpy(), exp.detach().numpy(), rtol=1e-4, atol=1e-4)
    for a, b, nm in zip(dev_in, exp_in, ('value', 'proj', 'ref')):
        scale = float(b.grad.abs().max())
        np.testing.assert_allclose(a.grad.cpu().double().numpy(), b.grad.numpy(), rtol=2e-3,
                                   atol=2e-5 * max(scale, 1.0), err_msg=nm)


@pytest.mark.parametrize('dtype', [torch.uint8, torch.float32])
@pytest.mark.parametrize('hw,div', [((270, 480), 1), ((97, 61), 32), ((120, 160), 32)])
def test_preprocess_clip_vs_oracle_resize_kernel_unpinned_no_cv2(dtype, hw, div):
    """Device input pipeline (resize keep-ratio, BGR->RGB, normalise, pad, stack) against the
    NumPy restatement of the reference's test pipeline.  PARITY UNPINNED for the bilinear resize
    itself: the reference calls cv2.resize and cv2 is not in this image; sizes, scale factors and
    pad shapes ARE pinned against the reference's own mmcv.rescale_size
    (tests/test_host_cpu.py::test_pipeline_shapes_vs_the_references_own_rescale_size)."""
    from oracle import preprocess_ref as PR
    from pavenet_amd.preprocess import preprocess_clip
    rng = np.random.default_rng(hw[0] + div)
    frames = rng.integers(0, 256, size=(3, hw[0], hw[1], 3)).astype(np.uint8)
    src = _t(frames) if dtype == torch.uint8 else _t(frames.astype(np.float32))
    scale = (200, 120)
    out, meta = preprocess_clip(src.cuda(), img_scale=scale, size_divisor=div)
    exp, emeta = PR.preprocess_clip(frames, img_scale=scale, size_divisor=div)
    assert tuple(out.shape) == exp.shape and meta['img_shape'] == emeta['img_shape']
    assert meta['batch_input_shape'] == emeta['batch_input_shape']
    np.testing.assert_allclose(meta['scale_factor'], emeta['scale_factor'])
    # both sides follow OpenCV's published order operation by operation (double coordinates, float
    # fractions, horizontal then vertical pass, no fused multiply-add): bit-exact
    np.testing.assert_array_equal(out.cpu().numpy(), exp)


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride', [(2, 13, 17, 64, 64, 1), (1, 20, 9, 128, 128, 1),
                                                   (3, 14, 22, 128, 256, 2), (1, 7, 5, 32, 192, 1),
                                                   (2, 9, 9, 256, 64, 2)])
def test_conv3x3_mfma_vs_torch(N, H, W, Cin, Cout, stride):
    """Implicit-GEMM 3x3 convolution on the exact-fp32 MFMA (+bias, +ReLU) vs F.conv2d on the CPU."""
    from pavenet_amd.ops import conv3x3_nhwc
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin**0.5)
    b = torch.randn(Cout, generator=g)
    for relu in (False, True):
        exp = torch.nn.functional.conv2d(x, w, b, stride, 1)
        exp = torch.relu(exp) if relu else exp
        xd = x.cuda().contiguous(memory_format=torch.channels_last)
        wt = w.permute(2, 3, 1, 0).contiguous().cuda()  # [3, 3, Cin, Cout]
        out = conv3x3_nhwc(xd, wt, b.cuda(), stride=stride, relu=relu)
        assert out.shape == exp.shape
        np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('M,K,N', [(1000, 64, 256), (129, 128, 512), (5, 32, 64), (4096, 256, 192)])
def test_rows_gemm_bias_res_act_vs_torch(M, K, N):
    """Bottleneck tail as one kernel: relu(a @ w + bias + identity), incl. in-place identity."""
    from pavenet_amd.ops import rows_gemm_bias_res_act
    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / K**0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
    exp = a.double() @ w.double() + b.double()
    out = rows_gemm_bias_res_act(ad, wd, bd)
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)
    out = rows_gemm_bias_res_act(ad, wd, None, r.cuda(), relu=True)
    np.testing.assert_allclose(out.cpu().numpy(),
                               torch.relu(a.double() @ w.double() + r.double()).numpy(),
                               rtol=1e-4, atol=1e-4)
    idt = r.cuda()
    out = rows_gemm_bias_res_act(ad, wd, bd, idt, relu=True, out=idt)  # accumulate into identity
    assert out.data_ptr() == idt.data_ptr()
    np.testing.assert_allclose(out.cpu().numpy(), torch.relu(exp + r.double()).numpy(),
                               rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('N,H,W', [(2, 37, 53, ), (1, 64, 96), (3, 16, 8), (1, 21, 130), (2, 33, 388),
                                   (1, 18, 200)])
def test_stem_conv7x7_split_vs_torch_fp64(N, H, W):
    """pave_conv7x7s2_nchw_split_f32 (7x7 / stride 2 / pad 3 stem read from the NCHW batch) against
    torch's convolution in fp64, incl. odd sizes and the edge columns.  W % 4 == 0 takes the
    LDS-window kernel (several 96-pixel segments per row at W = 388, a ragged last one), the other
    widths the per-lane window-load kernel; both must also agree bit for bit."""
    from pavenet_amd.ops import conv7x7s2_nchw_split, split_stem7x7_weight
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    exp = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 2, 3)
    wp = split_stem7x7_weight(w.cuda())
    y = conv7x7s2_nchw_split(x.cuda(), wp, b.cuda())
    assert tuple(y.shape) == tuple(exp.shape) and y.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(y.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=2e-5)
    y = conv7x7s2_nchw_split(x.cuda(), wp, None, relu=True)
    exp = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), None, 2, 3))
    np.testing.assert_allclose(y.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=2e-5)
    xn = x.clone()
    xn[0, 1, 3, 4] = float('nan')                     # a NaN pixel poisons only its own windows
    y = conv7x7s2_nchw_split(xn.cuda(), wp, None)
    bad = torch.isnan(y[0]).any(0).cpu()
    expn = torch.isnan(torch.nn.functional.conv2d(xn, w, None, 2, 3)[0]).any(0)
    assert torch.equal(bad, expn)
    if W % 4 == 0:
        from pavenet_amd import native
        new = conv7x7s2_nchw_split(x.cuda(), wp, b.cuda(), relu=True).clone()
        with native.diag_build(9):
            old = conv7x7s2_nchw_split(x.cuda(), wp, b.cuda(), relu=True).clone()
        assert torch.equal(new, old)


@pytest.mark.parametrize('N,H,W', [(2, 37, 53), (1, 21, 131), (2, 33, 389), (1, 75, 1333), (1, 18, 202)])
def test_stem_repitched_odd_width_equals_window_load_kernel(N, H, W):
    """Widths off the 4-pixel grid (the reference's PoseTrack test canvas is 750 x 1333,
    configs/_base_/datasets/posetrack17_video_keypoint.py:68-81 with size_divisor = 1): the image rows re-laid at a
    16-byte aligned pitch with zero pad columns (pave_repitch_rows_f32) through the LDS-window stem kernel
    (row_pitch of pave_conv7x7s2_nchw_split_f32) -- bit-identical to the per-lane window-load kernel on the dense
    image (same products in the same order), within fp32 rounding of torch's fp64 convolution, and a NaN in a pad
    column would show (the pad must be zeros, not stale memory: the buffer is poisoned first)."""
    from pavenet_amd.ops import conv7x7s2_nchw_split, repitch_rows, split_stem7x7_weight
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    wp = split_stem7x7_weight(w.cuda())
    xd = x.cuda()
    pitch = (W + 3) // 4 * 4
    poison = torch.full((N, 3, H, pitch), float('nan'), device='cuda')    # the allocator hands this block back
    del poison
    xp = repitch_rows(xd)
    assert tuple(xp.shape) == (N, 3, H, pitch) and xp.data_ptr() % 16 == 0
    assert torch.equal(xp[..., :W], xd) and bool((xp[..., W:] == 0).all())
    got = conv7x7s2_nchw_split(xp, wp, b.cuda(), relu=True, valid_w=W)
    old = conv7x7s2_nchw_split(xd, wp, b.cuda(), relu=True)            # dense odd width: the window-load kernel
    assert tuple(got.shape) == tuple(old.shape) == (N, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1)
    assert torch.equal(got, old)
    exp = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 2, 3))
    np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=2e-5)
    # fp16 operands: only the LDS-window kernel exists, so an odd width needs the re-laid rows
    wh = split_stem7x7_weight(w.cuda(), 16)
    goth = conv7x7s2_nchw_split(xp, wh, b.cuda(), relu=True, valid_w=W)
    exph = torch.relu(torch.nn.functional.conv2d(x.half().double(), w.half().double(), b.double(), 2, 3))
    np.testing.assert_allclose(goth.cpu().numpy(), exph.numpy(), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):
        conv7x7s2_nchw_split(xp, wp, None, valid_w=pitch + 1)


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride', [(2, 13, 17, 256, 512, 2), (1, 20, 9, 64, 128, 2),
                                                   (3, 8, 8, 128, 256, 1)])
def test_conv1x1_strided_split_vs_torch_fp64(N, H, W, Cin, Cout, stride):
    from pavenet_amd.ops import conv1x1_strided_split, split_weight_bf16x3
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, generator=g) / Cin**0.5
    b = torch.randn(Cout, generator=g)
    exp = torch.nn.functional.conv2d(x.double(), w.double()[:, :, None, None], b.double(), stride)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    y = conv1x1_strided_split(xd, split_weight_bf16x3(w.cuda()), b.cuda(), stride=stride)
    assert tuple(y.shape) == tuple(exp.shape)
    np.testing.assert_allclose(y.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)


def test_ref_update_vs_torch_formulation():
    from pavenet_amd.bricks import inverse_sigmoid
    from pavenet_amd.ops import ref_update
    g = torch.Generator().manual_seed(3)
    ref = torch.rand(4, 300, 30, generator=g) * 1.2 - 0.1       # some outside [0, 1]
    ref[0, 0, :4] = torch.tensor([0.0, 1.0, 1e-7, 1 - 1e-7])
    tmp = torch.randn(4, 300, 30, generator=g) * 3
    exp = (tmp.double() + inverse_sigmoid(ref.double())).sigmoid()
    got = ref_update(tmp.cuda(), ref.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=2e-6, atol=1e-7)
    # a 30-column slice of a 32-column matrix (a branch output as its GEMM leaves it) is read in place
    wide = torch.randn(4, 300, 32, generator=g).cuda()
    wide[..., :30] = tmp.cuda()
    assert torch.equal(ref_update(wide[..., :30], ref.cuda()), got)


@pytest.mark.parametrize('N,HW,C,G', [(3, 1000, 256, 32), (2, 35, 256, 32), (1, 7, 64, 8),
                                      (2, 4100, 128, 32)])
def test_groupnorm_nhwc_into_vs_torch_fp64(N, HW, C, G):
    """pave_groupnorm_nhwc_f32 against nn.functional.group_norm in fp64, writing into a slice of
    a larger token buffer (the rest of the buffer stays untouched); bit-reproducible."""
    from pavenet_amd.ops import groupnorm_nhwc_into
    g = torch.Generator().manual_seed(HW + C)
    x = torch.randn(N, HW, C, generator=g) * 3 + 1.5         # non-zero mean
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    exp = torch.nn.functional.group_norm(x.double().permute(0, 2, 1), G, gam.double(), bet.double(),
                                         1e-5).permute(0, 2, 1)
    buf = torch.full((N, HW + 13, C), 7.0).cuda()
    dst = buf[:, 5:5 + HW]
    groupnorm_nhwc_into(x.cuda(), gam.cuda(), bet.cuda(), G, 1e-5, dst)
    np.testing.assert_allclose(dst.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    assert float(buf[:, :5].min()) == 7.0 and float(buf[:, 5 + HW:].max()) == 7.0
    buf2 = torch.full((N, HW + 13, C), 7.0).cuda()
    groupnorm_nhwc_into(x.cuda(), gam.cuda(), bet.cuda(), G, 1e-5, buf2[:, 5:5 + HW])
    assert torch.equal(buf, buf2)


@pytest.mark.parametrize('N,hws,C,G', [(3, (1000, 260, 63, 20), 256, 32), (1, (7,), 64, 8),
                                       (2, (4100, 35, 9000, 1, 130, 17), 128, 32)])
def test_groupnorm_levels_equal_the_one_level_entry(N, hws, C, G):
    """pave_groupnorm_levels_nhwc_f32 (every level of the neck in the same three launches; more than four maps
    go in calls of four) against pave_groupnorm_nhwc_f32 level by level: bit-equal, written into slices of ONE
    token buffer whose other rows stay untouched; and against fp64 torch."""
    from pavenet_amd.ops import groupnorm_levels_into, groupnorm_nhwc_into
    g = torch.Generator().manual_seed(sum(hws) + C)
    S = sum(hws) + 11
    buf = torch.full((N, S, C), 7.0).cuda()
    ref = torch.full((N, S, C), 7.0).cuda()
    levels, st = [], 4
    for i, HW in enumerate(hws):
        x = (torch.randn(N, HW, C, generator=g) * (1 + i) + 0.5 * i).cuda()
        gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
        eps = 1e-5 * (1 + i)
        levels.append((x, gam, bet, eps, buf[:, st:st + HW]))
        groupnorm_nhwc_into(x, gam, bet, G, eps, ref[:, st:st + HW])
        exp = torch.nn.functional.group_norm(x.double().permute(0, 2, 1), G, gam.double(), bet.double(),
                                             eps).permute(0, 2, 1)
        np.testing.assert_allclose(ref[:, st:st + HW].cpu().numpy(), exp.cpu().numpy(), rtol=1e-5, atol=2e-5)
        st += HW
    groupnorm_levels_into(levels, G)
    assert torch.equal(buf, ref)
    assert float(buf[:, :4].min()) == 7.0 and float(buf[:, st:].max()) == 7.0


@pytest.mark.parametrize('N,H,W,C', [(2, 16, 24, 64), (1, 15, 9, 64), (3, 7, 8, 32)])
def test_bias_relu_maxpool_vs_torch(N, H, W, C):
    """Stem tail: maxpool3x3/s2/p1(relu(x + b)) in one pass == torch's two ops (bit-exact)."""
    from pavenet_amd.ops import bias_relu_maxpool_nhwc
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(N, C, H, W, generator=g)
    b = torch.randn(C, generator=g)
    exp = torch.nn.functional.max_pool2d(torch.relu(x + b[None, :, None, None]), 3, 2, 1)
    out = bias_relu_maxpool_nhwc(x.cuda().contiguous(memory_format=torch.channels_last), b.cuda())
    assert out.shape == exp.shape
    assert torch.equal(out.cpu(), exp)


def test_rows_gemm_prologue_and_second_source():
    """Full Bottleneck tail: relu([relu(a + ab) | a2] @ w + bias) — A-side bias/ReLU on load and
    the downsample branch as a second K range."""
    from pavenet_amd.ops import rows_gemm_bias_res_act
    g = torch.Generator().manual_seed(11)
    M, K, K2, N = 777, 64, 96, 256
    a, a2 = torch.randn(M, K, generator=g), torch.randn(M, K2, generator=g)
    ab, b = torch.randn(K, generator=g), torch.randn(N, generator=g)
    w = torch.randn(K + K2, N, generator=g) / (K + K2)**0.5
    A = torch.cat([torch.relu(a + ab), a2], 1).double()
    exp = torch.relu(A @ w.double() + b.double())
    out = rows_gemm_bias_res_act(a.cuda(), w.cuda(), b.cuda(), None, relu=True, a_bias=ab.cuda(),
                                 a2=a2.cuda())
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)
    r = torch.randn(M, N, generator=g)
    out = rows_gemm_bias_res_act(a.cuda(), w[:K].contiguous().cuda(), b.cuda(), r.cuda(),
                                 relu=False, a_bias=ab.cuda())
    exp = torch.relu(a + ab).double() @ w[:K].double() + b.double() + r.double()
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):
        rows_gemm_bias_res_act(a.cuda(), w.cuda(), b.cuda())          # w rows != K


def test_split_bf16x3_is_exact():
    from pavenet_amd.ops import split_bf16x3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4097, generator=g) * torch.logspace(-20, 20, 4097)
    x[:3] = torch.tensor([0.0, -0.0, 1.0])
    p = split_bf16x3(x.cuda()).cpu()
    parts = (p.to(torch.int32) << 16).view(torch.float32)   # bf16 bits -> fp32
    assert torch.equal(parts[0].double() + parts[1].double() + parts[2].double(), x.double())


@pytest.mark.parametrize('M,K,N', [(1000, 256, 1024), (257, 1024, 256), (5, 64, 128), (3000, 128, 256),
                                   (1300, 256, 64)])
def test_gemm_bf16x3_accuracy_vs_fp64(M, K, N):
    """The split-bf16 GEMM is as accurate as an fp32 GEMM: its error against fp64 is within 2.5x of
    torch's fp32 matmul error on the same data (and far below bf16 / tf32 levels)."""
    from pavenet_amd.ops import gemm_bf16x3, split_weight_bf16x3
    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K**0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ab = torch.randn(K, generator=g)
    ad, wd = a.cuda(), w.cuda()
    wp = split_weight_bf16x3(wd)
    exact = a.double() @ w.double().t()
    got = gemm_bf16x3(ad, wp).cpu().double()
    ref32 = (ad @ wd.t()).cpu().double()
    err, err32 = (got - exact).abs().max().item(), (ref32 - exact).abs().max().item()
    scale = exact.abs().max().item()
    print(f'bf16x3 GEMM M={M} K={K} N={N}: max err {err:.3e} (torch fp32 {err32:.3e}, scale {scale:.2f})')
    assert err <= max(2.5 * err32, 2e-7 * scale), (err, err32, scale)
    assert err < 1e-5 * scale            # bf16 would be ~4e-3, tf32 ~5e-4
    # epilogue / prologue variants
    out = gemm_bf16x3(ad, wp, b.cuda(), r.cuda(), relu=True)
    np.testing.assert_allclose(out.cpu().numpy(), torch.relu(exact + b.double() + r.double()).numpy(),
                               rtol=1e-5, atol=1e-5)
    idt = r.cuda()
    out = gemm_bf16x3(ad, wp, b.cuda(), idt, relu=False, out=idt, a_bias=ab.cuda())
    exp = torch.relu(a + ab).double() @ w.double().t() + b.double() + r.double()
    assert out.data_ptr() == idt.data_ptr()
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('M,K1,K2,N', [(1000, 64, 64, 256), (333, 128, 64, 128), (129, 48, 80, 64)])
def test_gemm_bf16x3_cat_two_sources_vs_fp64(M, K1, K2, N):
    """pave_gemm_bf16x3_cat_f32: [a | a2] @ W^T + bias (+ residual, ReLU) with the two row matrices
    read in place (the Bottleneck tail conv3 + stride-1 downsample, resnet.py:264-283) -- vs fp64."""
    from pavenet_amd.ops import gemm_bf16x3_cat, split_weight_bf16x3
    g = torch.Generator().manual_seed(M + K1 + N)
    a, a2 = torch.randn(M, K1, generator=g), torch.randn(M, K2, generator=g)
    w = torch.randn(N, K1 + K2, generator=g) / (K1 + K2)**0.5
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    wp = split_weight_bf16x3(w.cuda()) if (K1 + K2) % 64 == 0 and N % 64 == 0 else None
    if wp is None:
        pytest.skip('weight layout helper needs K % 64 == 0')
    exact = torch.cat([a, a2], 1).double() @ w.double().t() + b.double()
    out = gemm_bf16x3_cat(a.cuda(), a2.cuda(), wp, b.cuda(), None, relu=False)
    np.testing.assert_allclose(out.cpu().numpy(), exact.numpy(), rtol=1e-5, atol=1e-5)
    out = gemm_bf16x3_cat(a.cuda(), a2.cuda(), wp, b.cuda(), r.cuda(), relu=True)
    np.testing.assert_allclose(out.cpu().numpy(), torch.relu(exact + r.double()).numpy(),
                               rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride', [(2, 13, 17, 48, 48, 1), (1, 11, 9, 96, 48, 1),
                                                   (2, 14, 10, 48, 96, 2), (1, 9, 12, 96, 192, 2)])
def test_conv3x3_split_padded_channels_with_residual_vs_fp64(N, H, W, Cin, Cout, stride):
    """HRNet's 48- / 96-channel 3x3 convolutions (hrnet.py:183-260: BasicBlock conv + BN + identity
    + ReLU; fuse-layer stride-2 convolutions) on the exact 3-plane kernel: weight planes zero-padded
    to Cout % 64 == 0 and 9 Cin % 32 == 0, the real Cout columns stored, identity and ReLU in the
    epilogue -- against fp64."""
    from pavenet_amd.ops import conv3x3_split, split_conv3x3_weight
    g = torch.Generator().manual_seed(Cin * Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin**0.5)
    b = torch.randn(Cout, generator=g)
    exp = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride, 1)
    r = torch.randn(exp.shape, generator=g)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wp = split_conv3x3_weight(w.cuda())
    y = conv3x3_split(xd, wp, b.cuda(), stride=stride, relu=False, cout=Cout)
    assert tuple(y.shape) == tuple(exp.shape)
    np.testing.assert_allclose(y.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=2e-5)
    rd = r.cuda().contiguous(memory_format=torch.channels_last)
    y = conv3x3_split(xd, wp, b.cuda(), stride=stride, relu=True, residual=rd, cout=Cout)
    np.testing.assert_allclose(y.cpu().numpy(), torch.relu(exp + r.double()).numpy(), rtol=1e-5, atol=2e-5)
    xn = x.clone()
    xn[0, 3, 2, 2] = float('nan')     # the zero slab that pads K must not spread a NaN pixel
    y = conv3x3_split(xn.cuda().contiguous(memory_format=torch.channels_last), wp, None, stride=stride,
                      cout=Cout)
    bad = torch.isnan(y[0]).any(0).cpu()
    expn = torch.isnan(torch.nn.functional.conv2d(xn, w, None, stride, 1)[0]).any(0)
    assert torch.equal(bad, expn)


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride', [(2, 25, 42, 2048, 256, 2), (1, 13, 21, 1024, 128, 1),
                                                   (3, 9, 11, 944, 100, 1), (1, 14, 14, 1040, 64, 2),
                                                   (2, 25, 42, 384, 256, 2)])
def test_conv3x3_split_k_form_vs_fp64(N, H, W, Cin, Cout, stride):
    """The split-K form of the 3-plane 3x3 convolution (few output pixels, long K: the
    ChannelMapper's extra level, channel_mapper.py:84-97): the plan must exist for these shapes,
    the parts are summed in a fixed order (two runs are bit-identical), and the result -- bias,
    residual, ReLU applied by the second launch -- meets the same fp64 bound as the one-pass kernel,
    to which it is compared too (diag variant 6 = no split-K plan)."""
    from pavenet_amd import native
    from pavenet_amd.ops import conv3x3_split, split_conv3x3_weight
    lib = native.load()
    M = N * ((H - 1) // stride + 1) * ((W - 1) // stride + 1)
    # (a fixed number of parts per K range and tile-count class -- from K = 8 192 on: 8, or ~32 below 16 row tiles (a
    # one-clip batch); 4 for 2 048 <= K < 8 192 (the neck's extra level behind HRNet-w48: K = 3 456) --: the
    # summation order does not depend on the batch size inside a class)
    ws = lib.pave_conv3x3_splitk_workspace_bytes(N, H, W, Cin, Cout, stride)
    assert ws > 0 and ws % (M * Cout * 4) == 0
    parts = ws // (M * Cout * 4)
    tiles = (M + 127) // 128 * ((Cout + 255) // 256 if Cout % 256 == 0 else (Cout + 127) // 128)
    assert parts == 4 if 9 * Cin < 8192 else (parts == 8 if tiles >= 16 else 24 <= parts <= 32)
    ws4 = lib.pave_conv3x3_splitk_workspace_bytes(4 * N, H, W, Cin, Cout, stride)
    assert ws4 == 0 or (ws4 % (4 * M * Cout * 4) == 0 and ws4 // (4 * M * Cout * 4) in (4, 8, parts))
    assert lib.pave_conv3x3_splitk_workspace_bytes(N, H, W, 256, Cout, stride) == 4 * M * Cout * 4   # K = 2 304
    assert lib.pave_conv3x3_splitk_workspace_bytes(N, H, W, 128, Cout, stride) == 0    # K = 1 152: one pass
    g = torch.Generator().manual_seed(Cin * Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin**0.5)
    b = torch.randn(Cout, generator=g)
    exp = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride, 1)
    r = torch.randn(exp.shape, generator=g)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    rd = r.cuda().contiguous(memory_format=torch.channels_last)
    wp = split_conv3x3_weight(w.cuda())
    y = conv3x3_split(xd, wp, b.cuda(), stride=stride, relu=False, cout=Cout)
    assert tuple(y.shape) == tuple(exp.shape)
    np.testing.assert_allclose(y.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=2e-5)
    assert torch.equal(y, conv3x3_split(xd, wp, b.cuda(), stride=stride, relu=False, cout=Cout))
    y2 = conv3x3_split(xd, wp, b.cuda(), stride=stride, relu=True, residual=rd, cout=Cout)
    np.testing.assert_allclose(y2.cpu().numpy(), torch.relu(exp + r.double()).numpy(), rtol=1e-5, atol=2e-5)
    with native.diag_build(6) as dlib:
        assert dlib.pave_conv3x3_splitk_workspace_bytes(N, H, W, Cin, Cout, stride) == 0
        one = conv3x3_split(xd, wp, b.cuda(), stride=stride, relu=False, cout=Cout)
    np.testing.assert_allclose(y.cpu().numpy(), one.cpu().numpy(), rtol=1e-5, atol=2e-5)
    # a NaN pixel poisons exactly the outputs whose window holds it, in every part
    xn = x.clone()
    xn[0, Cin - 1, 2, 2] = float('nan')
    yn = conv3x3_split(xn.cuda().contiguous(memory_format=torch.channels_last), wp, None, stride=stride,
                       cout=Cout)
    expn = torch.isnan(torch.nn.functional.conv2d(xn, w, None, stride, 1)[0]).any(0)
    assert torch.equal(torch.isnan(yn[0]).any(0).cpu(), expn)


@pytest.mark.parametrize('M,K,N', [(3150, 2048, 512), (3150, 2048, 256), (1050, 4096, 300), (7350, 2048, 512)])
def test_plain_gemm_split_k_form_vs_fp64_and_one_pass(M, K, N):
    """The same split-K plan for the plain row GEMM (pave_gemm_bf16x3_splitk_f32: a one-clip batch's layer4 1x1
    reductions, the neck's C5 lateral): ops.gemm_bf16x3 takes it where the plan exists; bias / residual (in place
    too) / ReLU are applied by the reduction launch; fp64 bound, bit-reproducible, equal to the one-pass kernel up
    to the summation order; shapes without a plan (short K, many tiles) keep the one-pass entry."""
    from pavenet_amd import native, ops
    lib = native.load()
    ws = lib.pave_gemm_splitk_workspace_bytes(M, K, N)
    assert ws > 0 and ws % (M * N * 4) == 0 and ws // (M * N * 4) == 4
    assert lib.pave_gemm_splitk_workspace_bytes(M, 1024, N) == 0 and lib.pave_gemm_splitk_workspace_bytes(40000, K, N) == 0
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).relu_()
    w = torch.randn(N, K, generator=g) / K**0.5
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    wp = ops.split_weight_bf16x3(w.cuda(), pad=N % 64 != 0)
    kw = dict(n_out=N) if N % 64 else {}
    exp = a.double() @ w.double().t() + b.double()
    y = ops.gemm_bf16x3(a.cuda(), wp, b.cuda(), relu=True, **kw)
    np.testing.assert_allclose(y.cpu().numpy(), exp.clamp(min=0).numpy(), rtol=1e-5, atol=2e-5)
    assert torch.equal(y, ops.gemm_bf16x3(a.cuda(), wp, b.cuda(), relu=True, **kw))
    rd = r.cuda()
    y2 = ops.gemm_bf16x3(a.cuda(), wp, b.cuda(), rd, out=rd, **kw)
    assert y2.data_ptr() == rd.data_ptr()
    np.testing.assert_allclose(y2.cpu().numpy(), (exp + r.double()).numpy(), rtol=1e-5, atol=2e-5)
    with native.diag_build(6) as dlib:     # no split-K plan: the one-pass kernels
        assert dlib.pave_gemm_splitk_workspace_bytes(M, K, N) == 0
        one = ops.gemm_bf16x3(a.cuda(), wp, b.cuda(), relu=True, **kw)
    np.testing.assert_allclose(y.cpu().numpy(), one.cpu().numpy(), rtol=1e-5, atol=2e-5)
    an = a.clone()
    an[5, 7] = float('nan')                # a NaN poisons its row, in every part's sum
    yn = ops.gemm_bf16x3(an.cuda(), wp, None, **kw)
    assert torch.isnan(yn[5]).all() and not torch.isnan(yn[4]).any() and not torch.isnan(yn[6]).any()


@pytest.mark.parametrize('shifts', [(0,), (0, 1), (1, 0, 2), (0, 0, 1, 3), (2, 1, 0, 0)])
def test_fuse_sum_nhwc_equals_upsample_add_relu(shifts):
    """pave_fuse_sum_nhwc_f32 (HRNet fuse layer, hrnet.py:197-214): the sum over the branches in the
    reference's order, the coarser terms read through a nearest-neighbour up-sampling, + ReLU, in
    one pass -- bit-identical to F.interpolate + add + relu."""
    from pavenet_amd.ops import fuse_sum_nhwc
    g = torch.Generator().manual_seed(sum(shifts) + len(shifts))
    N, C, H, W = 2, 48, 16, 24
    terms = [(torch.randn(N, C, H >> s, W >> s, generator=g).cuda().contiguous(
        memory_format=torch.channels_last), s) for s in shifts]
    y = 0
    for t, s in terms:
        y = y + (torch.nn.functional.interpolate(t, scale_factor=2 ** s, mode='nearest') if s else t)
    exp = torch.relu(y)
    out = fuse_sum_nhwc(terms, relu=True)
    assert out.is_contiguous(memory_format=torch.channels_last) and tuple(out.shape) == (N, C, H, W)
    assert torch.equal(out, exp)
    assert torch.equal(fuse_sum_nhwc(terms, relu=False), y)
    with pytest.raises(RuntimeError):
        fuse_sum_nhwc([(terms[0][0], terms[0][1]), (terms[0][0][:, :, :-1], 0)])


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride', [(2, 19, 27, 64, 64, 1), (1, 33, 21, 48, 256, 2),
                                                   (3, 9, 11, 128, 128, 1), (2, 8, 8, 16, 36, 1)])
def test_conv3x3_buffer_addressed_form_equals_flat_form(N, H, W, Cin, Cout, stride):
    """The 3x3 form reads the map through a buffer resource (out-of-image taps = out-of-range lanes
    that read zeros); maps of 4 GiB and more take 64-bit lane addresses and a zero chunk (diag
    variant 5 forces that form).  Same operands, same order: bit-identical, borders, image
    boundaries inside a tile and a NaN pixel included."""
    from pavenet_amd import native
    from pavenet_amd.ops import conv3x3_split, split_conv3x3_weight
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    x[0, 1, 0, 0] = float('nan')
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wp = split_conv3x3_weight((torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).cuda())
    b = torch.randn(Cout, generator=g).cuda()
    with native.diag_build(5):
        flat = conv3x3_split(xd, wp, b, stride=stride, relu=False, cout=Cout).clone()
    with native.diag_build(17):     # (17: no half-tail form for 33 .. 48 outputs -- the flat form has none)
        buf = conv3x3_split(xd, wp, b, stride=stride, relu=False, cout=Cout).clone()
    torch.cuda.synchronize()
    assert torch.equal(torch.isnan(buf), torch.isnan(flat))
    assert torch.equal(torch.nan_to_num(buf), torch.nan_to_num(flat))
    assert int(torch.isnan(buf[0]).any(0).sum()) <= 4     # only the windows holding the NaN pixel


def test_conv3x3_buffer_addressed_form_between_2_and_4_gib():
    """A 2.2 GB map: lane byte offsets above 2^31 and the out-of-range sentinel of a padded tap must
    both stay correct (a sentinel inside the map would read data instead of zeros) -- against the
    64-bit lane-address form (diag variant 5), bit for bit, borders included."""
    from pavenet_amd import native
    from pavenet_amd.ops import conv3x3_split, split_conv3x3_weight
    N, H, W, Cin, Cout = 1, 2944, 2944, 64, 64
    assert 2 ** 31 < N * H * W * Cin * 4 < 2 ** 32 - 65536
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(N, H, W, Cin, device='cuda', generator=g).permute(0, 3, 1, 2)   # channels_last map
    wp = split_conv3x3_weight(torch.randn(Cout, Cin, 3, 3, device='cuda', generator=g) * 0.05)
    with native.diag_build(5):
        flat = conv3x3_split(x, wp, None, stride=2, relu=False, cout=Cout).clone()
    buf = conv3x3_split(x, wp, None, stride=2, relu=False, cout=Cout)
    torch.cuda.synchronize()
    assert torch.equal(buf, flat)


@pytest.mark.parametrize('form', ['identity_next64', 'identity_inplace_next128', 'downsample_next64',
                                  'identity_last', 'tail_identity_inplace_next64', 'tail_downsample_next128',
                                  'tail_identity_last'])
def test_bottleneck_chain_equals_separate_launches(form):
    """pave_bottleneck_chain_f32 (3x3 -> conv3 + identity | downsample -> next conv1 of one
    128-pixel tile per workgroup, the intermediate rows re-read through L2) must give exactly what
    the three entry points give launched one after the other, ragged last tile included, and the
    in-place form (out aliases the identity) too."""
    from pavenet_amd import native, ops
    g = torch.Generator().manual_seed(len(form))
    dev = 'cuda'
    N, H, W = 2, 19, 27

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)

    def cl(t):
        return t.contiguous(memory_format=torch.channels_last)

    c1 = cl(torch.relu(rnd(N, 64, H, W)))
    w2p, b2 = ops.split_conv3x3_weight(rnd(64, 64, 3, 3, scale=0.05)), rnd(64, scale=0.1)
    down = 'downsample' in form
    k2 = 64 if down else 0
    w3p, b3 = ops.split_weight_bf16x3(rnd(256, 64 + k2, scale=0.08)), rnd(256, scale=0.1)
    cn = 0 if form.endswith('last') else (128 if form.endswith('128') else 64)
    w1np = ops.split_weight_bf16x3(rnd(cn, 256, scale=0.05)) if cn else None
    b1n = rnd(cn, scale=0.1) if cn else None
    x_in = cl(rnd(N, 64 if down else 256, H, W))

    # separate launches (on the tile kernels -- diag variant 19: at this test's few hundred rows the shipped
    # selection would take the K-split small-row form, which equals them to rounding only)
    with native.diag_build(19):
        c2 = ops.conv3x3_split(c1, w2p, b2, stride=1, relu=True)
        c2rows = c2.permute(0, 2, 3, 1).reshape(-1, 64)
        xrows = x_in.permute(0, 2, 3, 1).reshape(-1, x_in.shape[1])
        if down:
            exp_out = ops.gemm_bf16x3_cat(c2rows, xrows, w3p, b3, None, relu=True)
        else:
            exp_out = ops.gemm_bf16x3(c2rows, w3p, b3, xrows, relu=True)
        exp_c1n = ops.gemm_bf16x3(exp_out, w1np, b1n, None, relu=True) if cn else None

    inplace = 'inplace' in form
    res = x_in.clone(memory_format=torch.channels_last) if inplace else x_in
    if form.startswith('tail'):    # the chain from conv3 on, behind a separately run 3x3
        out, c1n = ops.bottleneck_chain(None, None, None, w3p, b3, residual=None if down else res,
                                        a2=x_in if down else None, w1n_planes=w1np, b1n=b1n,
                                        out=res if inplace else None, c2=c2)
    else:
        out, c1n = ops.bottleneck_chain(c1, w2p, b2, w3p, b3, residual=None if down else res,
                                        a2=x_in if down else None, w1n_planes=w1np, b1n=b1n,
                                        out=res if inplace else None)
    torch.cuda.synchronize()
    if inplace:
        assert out.data_ptr() == res.data_ptr()
    assert torch.equal(out.permute(0, 2, 3, 1).reshape(-1, 256), exp_out)
    if cn:
        assert torch.equal(c1n.permute(0, 2, 3, 1).reshape(-1, cn), exp_c1n)
    else:
        assert c1n is None
    if cn and not form.startswith('tail'):
        # the 256-row tile form of the launch (two row tiles per wave in the 3x3 and the 64-output conv1,
        # the wide bodies as two 128-row halves; diag variant 16 forces it at this size)
        from pavenet_amd import native
        res2 = x_in.clone(memory_format=torch.channels_last) if inplace else x_in
        with native.diag_build(16):
            out2, c1n2 = ops.bottleneck_chain(c1, w2p, b2, w3p, b3, residual=None if down else res2,
                                              a2=x_in if down else None, w1n_planes=w1np, b1n=b1n,
                                              out=res2 if inplace else None)
            torch.cuda.synchronize()
        assert torch.equal(out2.permute(0, 2, 3, 1).reshape(-1, 256), exp_out)
        assert torch.equal(c1n2.permute(0, 2, 3, 1).reshape(-1, cn), exp_c1n)
    with pytest.raises(RuntimeError):
        ops.bottleneck_chain(c1, w2p, b2, w3p, b3, residual=res, a2=x_in if down else cl(rnd(N, 64, H, W)))


@pytest.mark.parametrize('M,K,N', [(700, 96, 48), (333, 192, 96), (129, 384, 192), (1000, 64, 36)])
def test_gemm_bf16x3_padded_output_width_vs_fp64(M, K, N):
    """Row GEMM with N % 64 != 0 (HRNet fuse-layer 1x1 convolutions 96 -> 48 etc.): planes padded
    to roundup(N, 64) rows, N columns stored; bias / residual / ReLU have N columns."""
    from pavenet_amd.ops import gemm_bf16x3, split_weight_bf16x3
    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K**0.5
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    wp = split_weight_bf16x3(w.cuda(), pad=True)
    exact = a.double() @ w.double().t() + b.double()
    out = gemm_bf16x3(a.cuda(), wp, b.cuda(), None, n_out=N)
    assert tuple(out.shape) == (M, N)
    np.testing.assert_allclose(out.cpu().numpy(), exact.numpy(), rtol=1e-5, atol=1e-5)
    out = gemm_bf16x3(a.cuda(), wp, b.cuda(), r.cuda(), relu=True, n_out=N)
    np.testing.assert_allclose(out.cpu().numpy(), torch.relu(exact + r.double()).numpy(),
                               rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('M,K,G,gn', [(1200, 512, 7, 512), (300, 256, 3, 64), (77, 64, 5, 128)])
def test_gemm_bf16x3_grouped_vs_fp64(M, K, G, gn):
    """pave_gemm_bf16x3_grouped_f32: G per-frame Linears of one layer (OT:6728-6740 kpt_branches) in
    one launch -- group g multiplies its own column block of a by its own weight -- vs fp64."""
    from pavenet_amd.ops import gemm_bf16x3_grouped, split_weight_bf16x3
    g = torch.Generator().manual_seed(M + K + G)
    a = torch.randn(M, G * K, generator=g)
    w = torch.randn(G, gn, K, generator=g) / K**0.5
    b = torch.randn(G, gn, generator=g)
    wp = split_weight_bf16x3(w.flatten(0, 1).contiguous().cuda())
    out = gemm_bf16x3_grouped(a.cuda(), wp, b.flatten().cuda(), gn, relu=True)
    exp = torch.relu(torch.einsum('mgk,gnk->mgn', a.view(M, G, K).double(), w.double()) + b.double())
    np.testing.assert_allclose(out.cpu().numpy(), exp.reshape(M, G * gn).numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('form', ['rows', 'rows_n64', 'rows_abias_res', 'ex', 'ln', 'strided', 'conv3x3',
                                  'conv3x3_s2'])
def test_gemm_generations_are_bit_identical(form):
    """The LDS-DMA generation of the 3-plane split GEMM (pave_gemm_dma.hip, the default) issues the
    same six products in the same order per accumulator as the first-generation kernels
    (pave_gemm_split.hip): every form must give bit-identical results (LayerNorm epilogue: same
    GEMM, different summation order of the row statistics -> 2e-6), including ragged M."""
    from pavenet_amd import native, ops
    g = torch.Generator().manual_seed(len(form))
    dev = 'cuda'

    def run():
        if form in ('rows', 'rows_n64', 'rows_abias_res'):
            M, K, N = (1237, 192, 384) if form == 'rows' else ((515, 256, 64) if form == 'rows_n64'
                                                                else (777, 128, 256))
            a = torch.randn(M, K, generator=g).to(dev)
            wp = ops.split_weight_bf16x3((torch.randn(N, K, generator=g) * 0.05).to(dev))
            b = torch.randn(N, generator=g).to(dev)
            if form == 'rows_abias_res':
                r, ab = torch.randn(M, N, generator=g).to(dev), torch.randn(K, generator=g).to(dev)
                return lambda: ops.gemm_bf16x3(a, wp, b, r, relu=True, a_bias=ab)
            return lambda: ops.gemm_bf16x3(a, wp, b, None, relu=(form == 'rows'))
        if form == 'ex':
            M, K, N, rows, ns = 901, 256, 640, 53, 256
            a = torch.randn(M, K, generator=g).to(dev)
            wp = ops.split_weight_bf16x3((torch.randn(N, K, generator=g) * 0.05).to(dev))
            tab = torch.randn(rows, N, generator=g).to(dev)
            return lambda: torch.cat(ops.gemm_bf16x3_ex(a, wp, None, tab, residual_rows=rows, n_split=ns), 1)
        if form == 'ln':
            M, K = 645, 320
            a = torch.randn(M, K, generator=g).to(dev)
            wp = ops.split_weight_bf16x3((torch.randn(256, K, generator=g) * 0.05).to(dev))
            b, r = torch.randn(256, generator=g).to(dev), torch.randn(M, 256, generator=g).to(dev)
            gam, bet = (torch.rand(256, generator=g) + 0.5).to(dev), torch.randn(256, generator=g).to(dev)
            return lambda: ops.gemm_bf16x3_ln(a, wp, b, r, gam, bet, 1e-5)
        if form == 'strided':
            x = torch.randn(2, 128, 17, 23, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
            wp = ops.split_weight_bf16x3((torch.randn(256, 128, generator=g) * 0.05).to(dev))
            b = torch.randn(256, generator=g).to(dev)
            return lambda: ops.conv1x1_strided_split(x, wp, b, stride=2, relu=True).contiguous()
        st = 2 if form == 'conv3x3_s2' else 1
        x = torch.randn(3, 64, 19, 27, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        wp = ops.split_conv3x3_weight((torch.randn(128, 64, 3, 3, generator=g) * 0.04).to(dev))
        b = torch.randn(128, generator=g).to(dev)
        return lambda: ops.conv3x3_split(x, wp, b, stride=st, relu=True).contiguous()

    fn = run()
    with native.diag_build(9):
        old = fn().clone()
    with native.diag_build(19):     # the LDS-DMA generation's tile kernels (19: without the K-split small-row form
        new = fn().clone()          # the shipped selection takes at these few rows -- equal to rounding only)
    torch.cuda.synchronize()
    if form == 'ln':
        np.testing.assert_allclose(new.cpu().numpy(), old.cpu().numpy(), rtol=0, atol=4e-6)
    else:
        assert torch.equal(new, old), float((new - old).abs().max())


@pytest.mark.parametrize('form', ['rows', 'rows_res', 'rows_padded', 'rows_k64', 'ex', 'ex640', 'rows_n384', 'rows_n96',
                                  'conv3x3_n96', 'strided',
                                  'conv3x3', 'conv3x3_res', 'cat', 'grouped'])
def test_gemm_wide_tile_form_is_bit_identical(form):
    """The wide tile form of the LDS-DMA GEMM (a wave owns 32 rows x 256 columns, W fragments read
    quarter by quarter, ring of 2) against the narrow form (32 x 128 per wave, ring of 3): same
    products in the same order per accumulator -> bit-identical on every row source / epilogue,
    ragged M and the shortest K (4 slabs) included."""
    from pavenet_amd import native, ops
    g = torch.Generator().manual_seed(100 + len(form))
    dev = 'cuda'

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)

    def run():
        if form in ('rows', 'rows_res', 'rows_padded', 'rows_k64'):
            M, K, N = (1237, 192, 512) if form != 'rows_k64' else (129, 64, 256)
            n_out = 452 if form == 'rows_padded' else N
            a = rnd(M, K)
            wp = ops.split_weight_bf16x3(rnd(n_out, K, scale=0.05), pad=(form == 'rows_padded'))
            b = rnd(n_out)
            r = rnd(M, n_out) if form != 'rows' else None
            if form == 'rows_padded':
                assert wp.shape[2] == 512
                return lambda: ops.gemm_bf16x3(a, wp, b, r, relu=True, n_out=n_out)
            return lambda: ops.gemm_bf16x3(a, wp, b, r, relu=(form == 'rows'))
        if form in ('ex', 'ex640'):
            # (N = 640: wide tiles for the first 512 columns, a narrow one for the last 128, one launch)
            M, K, N, rows, ns = 901, 256, (768 if form == 'ex' else 640), 53, 256
            a, wp, tab = rnd(M, K), ops.split_weight_bf16x3(rnd(N, K, scale=0.05)), rnd(rows, N)
            return lambda: torch.cat(ops.gemm_bf16x3_ex(a, wp, None, tab, residual_rows=rows, n_split=ns), 1)
        if form == 'rows_n96':      # 96 real columns in 128-row planes: three column tiles (HRNet)
            a, wp = rnd(700, 192), ops.split_weight_bf16x3(rnd(96, 192, scale=0.05), pad=True)
            b, r = rnd(96), rnd(700, 96)
            return lambda: ops.gemm_bf16x3(a, wp, b, r, relu=True, n_out=96)
        if form == 'conv3x3_n96':
            x = rnd(2, 96, 19, 27).contiguous(memory_format=torch.channels_last)
            wp, b = ops.split_conv3x3_weight(rnd(96, 96, 3, 3, scale=0.04)), rnd(96)
            r = rnd(2, 96, 19, 27).contiguous(memory_format=torch.channels_last)
            return lambda: ops.conv3x3_split(x, wp, b, stride=1, relu=True, residual=r, cout=96).contiguous()
        if form == 'rows_n384':
            a, wp, b, r = rnd(517, 128), ops.split_weight_bf16x3(rnd(384, 128, scale=0.05)), rnd(384), rnd(517, 384)
            return lambda: ops.gemm_bf16x3(a, wp, b, r, relu=True)
        if form == 'strided':
            x = rnd(2, 128, 17, 23).contiguous(memory_format=torch.channels_last)
            wp, b = ops.split_weight_bf16x3(rnd(256, 128, scale=0.05)), rnd(256)
            return lambda: ops.conv1x1_strided_split(x, wp, b, stride=2, relu=True).contiguous()
        if form in ('conv3x3', 'conv3x3_res'):
            x = rnd(3, 48, 19, 27).contiguous(memory_format=torch.channels_last)
            wp, b = ops.split_conv3x3_weight(rnd(256, 48, 3, 3, scale=0.04)), rnd(256)
            r = rnd(3, 256, 19, 27).contiguous(memory_format=torch.channels_last) if form == 'conv3x3_res' else None
            return lambda: ops.conv3x3_split(x, wp, b, stride=1, relu=True, residual=r).contiguous()
        if form == 'cat':
            M = 1111
            a, a2 = rnd(M, 64), rnd(M, 128)
            wp, b = ops.split_weight_bf16x3(rnd(256, 192, scale=0.05)), rnd(256)
            return lambda: ops.gemm_bf16x3_cat(a, a2, wp, b, None, relu=True)
        M, G, K, gn = 707, 3, 128, 256
        a, wp, b = rnd(M, G * K), ops.split_weight_bf16x3(rnd(G * gn, K, scale=0.05)), rnd(G * gn)
        return lambda: ops.gemm_bf16x3_grouped(a, wp, b, gn, relu=True)

    fn = run()
    with native.diag_build(8):
        narrow = fn().clone()
    with native.diag_build(7):
        wide = fn().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(wide).all()
    assert torch.equal(wide, narrow), float((wide - narrow).abs().max())


@pytest.mark.parametrize('seed', range(16))
def test_gemm_forms_random_shapes_bit_identical(seed):
    """Property test over random shapes and epilogue options: the first-generation kernel (diag
    variant 9), the LDS-DMA generation with narrow tiles only (8) and with wide / mixed tiles forced
    wherever they apply (7) and the default dispatch must agree bit for bit -- single partial tiles,
    exact multiples of the 128-row tile, K = 64, two outputs and the row-periodic residual table
    included."""
    from pavenet_amd import native, ops
    rs = np.random.RandomState(1000 + seed)
    g = torch.Generator().manual_seed(2000 + seed)
    M = int(rs.choice([1, 31, 128, 129, 256, 1000, 2047, 2560]))
    K = int(rs.choice([64, 128, 192, 320, 1024]))
    N = int(rs.choice([64, 128, 256, 384, 512, 640, 768]))
    use_res, use_bias, relu = bool(rs.randint(2)), bool(rs.randint(2)), bool(rs.randint(2))
    table = use_res and N >= 256 and bool(rs.randint(2))
    nsplit = 256 if (N >= 512 and rs.randint(2)) else 0
    a = torch.randn(M, K, generator=g).cuda()
    wp = ops.split_weight_bf16x3((torch.randn(N, K, generator=g) * 0.05).cuda())
    b = torch.randn(N, generator=g).cuda() if use_bias else None
    rows = int(rs.choice([7, 53, 200])) if table else 0
    r = torch.randn(rows if table else M, N, generator=g).cuda() if use_res else None

    def run():
        if table or nsplit:
            o1, o2 = ops.gemm_bf16x3_ex(a, wp, b, r, residual_rows=rows, n_split=nsplit, relu=relu)
            return o1 if o2 is None else torch.cat([o1, o2], 1)
        return ops.gemm_bf16x3(a, wp, b, r, relu=relu)

    outs = {}
    for v in (9, 8, 7):
        with native.diag_build(v):
            outs[v] = run().clone()
    outs[0] = run().clone()     # the shipped library's own selection
    torch.cuda.synchronize()
    # (round 6: few-tile launches with K >= 512 take the K-split small-row form in the shipped selection -- the
    # same fp32 terms added in another association: equal to rounding there, bit-identical everywhere else)
    t32 = ((M + 31) // 32) * ((N + 31) // 32)
    ksplit = not nsplit and M <= 2048 and ((K >= 512 and t32 <= 4096) or (K >= 256 and t32 <= 1280))
    for v in (8, 7, 0):
        if v == 0 and ksplit:
            scale = float(outs[9].abs().max()) + 1e-6
            assert float((outs[0] - outs[9]).abs().max()) <= 8e-6 * scale, (seed, M, K, N)
            continue
        assert torch.equal(outs[v], outs[9]), (seed, M, K, N, v, float((outs[v] - outs[9]).abs().max()))
    assert torch.isfinite(outs[0]).all()


def test_fill_rows_sets_listed_rows_only():
    """pave_fill_rows_f32: the listed rows take the vector (or zeros), every other row is untouched, a row
    index outside the matrix is skipped; row-strided views work (the value half of a wider matrix)."""
    from pavenet_amd.ops import fill_rows_
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1000, 512, generator=g).cuda()
    rows = torch.tensor([0, 7, 999, 500, 1000, -1, 7], dtype=torch.int32).cuda()
    vals = torch.randn(256, generator=g).cuda()
    exp = x.clone()
    exp[[0, 7, 999, 500], :256] = vals
    fill_rows_(x[:, :256], rows, vals)
    assert torch.equal(x, exp)
    exp[[0, 7, 999, 500], 256:] = 0
    fill_rows_(x[:, 256:], rows, None)
    assert torch.equal(x, exp)
    fill_rows_(x, rows[:0], None)      # an empty list: nothing to do
    assert torch.equal(x, exp)


def test_enc_tile_fp16_output_is_the_rounded_fp32_output():
    """variant | 8: the sampler stores its rows as fp16 (fp16 operand mode: they only feed output_proj's MFMA) --
    exactly the fp32 rows rounded to nearest even, prepared and un-prepared input, idle tile slots untouched."""
    from pavenet_amd import ops
    levels = [(20, 28), (10, 14), (5, 7), (3, 4)]
    S = sum(h * w for h, w in levels)
    F_ = 2
    g = torch.Generator(device='cuda').manual_seed(3)
    value = torch.randn(F_, S, 8, 32, device='cuda', generator=g)
    proj = torch.randn(F_ * S, 384, device='cuda', generator=g)
    ref = torch.rand(1, F_ * S, 4, 2, device='cuda', generator=g)
    a32 = ops.deform_attn_enc_tile(value, proj, ref, levels_hw=levels)
    a16 = ops.deform_attn_enc_tile(value, proj, ref, levels_hw=levels, out_half=True)
    assert a16.dtype == torch.float16 and torch.equal(a16, a32.half())
    a = torch.randn(F_ * S, 256, device='cuda', generator=g)
    wp = ops.split_weight_bf16x3(torch.randn(640, 256, device='cuda', generator=g) * 0.05)
    table = torch.randn(S, 640, device='cuda', generator=g) * 0.1
    v, samp = ops.gemm_bf16x3_encproj(a, wp, table, ref.view(-1, 4, 2), levels)
    p32 = ops.deform_attn_enc_tile(v.view(F_, S, 8, 32), samp, None, levels_hw=levels, prepared=True)
    p16 = ops.deform_attn_enc_tile(v.view(F_, S, 8, 32), samp, None, levels_hw=levels, prepared=True, out_half=True)
    assert torch.equal(p16, p32.half())


def test_enc_tile_c_abi_refuses_unsupported_variants_before_launching():
    """pave_enc_deform_attn_tile_f32 called directly (ctypes): prepared input with the wide-window variant
    (5: the non-prepared kernel would read the null `ref`), (6) and bits above the 4-bit mask must come back
    as an error with NOTHING enqueued -- `out` keeps its contents."""
    import ctypes
    from pavenet_amd import native
    lib = native.load()
    levels = [(16, 24), (8, 12), (4, 6), (2, 3)]
    S = sum(h * w for h, w in levels)
    value = torch.randn(1, S, 8, 32, device='cuda')
    proj = torch.rand(S, 384, device='cuda')
    out = torch.full((S, 256), 3.0, device='cuda')
    hw = (ctypes.c_int * 8)(*[v for l in levels for v in l])
    for variant in (5, 6, 7, 13, 16, 20, -1):
        st = lib.pave_enc_deform_attn_tile_f32(value.data_ptr(), proj.data_ptr(), None, out.data_ptr(), 1, S,
                                               ctypes.cast(hw, ctypes.c_void_p), 384, variant, None, None)
        assert st != 0, variant
    torch.cuda.synchronize()
    assert bool((out == 3.0).all())


@pytest.mark.parametrize('n,H,W,Cin,Cout,res,relu', [(2, 40, 56, 48, 48, True, True), (1, 33, 21, 48, 48, False, False),
                                                      (3, 9, 11, 64, 36, True, False), (2, 19, 27, 16, 40, False, False),
                                                      (1, 7, 5, 96, 44, True, True), (1, 1, 1, 48, 48, False, False)])
def test_conv3x3_half_tail_form_vs_padded_form_and_fp64(n, H, W, Cin, Cout, res, relu):
    """33 .. 48 output channels of a 3x3 convolution (HRNet-w48's 48-channel branch): the columns behind the first
    32 run as v_mfma_f32_16x16x32_bf16 over PAIRS of K slabs (A operands re-laid with v_permlane32/16_swap, B pieces
    of both slabs' stages in one register) instead of zero-padded 32x32x16 products.  Against the padded form (diag
    variant 17): the first 32 channels bit for bit, the tail to fp32 summation order; against fp64; odd slab
    counts (K = 9 Cin padded to a multiple of 32), ragged tiles, borders, identity + ReLU, a NaN pixel."""
    from pavenet_amd import native, ops
    g = torch.Generator().manual_seed(n * 1000 + H * 10 + Cout)
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(n, Cout, H, W, generator=g) if res else None
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)   # noqa: E731
    wp = ops.split_conv3x3_weight(w.cuda())
    run = lambda xx: ops.conv3x3_split(cl(xx), wp, b.cuda(), relu=relu, residual=cl(r) if res else None, cout=Cout)  # noqa: E731
    got = run(x)
    with native.diag_build(17):
        pad = run(x).clone()
    torch.cuda.synchronize()
    assert torch.equal(got[:, :32], pad[:, :32])
    np.testing.assert_allclose(got[:, 32:].cpu().numpy(), pad[:, 32:].cpu().numpy(), rtol=2e-6, atol=2e-6)
    exp = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res:
        exp = exp + r.double()
    if relu:
        exp = exp.relu()
    np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    if H > 2 and not relu:      # (max(0, NaN) = 0: with the ReLU epilogue a NaN does not survive, in either form)
        xn = x.clone()
        xn[0, 1, 1, 1] = float('nan')
        gn = run(xn)
        bad = torch.isnan(gn[0]).any(0)
        assert bad[:3, :3].all() and int(bad.sum()) == min(3, H) * min(3, W)   # exactly the windows holding the pixel
        assert not torch.isnan(gn[1:]).any()


@pytest.mark.parametrize('seed', range(10))
def test_gemm_two_row_tiles_per_wave_form_is_bit_identical(seed):
    """The 64- / 96-column tile forms with TWO row tiles per wave (256-row blocks, every W fragment read once
    for both tiles; diag variant 16 forces the form, 15 forbids it) against one row tile per wave: the
    products of an accumulator keep their order, so plain rows (bias / residual / ReLU /
    padded output width) and the 3x3 implicit GEMM (48-, 64- and 96-channel maps, identity + ReLU, image
    borders, a NaN pixel) must agree bit for bit -- single rows, partial and exact 256-row tiles included."""
    from pavenet_amd import native, ops
    rs = np.random.RandomState(3000 + seed)
    g = torch.Generator().manual_seed(4000 + seed)
    if seed % 2 == 0:
        M = int(rs.choice([1, 200, 255, 256, 257, 1000, 2049]))
        K = int(rs.choice([64, 96, 256, 576]))
        N = int(rs.choice([48, 64, 192]))
        use_res, use_bias, relu = bool(rs.randint(2)), bool(rs.randint(2)), bool(rs.randint(2))
        rows = 0
        a = torch.randn(M, K, generator=g).cuda()
        wp = ops.split_weight_bf16x3((torch.randn(N, K, generator=g) * 0.05).cuda(), pad=True)
        b = torch.randn(N, generator=g).cuda() if use_bias else None
        r = torch.randn(rows if rows else M, N, generator=g).cuda() if use_res else None

        def run():
            return ops.gemm_bf16x3(a, wp, b, r, relu=relu, n_out=N)
        what = ('rows', M, K, N, use_res, rows, relu)
    else:
        n, H, W = int(rs.choice([1, 2, 3])), int(rs.choice([8, 19, 33])), int(rs.choice([11, 16, 27]))
        C = int(rs.choice([48, 64, 96]))
        use_res, relu = bool(rs.randint(2)), bool(rs.randint(2))
        x = torch.randn(n, C, H, W, generator=g)
        x[0, 1, 0, 0] = float('nan')
        xd = x.cuda().contiguous(memory_format=torch.channels_last)
        wp = ops.split_conv3x3_weight((torch.randn(C, C, 3, 3, generator=g) * 0.05).cuda())
        b = torch.randn(C, generator=g).cuda()
        r = torch.randn(n, C, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last) \
            if use_res else None

        def run():
            return torch.nan_to_num(ops.conv3x3_split(xd, wp, b, stride=1, relu=relu, residual=r, cout=C), nan=7.0)
        what = ('3x3', n, H, W, C, use_res, relu)
    with native.diag_build(15):
        one = run().clone()
    with native.diag_build(16):
        two = run().clone()
    torch.cuda.synchronize()
    assert torch.equal(one, two), (what, float((one - two).abs().max()))
    assert torch.isfinite(two).all()


@pytest.mark.parametrize('M,K,N,rows,nsplit', [(1000, 256, 640, 125, 256), (777, 64, 384, 0, 128),
                                               (300, 128, 256, 7, 0), (513, 256, 512, 0, 256)])
def test_gemm_bf16x3_ex_row_table_and_two_outputs(M, K, N, rows, nsplit):
    """pave_gemm_bf16x3_ex_f32: residual as a [rows, N] table indexed by m % rows, output cut at
    column nsplit into two dense matrices -- against fp64."""
    from pavenet_amd.ops import gemm_bf16x3_ex, split_weight_bf16x3
    g = torch.Generator().manual_seed(M + K + N + rows)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K**0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(rows if rows else M, N, generator=g)
    wp = split_weight_bf16x3(w.cuda())
    exp = a.double() @ w.double().t() + b.double()
    exp = exp + (r.double()[torch.arange(M) % rows] if rows else r.double())
    o1, o2 = gemm_bf16x3_ex(a.cuda(), wp, b.cuda(), r.cuda(), residual_rows=rows, n_split=nsplit)
    if nsplit:
        assert tuple(o1.shape) == (M, nsplit) and tuple(o2.shape) == (M, N - nsplit)
        assert o1.is_contiguous() and o2.is_contiguous()
        got = torch.cat([o1, o2], 1)
    else:
        assert o2 is None
        got = o1
    np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError):
        gemm_bf16x3_ex(a.cuda(), wp, None, None, n_split=64)          # not a multiple of 128
    with pytest.raises(RuntimeError):
        gemm_bf16x3_ex(a.cuda(), wp, None, r.cuda()[:3], residual_rows=5)   # table shape


@pytest.mark.parametrize('M,K', [(1000, 256), (257, 1024), (5, 64), (4100, 256)])
def test_gemm_bf16x3_layernorm_epilogue_vs_fp64(M, K):
    """pave_gemm_bf16x3_ln_f32 = LayerNorm(a W^T + bias + residual) gamma + beta in one launch,
    against the fp64 formulation; also in place over the residual, and without a residual."""
    from pavenet_amd.ops import gemm_bf16x3_ln, split_weight_bf16x3
    N = 256
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K**0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) * 2 + 0.5          # non-zero row means
    gam, bet = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    wp = split_weight_bf16x3(w.cuda())
    pre = a.double() @ w.double().t() + b.double() + r.double()
    exp = torch.nn.functional.layer_norm(pre, (N,), gam.double(), bet.double(), 1e-5)
    out = gemm_bf16x3_ln(a.cuda(), wp, b.cuda(), r.cuda(), gam.cuda(), bet.cuda(), 1e-5)
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)
    idt = r.cuda()
    out = gemm_bf16x3_ln(a.cuda(), wp, b.cuda(), idt, gam.cuda(), bet.cuda(), 1e-5, out=idt)
    assert out.data_ptr() == idt.data_ptr()
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)
    pre = a.double() @ w.double().t()
    exp = torch.nn.functional.layer_norm(pre, (N,), gam.double(), bet.double(), 1e-3)
    out = gemm_bf16x3_ln(a.cuda(), wp, None, None, gam.cuda(), bet.cuda(), 1e-3)
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)
    with pytest.raises(RuntimeError):
        gemm_bf16x3_ln(a.cuda(), split_weight_bf16x3(torch.randn(128, K).cuda()), None, None,
                       gam.cuda()[:128], bet.cuda()[:128], 1e-5)     # N != 256


@pytest.mark.parametrize('M,K,N', [(1000, 256, 1024), (300, 1024, 256), (129, 64, 512)])
def test_gemm_bf16x3_eight_wave_tile_equals_four_wave_tile(M, K, N):
    """The 128 x 256 / 8-wave tile form (tools switch) gives the same numbers as the 128 x 128
    form: same products, same accumulation order per output element -> bit-identical."""
    from pavenet_amd import native
    from pavenet_amd.ops import gemm_bf16x3, gemm_bf16x3_ex, split_weight_bf16x3
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K**0.5).cuda()
    b, r, ab = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda(), \
        torch.randn(K, generator=g).cuda()
    wp = split_weight_bf16x3(w)
    with native.diag_build(4):
        ref = gemm_bf16x3(a, wp, b, r, relu=True, a_bias=ab)
        ref2 = gemm_bf16x3_ex(a, wp, b, r[:7].contiguous(), residual_rows=7, n_split=256) \
            if N > 256 else None
    with native.diag_build(3):
        got = gemm_bf16x3(a, wp, b, r, relu=True, a_bias=ab)
        got2 = gemm_bf16x3_ex(a, wp, b, r[:7].contiguous(), residual_rows=7, n_split=256) \
            if N > 256 else None
    assert torch.equal(ref, got)
    if ref2 is not None:
        assert torch.equal(ref2[0], got2[0]) and torch.equal(ref2[1], got2[1])
    exact = torch.relu(torch.relu(a + ab).double() @ w.double().t() + b.double() + r.double())
    np.testing.assert_allclose(got.cpu().numpy(), exact.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('planes,rel', [(1, 6e-3), (2, 4e-5), (16, 8e-4)])
def test_gemm_bf16_split_reduced_planes(planes, rel):
    """1 plane = plain bf16 operands (round to nearest), 2 planes ~ 16 significand bits: the error
    against fp64 sits at the expected level for each (and the 3-plane form is 100x below)."""
    from pavenet_amd.ops import gemm_bf16x3, split_bf16x3, split_weight_bf16x3
    g = torch.Generator().manual_seed(planes)
    M, K, N = 777, 512, 256
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K**0.5
    b = torch.randn(N, generator=g)
    exact = a.double() @ w.double().t() + b.double()
    got = gemm_bf16x3(a.cuda(), split_weight_bf16x3(w.cuda(), planes), b.cuda(),
                      fp16=planes == 16).cpu().double()
    err = (got - exact).abs().max().item() / exact.abs().max().item()
    assert err < rel, err
    assert err > rel / 300, 'suspiciously exact: is the plane count honoured?'
    if planes == 1:     # the single plane is the round-to-nearest bf16 of the value
        p = split_bf16x3(w.cuda(), 1).cpu()[0]
        assert torch.equal(p.view(torch.bfloat16), w.to(torch.bfloat16))
    if planes == 16:    # PLANES_FP16: the round-to-nearest fp16 of the value
        p = split_bf16x3(w.cuda(), 16).cpu()[0]
        assert torch.equal(p.view(torch.float16), w.to(torch.float16))


@pytest.mark.parametrize('form', ['rows', 'rows_small', 'rows_n48', 'ex', 'ln_wide', 'ln_8wave', 'ln_small', 'cat',
                                  'grouped', 'strided', 'conv3x3', 'conv3x3_c48_res', 'conv3x3_splitk', 'stem',
                                  'chain', 'encproj'])
def test_fp16_operand_mode_on_every_fused_form(form):
    """BASELINE configs[4]'s "fp16 MFMA projections" on the LDS-DMA kernel generation (nplanes =
    PAVE_PLANES_FP16): ONE plane of fp16 weights, the fp32 activation rows rounded to fp16 at operand fetch,
    fp32 accumulation -- i.e. the EXACT product of the fp16-rounded operands up to fp32 summation error, for
    every form the exact 3-plane mode has (tile / wide / small-row GEMMs, two outputs + row table, LayerNorm
    epilogue in its three forms, two row sources, grouped columns, strided pixels, 3x3 incl. padded 48-channel
    planes + identity and the split-K form, the 7x7 stem, the layer1 chain, the encoder projection with the
    sampler arithmetic).  Plain rows / `ex` / 3x3 are also bit-identical to the first-generation fp16 kernels."""
    from pavenet_amd import native, ops
    F16 = ops.PLANES_FP16
    g = torch.Generator().manual_seed(len(form) * 7 + 1)
    h = lambda t: t.half().double()                                   # noqa: E731  (what the kernel multiplies)
    rnd = lambda *sh, sc=1.0: torch.randn(*sh, generator=g) * sc       # noqa: E731
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)   # noqa: E731

    def close(got, exp, tol=2e-5):
        exp = exp.double()
        err = float((got.double().cpu() - exp).abs().max() / exp.abs().max().clamp_min(1e-6))
        assert err < tol, (form, err)

    if form in ('rows', 'rows_small', 'rows_n48'):
        M, K, N = {'rows': (9000, 256, 384), 'rows_small': (300, 1024, 256), 'rows_n48': (20000, 96, 48)}[form]
        a, w, b, r = rnd(M, K), rnd(N, K, sc=K ** -0.5), rnd(N), rnd(M, N)
        wp = ops.split_weight_bf16x3(w.cuda(), F16, pad=True)
        assert wp.dtype == torch.float16 and wp.shape[1] == 1
        got = ops.gemm_bf16x3(a.cuda(), wp, b.cuda(), r.cuda(), relu=True, n_out=N)
        close(got, torch.relu(h(a) @ h(w).t() + b.double() + r.double()))
        if form == 'rows':
            with native.diag_build(9):      # first-generation fp16 kernel: same conversion, same order
                old = ops.gemm_bf16x3(a.cuda(), wp, b.cuda(), r.cuda(), relu=True).clone()
            assert torch.equal(got, old)
            # wrong mode refused: a float16 plane where only 3 bf16 planes or fp16 are valid is fine, int16 x1 is not
            with pytest.raises(RuntimeError):
                ops.gemm_bf16x3_ln(a.cuda(), ops.split_weight_bf16x3(w[:256].cuda(), 1), None, None, b[:256].cuda(),
                                   b[:256].cuda(), 1e-5)
    elif form == 'ex':
        M, K, N, rows = 9000, 256, 640, 125
        a, w, tab = rnd(M, K), rnd(N, K, sc=K ** -0.5), rnd(rows, N)
        wp = ops.split_weight_bf16x3(w.cuda(), F16)
        o1, o2 = ops.gemm_bf16x3_ex(a.cuda(), wp, None, tab.cuda(), residual_rows=rows, n_split=256)
        close(torch.cat([o1, o2], 1), h(a) @ h(w).t() + tab.double()[torch.arange(M) % rows])
        with native.diag_build(9):
            p1, p2 = ops.gemm_bf16x3_ex(a.cuda(), wp, None, tab.cuda(), residual_rows=rows, n_split=256)
            p1, p2 = p1.clone(), p2.clone()
        assert torch.equal(o1, p1) and torch.equal(o2, p2)
    elif form.startswith('ln'):
        M, K = (300, 1024) if form == 'ln_small' else (3000, 256)
        a, w, b, r, ga, be = rnd(M, K), rnd(256, K, sc=K ** -0.5), rnd(256), rnd(M, 256), rnd(256) + 1.0, rnd(256)
        wp = ops.split_weight_bf16x3(w.cuda(), F16)
        args = (a.cuda(), wp, b.cuda(), r.cuda(), ga.cuda(), be.cuda(), 1e-5)
        if form == 'ln_small':
            got = ops.gemm_bf16x3_ln(*args)
        else:
            with native.diag_build(14 if form == 'ln_wide' else 13):
                got = ops.gemm_bf16x3_ln(*args).clone()
        exp = torch.nn.functional.layer_norm(h(a) @ h(w).t() + b.double() + r.double(), (256,), ga.double(),
                                             be.double(), 1e-5)
        close(got, exp, 3e-5)
    elif form == 'cat':
        M, K1, K2, N = 9000, 64, 64, 256
        a, a2, w, b = rnd(M, K1).relu(), rnd(M, K2).relu(), rnd(N, K1 + K2, sc=0.1), rnd(N)
        got = ops.gemm_bf16x3_cat(a.cuda(), a2.cuda(), ops.split_weight_bf16x3(w.cuda(), F16), b.cuda(), relu=True)
        close(got, torch.relu(torch.cat([h(a), h(a2)], 1) @ h(w).t() + b.double()))
    elif form == 'grouped':
        M, K, G, gn = 1200, 128, 3, 64
        a, w, b = rnd(M, G * K), rnd(G * gn, K, sc=0.1), rnd(G * gn)
        got = ops.gemm_bf16x3_grouped(a.cuda(), ops.split_weight_bf16x3(w.cuda(), F16), b.cuda(), gn, relu=True)
        exp = torch.cat([h(a[:, i * K:(i + 1) * K]) @ h(w[i * gn:(i + 1) * gn]).t() for i in range(G)], 1) + b.double()
        close(got, torch.relu(exp))
    elif form == 'strided':
        x, w, b = rnd(2, 64, 31, 45), rnd(128, 64, sc=0.1), rnd(128)
        got = ops.conv1x1_strided_split(cl(x), ops.split_weight_bf16x3(w.cuda(), F16), b.cuda(), stride=2, relu=True)
        exp = torch.relu(torch.nn.functional.conv2d(h(x), h(w)[:, :, None, None], b.double(), stride=2))
        close(got, exp)
    elif form in ('conv3x3', 'conv3x3_c48_res', 'conv3x3_splitk'):
        n, H, W, Cin, Cout, st = {'conv3x3': (2, 33, 47, 64, 128, 2), 'conv3x3_c48_res': (2, 40, 56, 48, 48, 1),
                                  'conv3x3_splitk': (1, 26, 42, 1024, 256, 2)}[form]
        x, w, b = rnd(n, Cin, H, W), rnd(Cout, Cin, 3, 3, sc=(9 * Cin) ** -0.5), rnd(Cout)
        res = rnd(n, Cout, (H - 1) // st + 1, (W - 1) // st + 1) if form == 'conv3x3_c48_res' else None
        wp = ops.split_conv3x3_weight(w.cuda(), F16)
        got = ops.conv3x3_split(cl(x), wp, b.cuda(), stride=st, relu=True, cout=Cout,
                                residual=cl(res) if res is not None else None)
        exp = torch.nn.functional.conv2d(h(x), h(w), b.double(), stride=st, padding=1)
        close(got, torch.relu(exp + (res.double() if res is not None else 0)))
        if form == 'conv3x3':
            with native.diag_build(9):
                old = ops.conv3x3_split(cl(x), wp, b.cuda(), stride=st, relu=True).clone()
            assert torch.equal(got, old)
    elif form == 'stem':
        x, w, b = rnd(2, 3, 64, 96), rnd(64, 3, 7, 7, sc=0.08), rnd(64)
        wp = ops.split_stem7x7_weight(w.cuda(), F16)
        assert wp.dtype == torch.float16 and tuple(wp.shape) == (23, 1, 64, 16)
        got = ops.conv7x7s2_nchw_split(x.cuda(), wp, b.cuda(), relu=True)
        close(got, torch.relu(torch.nn.functional.conv2d(h(x), h(w), b.double(), stride=2, padding=3)))
        with pytest.raises(RuntimeError):     # odd width: no fp16 form of the per-lane window kernel
            ops.conv7x7s2_nchw_split(x[..., :95].contiguous().cuda(), wp, b.cuda())
    elif form == 'chain':
        n, H, W = 2, 19, 27
        c1, xin = cl(rnd(n, 64, H, W).relu()), cl(rnd(n, 256, H, W))
        w2p, b2 = ops.split_conv3x3_weight(rnd(64, 64, 3, 3, sc=0.05).cuda(), F16), rnd(64, sc=0.1).cuda()
        w3p, b3 = ops.split_weight_bf16x3(rnd(256, 64, sc=0.08).cuda(), F16), rnd(256, sc=0.1).cuda()
        w1p, b1 = ops.split_weight_bf16x3(rnd(64, 256, sc=0.05).cuda(), F16), rnd(64, sc=0.1).cuda()
        with native.diag_build(19):    # (the separate launches on the tile kernels: at these few rows the shipped
            c2 = ops.conv3x3_split(c1, w2p, b2, relu=True)     # selection takes the K-split small-row form)
            exp_out = ops.gemm_bf16x3(c2.permute(0, 2, 3, 1).reshape(-1, 64), w3p, b3,
                                      xin.permute(0, 2, 3, 1).reshape(-1, 256), relu=True)
            exp_c1n = ops.gemm_bf16x3(exp_out, w1p, b1, relu=True)
        for v in (15, 16):      # 128- and 256-row tiles of the chain launch
            with native.diag_build(v):
                out, c1n = ops.bottleneck_chain(c1, w2p, b2, w3p, b3, residual=xin, w1n_planes=w1p, b1n=b1)
                torch.cuda.synchronize()
            assert torch.equal(out.permute(0, 2, 3, 1).reshape(-1, 256), exp_out)
            assert torch.equal(c1n.permute(0, 2, 3, 1).reshape(-1, 64), exp_c1n)
    else:   # encproj: the sampler arithmetic in the epilogue = the sampler's own, on the fp16 products
        levels, F_ = [(16, 24), (8, 12), (4, 6), (2, 3)], 3
        S = sum(a_ * b_ for a_, b_ in levels)
        M = F_ * S
        a, w, tab = rnd(M, 256).cuda(), (rnd(640, 256) * 0.05).cuda(), (rnd(S, 640) * 0.1).cuda()
        wp = ops.split_weight_bf16x3(w, F16)
        ref = torch.rand(M, 4, 2, generator=g).cuda()
        v0, proj = ops.gemm_bf16x3_ex(a, wp, None, tab, residual_rows=S, n_split=256)
        v1, samp = ops.gemm_bf16x3_encproj(a, wp, tab, ref, levels)
        assert torch.equal(v0, v1)
        raw = ops.deform_attn_enc_tile(v0.view(F_, S, 8, 32), proj, ref.view(1, M, 4, 2), levels_hw=levels)
        pre = ops.deform_attn_enc_tile(v1.view(F_, S, 8, 32), samp, None, levels_hw=levels, prepared=True)
        assert torch.equal(raw, pre)
    torch.cuda.synchronize()


@pytest.mark.parametrize('M', [1000, 128, 4100])
def test_fp16_activations_between_two_launches_change_no_value(M):
    """pave_gemm_fp16_act_f32: the hidden activation of an FFN stored as fp16 between fc1 and fc2 + LayerNorm
    (fp16 operand mode).  fc1 with an fp16 output = the fp32-output launch rounded to fp16, bit for bit; fc2 +
    identity + LayerNorm on the fp16 rows = the same launch fed the fp32 rows (which it would round to fp16 at
    operand fetch), bit for bit -- so the pair gives exactly the values of the fp32-activation chain; and against
    the fp64 formulation on fp16-rounded operands.  Ragged M, in-place identity."""
    from pavenet_amd import ops
    F16 = ops.PLANES_FP16
    g = torch.Generator().manual_seed(M)
    x, w1, b1 = torch.randn(M, 256, generator=g), torch.randn(1024, 256, generator=g) / 16, torch.randn(1024, generator=g)
    w2, b2 = torch.randn(256, 1024, generator=g) / 32, torch.randn(256, generator=g)
    idt, ga, be = torch.randn(M, 256, generator=g), torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    p1, p2 = ops.split_weight_bf16x3(w1.cuda(), F16), ops.split_weight_bf16x3(w2.cuda(), F16)
    xd = x.cuda()
    h32 = ops.gemm_fp16_act(xd, p1, b1.cuda(), relu=True)                      # fp32 in, fp32 out
    h16 = ops.gemm_fp16_act(xd, p1, b1.cuda(), relu=True, out_half=True)       # fp32 in, fp16 out
    assert h16.dtype == torch.float16 and torch.equal(h16, h32.half())
    from pavenet_amd import native
    with native.diag_build(19):     # (tile kernels: the shipped selection splits K over a block's waves at these rows)
        assert torch.equal(h32, ops.gemm_bf16x3(xd, p1, b1.cuda(), relu=True))     # = the ordinary fp16-mode launch
    ln = (ga.cuda(), be.cuda(), 1e-5)
    o32 = ops.gemm_fp16_act(h32, p2, b2.cuda(), residual=idt.cuda(), ln=ln)
    o16 = ops.gemm_fp16_act(h16, p2, b2.cuda(), residual=idt.cuda(), ln=ln)
    assert torch.equal(o16, o32)
    buf = idt.cuda().clone()
    assert ops.gemm_fp16_act(h16, p2, b2.cuda(), residual=buf, ln=ln, out=buf).data_ptr() == buf.data_ptr()
    assert torch.equal(buf, o16)
    hh = torch.relu(x.half().double() @ w1.half().double().t() + b1.double()).half().double()
    exp = torch.nn.functional.layer_norm(hh @ w2.half().double().t() + b2.double() + idt.double(), (256,), ga.double(),
                                         be.double(), 1e-5)
    np.testing.assert_allclose(o16.cpu().numpy(), exp.numpy(), rtol=3e-4, atol=3e-4)   # (h rounds to fp16 near ties)
    # fp16 rows into a plain fp32-output launch as well
    y16 = ops.gemm_fp16_act(h16, ops.split_weight_bf16x3(torch.cat([w2, w2], 0).cuda(), F16), None)
    y32 = ops.gemm_fp16_act(h32, ops.split_weight_bf16x3(torch.cat([w2, w2], 0).cuda(), F16), None)
    assert torch.equal(y16, y32)


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride', [(2, 13, 17, 64, 64, 1), (1, 20, 9, 128, 128, 1),
                                                   (3, 14, 22, 128, 256, 2), (2, 9, 9, 256, 64, 2),
                                                   (1, 40, 56, 64, 192, 1)])
def test_conv3x3_split_vs_torch(N, H, W, Cin, Cout, stride):
    """The split-operand kernel as an implicit-GEMM 3x3 convolution: 3 planes at fp32 level, one
    fp16 plane at the 16-bit level, zero padding and stride handled in the operand loader."""
    from pavenet_amd.ops import conv3x3_split, split_conv3x3_weight
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin**0.5)
    b = torch.randn(Cout, generator=g)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    exp = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride, 1)
    out = conv3x3_split(xd, split_conv3x3_weight(w.cuda(), 3), b.cuda(), stride=stride)
    assert out.shape == exp.shape
    np.testing.assert_allclose(out.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    out = conv3x3_split(xd, split_conv3x3_weight(w.cuda(), 3), b.cuda(), stride=stride, relu=True)
    np.testing.assert_allclose(out.cpu().numpy(), torch.relu(exp).numpy(), rtol=1e-5, atol=1e-5)
    out = conv3x3_split(xd, split_conv3x3_weight(w.cuda(), 16), None, stride=stride, fp16=True)
    np.testing.assert_allclose(out.cpu().numpy(), (exp - b.double()[None, :, None, None]).numpy(),
                               rtol=5e-3, atol=5e-3)


def test_kernels_survive_non_finite_inputs():
    """NaN / Inf sampling locations, offsets and logits must never turn into out-of-range reads
    (the reference kernel's range tests are false for NaN, ms_deform_attn_cuda_kernel.cuh:226-233):
    results may be NaN, the process must survive."""
    from pavenet_amd.ops import (deform_attn_grid_fused, deform_attn_pose_fused,
                                 ms_deform_attn_forward)
    shapes, lsi, sd, ld = _levels(LEVELS)
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(9)
    v = torch.randn(2, S, 8, 32, generator=g).cuda()
    bad = torch.tensor([float('nan'), float('inf'), -float('inf'), 1e30, -1e30])
    loc = torch.rand(2, 40, 8, 4, 4, 2, generator=g)
    loc.view(-1)[::7] = bad.repeat(loc.numel() // 7 // 5 + 1)[:loc.view(-1)[::7].numel()]
    aw = torch.rand(2, 40, 8, 4, 4, generator=g)
    aw.view(-1)[::11] = float('nan')
    out = ms_deform_attn_forward(v, sd, ld, loc.cuda(), aw.cuda(), 64)
    torch.cuda.synchronize()
    assert out.shape == (2, 40, 256)
    # fused encoder form (T = 1, head-major kernel) and T = 2 form: poisoned projections / refs
    for T in (1, 2):
        U = 96
        proj = torch.randn(U, T * 8 * 16 * 3, generator=g)
        proj.view(-1)[::13] = bad.repeat(proj.numel() // 13 // 5 + 1)[:proj.view(-1)[::13].numel()]
        ref = torch.rand(T, U, 4, 2, generator=g)
        ref.view(-1)[::17] = float('nan')
        vv = torch.randn(2 * T, S, 8, 32, generator=g).cuda()
        o = deform_attn_grid_fused(vv, sd, ld, proj.cuda(), ref.cuda(), T=T, n_clips=2,
                                   units_per_clip=U // 2)
        torch.cuda.synchronize()
        assert o.shape == (U, 256)
    K, Q, T = 15, 6, 3
    proj = torch.randn(2 * Q, T * 8 * 4 * K * 3, generator=g)
    proj.view(-1)[::19] = float('nan')
    ref = torch.rand(2, T * Q, 4, 2 * K, generator=g)
    ref.view(-1)[::23] = float('inf')
    vv = torch.randn(2 * T, S, 8, 32, generator=g).cuda()
    o = deform_attn_pose_fused(vv, sd, ld, proj.cuda(), ref.cuda(), T=T, n_clips=2, num_query=Q,
                               num_keypoints=K)
    torch.cuda.synchronize()
    assert o.shape == (2 * Q, 256)


@pytest.mark.parametrize('n_seq,L', [(4, 300), (80, 15), (2, 33), (3, 1), (1, 97), (2, 568)])
def test_mha_core_vs_fp64(n_seq, L):
    """pave_mha_core_f32 (the scaled-dot-product core of the decoders' self-attention,
    bricks/transformer.py:406-551 -> nn.MultiheadAttention) against softmax(q k^T / sqrt(d)) v in
    fp64; logits of a few units so that the softmax is neither flat nor one-hot."""
    from pavenet_amd.ops import mha_core
    H, d = 8, 32
    E = H * d
    g = torch.Generator().manual_seed(n_seq * 1000 + L)
    qkv = torch.randn(n_seq * L, 3 * E + 64, generator=g)[:, :3 * E + 64]   # ld > 3 E: a padded row
    qkv[:, :E] *= 1.5
    got = mha_core(qkv.cuda(), n_seq, L, H).cpu()
    x = qkv[:, :3 * E].double().view(n_seq, L, 3, H, d)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), -1)
    exp = (p @ v).transpose(1, 2).reshape(n_seq * L, E)
    assert tuple(got.shape) == (n_seq * L, E)
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=2e-5, atol=2e-6)
    with pytest.raises(RuntimeError):
        mha_core(qkv.cuda(), n_seq, L + 1, H)
    if L == 568:
        big = torch.zeros(569, 3 * E, device='cuda')
        with pytest.raises(RuntimeError):
            mha_core(big, 1, 569, H)      # K and V of a head no longer fit in LDS


@pytest.mark.parametrize('L,N', [(300, 4), (15, 40)])
def test_self_attention_split_path_vs_reference_module(L, N):
    """bricks.MultiheadAttention in the headline GEMM mode (q|k|v split GEMM with the positional
    table in its epilogue, pave_mha_core_f32, out_proj + identity + LayerNorm GEMM: three launches
    of this package's kernels) against nn.MultiheadAttention + LayerNorm run in fp64."""
    from pavenet_amd import bricks
    torch.manual_seed(L)
    m = bricks.MultiheadAttention(256, 8).cuda().eval()
    norm = torch.nn.LayerNorm(256).cuda()
    with torch.no_grad():
        m.attn.in_proj_bias.normal_(0, 0.1)
        m.attn.out_proj.bias.normal_(0, 0.1)
        norm.weight.normal_(1, 0.1)
        norm.bias.normal_(0, 0.1)
    emb = torch.randn(L, 512, device='cuda')
    pos = emb[:, :256].unsqueeze(0).expand(N, -1, -1).transpose(0, 1)    # [L, N, E], stride 0 over N
    x = torch.randn(N, L, 256, device='cuda').transpose(0, 1)           # seq-first view
    old = bricks.get_gemm_mode()
    try:
        bricks.set_gemm_mode('bf16x3')
        with torch.no_grad():
            got = m(x, query_pos=pos, post_norm=norm)
            assert '_pave_qkv' in m.__dict__, 'the split path did not run'
            again = m(x, query_pos=pos, post_norm=norm)
        assert torch.equal(got, again)
    finally:
        bricks.set_gemm_mode(old)
    md, nd = copy.deepcopy(m).double(), copy.deepcopy(norm).double()
    with torch.no_grad():
        exp = nd(md._forward_reference(x.double(), query_pos=pos.double()))
    np.testing.assert_allclose(got.cpu().numpy(), exp.cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('rows,n,k', [(4, 22323, 300), (4, 300, 20), (1, 7, 7), (3, 1000, 1), (2, 32768, 1024),
                                      (5, 1025, 17)])
def test_topk_rows_vs_torch(rows, n, k):
    """pave_topk_rows_f32 (the proposal / score top-k, OT:21383-21385, HEAD:1416) against torch.topk:
    same values in the same (descending) order, indices that address those values; exact ties are
    returned lowest index first; strided views are read in place."""
    from pavenet_amd.ops import topk_rows
    g = torch.Generator().manual_seed(rows * 7 + n)
    x = torch.randn(rows, n, generator=g)
    if n >= 1000:
        x[:, ::97] = x[:, 5:6]          # exact ties (also across the k-th value for some rows)
        x[0, 3] = float('inf')
        x[0, 4] = float('-inf')
    xd = x.cuda()
    v, i = topk_rows(xd, k)
    ev = torch.topk(x, k, dim=1)[0]
    assert torch.equal(v.cpu(), ev)
    assert torch.equal(torch.gather(x, 1, i.cpu()), ev)
    for r in range(rows):                                   # a selection: no index twice
        assert len(set(i[r].tolist())) == k
    # ties: equal values come out with ascending indices
    vi = torch.stack([v.cpu(), -i.cpu().float()], -1)
    for r in range(rows):
        for a in range(k - 1):
            assert (vi[r, a, 0] > vi[r, a + 1, 0]) or (vi[r, a, 0] == vi[r, a + 1, 0] and i[r, a] < i[r, a + 1])
    # a strided view: one column of a [rows, n, 4] tensor (the class logit of the proposal branch)
    wide = torch.randn(rows, n, 4, generator=g).cuda()
    v2, i2 = topk_rows(wide[..., 1], k)
    assert torch.equal(v2.cpu(), torch.topk(wide[..., 1].cpu(), k, dim=1)[0])
    assert torch.equal(torch.gather(wide[..., 1], 1, i2), v2)
    x[0, 0] = float('nan')                                   # NaN ranks first, as in torch.topk
    v3, i3 = topk_rows(x.cuda(), k)
    assert int(i3[0, 0]) == 0 and torch.isnan(v3[0, 0])
    with pytest.raises(RuntimeError):
        topk_rows(xd, n + 1)


@pytest.mark.parametrize('n,T', [(1, 3), (4, 7)])
def test_proposal_query_kernels_vs_torch(n, T):
    """The two launches behind the proposal top-k (OT:21386-21418) against the tensor expressions they replace:
    tgt = gather(output_memory), query = tgt + query_embed, kpt[..., 0::2 / 1::2] += gathered proposal logits (with
    +inf proposals), reference points = sigmoid(kpt) repeated for the T frames -- bit for bit; strided inputs (the
    centre frames of a [n*T, S, C] memory, a 30-column slice of a 32-column matrix)."""
    from pavenet_amd.ops import gather_rows_add, proposal_refs_
    g = torch.Generator().manual_seed(11 + n)
    S, C, Q, K2 = 997, 256, 300, 30
    mem = torch.randn(n * T, S, C, generator=g).cuda()
    src = mem[T // 2::T]                                         # [n, S, C], batch stride T S C
    idx = torch.stack([torch.randperm(S, generator=g)[:Q] for _ in range(n)]).cuda()
    add = torch.randn(Q, C, generator=g).cuda()
    rows, total = gather_rows_add(src, idx, add)
    exp = torch.gather(src, 1, idx.unsqueeze(-1).repeat(1, 1, C))
    assert torch.equal(rows, exp) and torch.equal(total, exp + add.unsqueeze(0))
    assert torch.equal(gather_rows_add(src, idx), exp)
    for shared in (False, True):
        props = torch.randn(1 if shared else n, S, 2, generator=g).cuda() * 3
        props[:, ::13] = float('inf')                            # invalid proposals (OT:21204)
        wide = torch.randn(n * Q, 32, generator=g).cuda()
        keep = wide.clone()
        kpt = wide[:, :K2].unflatten(0, (n, Q))
        ref = kpt.clone()
        tp = torch.gather(props.expand(n, -1, -1), 1, idx.unsqueeze(-1).repeat(1, 1, 2))
        ref[..., 0::2] += tp[..., 0:1]
        ref[..., 1::2] += tp[..., 1:2]
        refs = proposal_refs_(kpt, props, idx, T)
        assert torch.equal(kpt, ref) and torch.equal(wide[:, K2:], keep[:, K2:])
        exp_refs = ref.sigmoid().repeat(1, T, 1)
        assert refs.shape == (n, T * Q, K2)
        np.testing.assert_allclose(refs.cpu().numpy(), exp_refs.cpu().numpy(), rtol=0, atol=1.2e-7)
        assert torch.equal(refs[:, :Q], refs[:, (T - 1) * Q:])
    bad = idx.clone()
    bad[0, 0], bad[-1, -1] = -1, S                               # never dereferenced
    rows = gather_rows_add(src, bad)
    assert torch.equal(rows[0, 0], torch.zeros(C, device='cuda')) and torch.equal(rows[0, 1], exp[0, 1])
    with pytest.raises(RuntimeError):
        gather_rows_add(src.cpu(), idx.cpu())


def test_gather_frame_poses_and_pose_finalize_vs_torch():
    """The selection gather (HEAD:1419-1427, 610) and the fused post-processing (HEAD:1440-1490,
    get_p) against the tensor expressions they replace."""
    from pavenet_amd.ops import gather_frame_poses, pose_finalize
    g = torch.Generator().manual_seed(3)
    B, T, Q, N, K = 3, 5, 300, 20, 15
    poses = torch.rand(B, T * Q, 2 * K, generator=g).cuda()
    idx = torch.stack([torch.randperm(Q, generator=g)[:N] for _ in range(B)]).cuda()
    got = gather_frame_poses(poses, idx, T)
    gidx = idx.unsqueeze(-1).expand(-1, -1, 2 * K)
    exp = torch.cat([torch.gather(poses[:, t * Q:(t + 1) * Q], 1, gidx).reshape(B * N, 2 * K)
                     for t in range(T)], 0)
    assert torch.equal(got.flatten(0, 1), exp)
    kp = (torch.rand(B, N, K, 2, generator=g) * 1.2 - 0.1).cuda()      # some outside [0, 1]
    sg = (torch.rand(B, N, K, 2, generator=g) * 0.5 + 0.02).cuda()
    sc = torch.rand(B, N, generator=g).cuda()
    wh = torch.tensor([[1344., 800.], [1200., 780.], [640., 480.]]).cuda()
    sf = torch.tensor([[1.5, 1.25], [0.8, 0.9], [1., 1.]]).cuda()
    sg4 = torch.zeros(B, N, K, 4).cuda()          # the sigma branch's output as its GEMM leaves it: [.., 4] rows
    sg4[..., :2] = sg
    for rescale in (False, True):
        dk, db = pose_finalize(kp, sg, sc, wh, sf if rescale else None)
        dk4, db4 = pose_finalize(kp, sg4[..., :2], sc, wh, sf if rescale else None)
        assert torch.equal(dk, dk4) and torch.equal(db, db4)
        whb, sfb = wh.view(B, 1, 1, 2), sf.view(B, 1, 1, 2)
        k = kp * whb
        k = torch.minimum(k.clamp(min=0), whb)
        if rescale:
            k = k / sfb
        x1, y1 = k[..., 0].min(2, keepdim=True)[0], k[..., 1].min(2, keepdim=True)[0]
        x2, y2 = k[..., 0].max(2, keepdim=True)[0], k[..., 1].max(2, keepdim=True)[0]
        eb = torch.cat([x1, y1, x2, y2, sc.unsqueeze(-1)], 2)
        p = 1 - torch.exp(-(0.2 / sg))
        p = (p[..., 0] * p[..., 1])[..., None] * 0.7
        p5 = p ** 5
        ek = torch.cat(((k * p5) / (p5 + 1e-10), sc[:, :, None, None] * p), 3)
        assert torch.equal(db, eb)
        np.testing.assert_allclose(dk.cpu().numpy(), ek.cpu().numpy(), rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize('cat_dim', [0, 1])
def test_ref_update_frames_equals_copy_plus_ref_update(cat_dim):
    from pavenet_amd.ops import ref_update, ref_update_frames
    g = torch.Generator().manual_seed(11 + cat_dim)
    T, o, op = 5, 30 if cat_dim else 2, 64
    lead = (4, 300) if cat_dim else (40, 15)
    R = lead[0] * lead[1]
    y = torch.randn(R, T * op, generator=g).cuda()
    yt = y.view(R, T, op)[:, :, :o].permute(1, 0, 2).reshape((T,) + lead + (o,))
    if cat_dim == 0:
        cat = yt.reshape((T * lead[0], lead[1], o))
    else:
        cat = yt.permute(1, 0, 2, 3).reshape(lead[0], T * lead[1], o)
    ref = torch.rand(cat.shape, generator=g).cuda()
    exp = ref_update(cat.contiguous(), ref)
    got = ref_update_frames(y, ref, T, o, lead[1] if cat_dim else R)
    assert torch.equal(got, exp)


@pytest.mark.parametrize('levels,F,sigma', [([(100, 168), (50, 84), (25, 42), (13, 21)], 2, 0.9),
                                            ([(16, 20), (8, 10), (4, 5), (2, 3)], 3, 3.0)])
def test_encoder_projection_epilogue_equals_sampler_side_arithmetic(levels, F, sigma):
    """pave_gemm_bf16x3_encproj_f32 + the sampler in its prepared mode (softmax over the 16 logits
    and location -> pixel arithmetic done in the merged GEMM's epilogue, MO:373-404) against the
    plain merged GEMM + the sampler doing that arithmetic itself: the same bits (one definition of
    the arithmetic, csrc/pave_enc_math.h), value matrix included; non-trivial valid ratios, ragged M."""
    from pavenet_amd import ops
    S = sum(h * w for h, w in levels)
    M, K = F * S, 256
    g = torch.Generator(device='cuda').manual_seed(S + F)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(640, K, device='cuda', generator=g) * 0.05
    w[256:512] *= sigma * 0.3          # offsets of ~sigma pixels
    table = torch.randn(S, 640, device='cuda', generator=g) * 0.1
    wp = ops.split_weight_bf16x3(w)
    ys = torch.cat([((torch.arange(h * w_, device='cuda') // w_).float() + 0.5) / h for h, w_ in levels])
    xs = torch.cat([((torch.arange(h * w_, device='cuda') % w_).float() + 0.5) / w_ for h, w_ in levels])
    vr = torch.tensor([[0.93, 0.88], [0.95, 0.9], [1.0, 0.85], [0.9, 1.0]], device='cuda')   # per level
    ref = (torch.stack([xs, ys], -1)[None, :, None, :] * vr[None, None]).expand(F, S, 4, 2).reshape(M, 4, 2).contiguous()
    v0, proj = ops.gemm_bf16x3_ex(a, wp, None, table, residual_rows=S, n_split=256)
    v1, samp = ops.gemm_bf16x3_encproj(a, wp, table, ref, levels)
    assert torch.equal(v0, v1)
    raw = ops.deform_attn_enc_tile(v0.view(F, S, 8, 32), proj, ref.view(1, M, 4, 2), levels_hw=levels)
    pre = ops.deform_attn_enc_tile(v1.view(F, S, 8, 32), samp, None, levels_hw=levels, prepared=True)
    torch.cuda.synchronize()
    assert torch.isfinite(pre).all()
    assert torch.equal(raw, pre), float((raw - pre).abs().max())
    # the prepared matrix is what it says: softmax weights sum to one per (row, head)
    wsum = samp[:, 256:].view(M, 8, 16).sum(-1)
    np.testing.assert_allclose(wsum.cpu().numpy(), 1.0, rtol=0, atol=1e-5)
    # value_bias: the value columns take one bias row and the table's first 256 columns are not read
    vb = torch.randn(256, device='cuda', generator=g)
    tb = table.clone()
    tb[:, :256] = vb
    v2, samp2 = ops.gemm_bf16x3_encproj(a, wp, tb, ref, levels)
    tb[:, :256] = float('nan')
    v3, samp3 = ops.gemm_bf16x3_encproj(a, wp, tb, ref, levels, value_bias=vb)
    assert torch.equal(v2, v3) and torch.equal(samp2, samp3) and torch.equal(samp2, samp)


@pytest.mark.parametrize('N,H,W', [(2, 33, 47), (1, 64, 96), (3, 7, 5)])
def test_hrnet_stem_conv3x3s2_c3_vs_torch_fp64(N, H, W):
    """pave_conv3x3s2_c3_nchw_f32 (HRNet stem conv1 + folded BN + ReLU, hrnet.py:549-556: 3x3 / stride
    2 / pad 1, 3 -> 64 channels read from the NCHW batch) against torch.conv2d in fp64; odd sizes,
    borders, and a NaN pixel poisoning exactly the outputs whose window holds it."""
    from pavenet_amd.ops import conv3x3s2_c3_nchw
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(64, generator=g)
    taps = w.permute(1, 2, 3, 0).reshape(27, 64).contiguous()
    y = conv3x3s2_c3_nchw(x.cuda(), taps.cuda(), b.cuda(), relu=True)
    exp = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 2, 1))
    assert tuple(y.shape) == tuple(exp.shape) and y.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(y.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    y0 = conv3x3s2_c3_nchw(x.cuda(), taps.cuda(), None, relu=False)
    exp0 = torch.nn.functional.conv2d(x.double(), w.double(), None, 2, 1)
    np.testing.assert_allclose(y0.cpu().numpy(), exp0.numpy(), rtol=1e-5, atol=1e-5)
    xn = x.clone()
    xn[0, 1, H // 2, W // 2] = float('nan')
    bad = torch.isnan(conv3x3s2_c3_nchw(xn.cuda(), taps.cuda(), b.cuda())[0]).any(0).cpu()
    assert torch.equal(bad, torch.isnan(torch.nn.functional.conv2d(xn, w, b, 2, 1)[0]).any(0))


@pytest.mark.parametrize('M,K,res', [(70000, 256, True), (66001, 1024, True), (300, 256, False), (1200, 1024, True)])
def test_gemm_ln_wide_form_vs_8wave_form_and_fp64(M, K, res):
    """Linear + identity + LayerNorm in one launch (bricks/transformer.py:1316-1353): the wide form
    (LayerNorm statistics on the accumulator layout, diag variant 14) against the 8-wave form (13)
    -- same products, the LayerNorm sums in another order: <= 4e-6 -- and both against fp64;
    ragged M, in-place identity."""
    from pavenet_amd import native, ops
    from pavenet_amd.ops import gemm_bf16x3_ln, split_weight_bf16x3
    g = torch.Generator(device='cuda').manual_seed(M + K)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(256, K, device='cuda', generator=g) / K ** 0.5
    b, ga, be = (torch.randn(256, device='cuda', generator=g) for _ in range(3))
    r = torch.randn(M, 256, device='cuda', generator=g) if res else None
    wp = split_weight_bf16x3(w)
    # (one launch per call: 66 001 rows are 516 tiles, whose last 4 the wrapper would hand to the small-row forms --
    # ops.round_split_rows, test_round_split_of_a_nearly_empty_last_block_round)
    monkey = ops.ROUND_SPLIT
    ops.ROUND_SPLIT = False
    try:
        _gemm_ln_forms(M, K, res, a, w, wp, b, r, ga, be)
    finally:
        ops.ROUND_SPLIT = monkey


def _gemm_ln_forms(M, K, res, a, w, wp, b, r, ga, be):
    from pavenet_amd import native
    from pavenet_amd.ops import gemm_bf16x3_ln
    with native.diag_build(13):
        eight = gemm_bf16x3_ln(a, wp, b, r, ga, be, 1e-5).clone()
    with native.diag_build(14):
        wide = gemm_bf16x3_ln(a, wp, b, r, ga, be, 1e-5).clone()
        inplace = r.clone() if res else None
        if res:
            gemm_bf16x3_ln(a, wp, b, inplace, ga, be, 1e-5, out=inplace)
    default = gemm_bf16x3_ln(a, wp, b, r, ga, be, 1e-5)
    if M >= 512 * 128:          # the shipped library's choice: wide form from 512 row tiles on
        assert torch.equal(default, wide)
    elif M >= 8192:             # 8-wave form between 8192 rows and that; below: small-row GEMM + LayerNorm pass
        assert torch.equal(default, eight)
    if res:
        assert torch.equal(inplace, wide)
    np.testing.assert_allclose(wide.cpu().numpy(), eight.cpu().numpy(), rtol=0, atol=4e-6)
    x = a.double() @ w.double().t() + b.double() + (r.double() if res else 0)
    exp = torch.nn.functional.layer_norm(x, (256,), ga.double(), be.double(), 1e-5)
    np.testing.assert_allclose(wide.cpu().numpy(), exp.cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('seed', range(10))
def test_small_row_gemm_form_equals_tile_kernels_bit_for_bit(seed):
    """The small-row form of the split GEMM (a wave per 32-row tile, operands straight from L2, no LDS:
    the decoders' / heads' Linears, M < 8192) against the 128-row tile kernels (diag variant 8) on
    random shapes and epilogues -- same products in the same order: bit for bit -- incl. zero-padded
    planes (n_out), the row-periodic residual table, grouped columns, ragged M."""
    from pavenet_amd import native, ops
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))   # noqa: E731
    dev = 'cuda'
    M = ri(1, 1300)
    K = 32 * ri(2, 40)
    kind = ('plain', 'pad', 'ex', 'grouped')[seed % 4]
    rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)   # noqa: E731
    relu = bool(seed & 1)
    if kind == 'plain':
        N = 64 * ri(1, 8)            # (the small-row form takes outputs up to 512 columns)
        a, w, b, r = rnd(M, K), rnd(N, K, scale=0.05), rnd(N), rnd(M, N)
        wp = ops.split_weight_bf16x3(w, pad=True)
        run = lambda: ops.gemm_bf16x3(a, wp, b, r, relu=relu)                 # noqa: E731
    elif kind == 'pad':
        N = 4 * ri(1, 128)
        a, w, b = rnd(M, K), rnd(N, K, scale=0.05), rnd(N)
        wp = ops.split_weight_bf16x3(w, pad=True)
        run = lambda: ops.gemm_bf16x3(a, wp, b, None, relu=relu, n_out=N)     # noqa: E731
    elif kind == 'ex':
        N, rows = 128 * ri(1, 4), ri(1, 400)
        a, w, tab = rnd(M, K), rnd(N, K, scale=0.05), rnd(rows, N)
        wp = ops.split_weight_bf16x3(w, pad=True)
        run = lambda: ops.gemm_bf16x3_ex(a, wp, None, tab, residual_rows=rows)[0]   # noqa: E731
    else:
        G, gn = ri(2, 7), 64
        G = min(G, 8)             # G * 64 <= 512 columns
        a, w, b = rnd(M, G * K), rnd(G * gn, K, scale=0.05), rnd(G * gn)
        wp = ops.split_weight_bf16x3(w, pad=True)
        run = lambda: ops.gemm_bf16x3_grouped(a, wp, b, gn, relu=relu)        # noqa: E731
    with native.diag_build(19):          # the one-wave small-row form (the shipped selection until round 6)
        small = run().clone()
    with native.diag_build(8):
        tile = run().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(small).all()
    assert torch.equal(small, tile), (kind, M, K, float((small - tile).abs().max()))
    # the shipped selection (round 6): the K axis split over the four waves of a block, partial tiles added in a
    # fixed order -- the same fp32 terms in another association: equal to rounding, and bit-identical run to run
    ksplit = run().clone()
    again = run().clone()
    torch.cuda.synchronize()
    assert torch.equal(ksplit, again), 'the K-split small-row form is not deterministic'
    scale = float(tile.abs().max()) + 1e-6
    assert float((ksplit - tile).abs().max()) <= 4e-6 * scale * max(1.0, (K / 256) ** 0.5), \
        (kind, M, K, float((ksplit - tile).abs().max()), scale)


@pytest.mark.parametrize('M,K', [(1200, 256), (1200, 1024), (37, 64), (300, 512)])
def test_small_row_linear_layernorm_vs_fp64(M, K):
    """Linear + identity + LayerNorm at few rows: the small-row GEMM + the LayerNorm pass (the shipped
    selection below 8192 rows) against fp64 and against the one-launch 8-wave form (diag variant 13)."""
    from pavenet_amd import native
    from pavenet_amd.ops import gemm_bf16x3_ln, split_weight_bf16x3
    g = torch.Generator(device='cuda').manual_seed(M + K)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(256, K, device='cuda', generator=g) / K ** 0.5
    b, ga, be = (torch.randn(256, device='cuda', generator=g) for _ in range(3))
    r = torch.randn(M, 256, device='cuda', generator=g)
    wp = split_weight_bf16x3(w)
    got = gemm_bf16x3_ln(a, wp, b, r, ga, be, 1e-5)
    inplace = r.clone()
    gemm_bf16x3_ln(a, wp, b, inplace, ga, be, 1e-5, out=inplace)
    assert torch.equal(inplace, got)
    with native.diag_build(13):
        one = gemm_bf16x3_ln(a, wp, b, r, ga, be, 1e-5)
    np.testing.assert_allclose(got.cpu().numpy(), one.cpu().numpy(), rtol=0, atol=1e-5 if K >= 512 else 4e-6)
    x = a.double() @ w.double().t() + b.double() + r.double()
    exp = torch.nn.functional.layer_norm(x, (256,), ga.double(), be.double(), 1e-5)
    np.testing.assert_allclose(got.cpu().numpy(), exp.cpu().numpy(), rtol=2e-5, atol=2e-5)


def test_neck_1x1_level_with_96_channels_runs_the_split_gemm():
    """HRNet-w48's first neck level (ChannelMapper, channel_mapper.py:63-78, in_channels 96 -> 256,
    1x1, no bias): K = 96 is off the 64-grid of the tile kernels' fast shapes but K % 32 == 0, so
    the level takes the 3-plane kernel on zero-padded planes instead of a library convolution;
    checked against conv2d in fp64."""
    from pavenet_amd import bricks
    from pavenet_amd.necks import ChannelMapper
    g = torch.Generator().manual_seed(96)
    conv = torch.nn.Conv2d(96, 256, 1, bias=False).cuda()
    x = torch.randn(2, 96, 50, 84, generator=g).cuda()
    old = bricks.get_gemm_mode()
    try:
        bricks.set_gemm_mode('bf16x3')
        with torch.no_grad():
            y = ChannelMapper._conv_split(conv, x)
    finally:
        bricks.set_gemm_mode(old)
    assert y is not None, 'the K = 96 level fell back to the library'
    exp = torch.nn.functional.conv2d(x.double(), conv.weight.double())
    np.testing.assert_allclose(y.detach().cpu().numpy(), exp.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_round_split_of_a_nearly_empty_last_block_round():
    """ops.round_split_rows: 3 x 22 323 rows (a one-clip T = 3 step) are 524 row tiles for the chip's 512 block slots,
    so every wide / LayerNorm-epilogue launch of the encoder ran a second round for 12 tiles.  The rows of the full
    rounds and the remainder (1 433 rows, small-row forms) are launched separately: same values as the one-launch
    form up to the summation order of the form each part takes; the rule itself on a few sizes."""
    from pavenet_amd import ops
    assert ops.round_split_rows(3 * 22323, 1) == 512 * 128          # LayerNorm form: one tile per row tile
    assert ops.round_split_rows(3 * 22323, 4) == 512 * 128          # FFN1 (N = 1024): 4 tiles per row tile, 128 per round
    assert ops.round_split_rows(28 * 22323, 1) is None              # 4 884 tiles: the last round is half full
    assert ops.round_split_rows(3 * 20906, 1) is None               # the 750 x 1333 canvas: 490 tiles, one round
    assert ops.round_split_rows(512 * 128, 1) is None and ops.round_split_rows(300, 1) is None
    M = 3 * 22323
    g = torch.Generator(device='cuda').manual_seed(5)
    a = torch.randn(M, 256, device='cuda', generator=g)
    w1 = torch.randn(1024, 256, device='cuda', generator=g) / 16
    b1 = torch.randn(1024, device='cuda', generator=g)
    wp1 = ops.split_weight_bf16x3(w1)
    w2 = torch.randn(256, 1024, device='cuda', generator=g) / 32
    wp2 = ops.split_weight_bf16x3(w2)
    b2, ga, be = (torch.randn(256, device='cuda', generator=g) for _ in range(3))
    res = torch.randn(M, 256, device='cuda', generator=g)
    try:
        ops.ROUND_SPLIT = False
        h_one = ops.gemm_bf16x3(a, wp1, b1, relu=True)
        o_one = ops.gemm_bf16x3_ln(h_one, wp2, b2, res, ga, be, 1e-5)
    finally:
        ops.ROUND_SPLIT = True
    h_two = ops.gemm_bf16x3(a, wp1, b1, relu=True)
    o_two = ops.gemm_bf16x3_ln(h_one, wp2, b2, res, ga, be, 1e-5)
    inplace = res.clone()
    assert ops.gemm_bf16x3_ln(h_one, wp2, b2, inplace, ga, be, 1e-5, out=inplace).data_ptr() == inplace.data_ptr()
    torch.cuda.synchronize()
    M1 = 512 * 128
    assert torch.equal(h_two[:M1], h_one[:M1]) and torch.equal(o_two[:M1], o_one[:M1])     # the full rounds: same launch form
    np.testing.assert_allclose(h_two[M1:].cpu().numpy(), h_one[M1:].cpu().numpy(), rtol=0, atol=4e-6 * float(h_one.abs().max()))
    np.testing.assert_allclose(o_two[M1:].cpu().numpy(), o_one[M1:].cpu().numpy(), rtol=0, atol=2e-5)
    assert torch.equal(inplace, o_two)
    exp = torch.nn.functional.layer_norm(h_one[M1:].double() @ w2.double().t() + b2.double() + res[M1:].double(), (256,),
                                         ga.double(), be.double(), 1e-5)
    np.testing.assert_allclose(o_two[M1:].cpu().numpy(), exp.cpu().numpy(), rtol=2e-5, atol=2e-5)
