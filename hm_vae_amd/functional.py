"""torch.autograd glue for using the HIP operators as free-standing modules (torch layout (B, C, T) in/out).

The fused engine (engine.py) does not go through here; this file exists so that `SkeletonConv`, `SkeletonPool`,
`SkeletonUnpool`, `ForwardKinematicsLayer` and `rotation_matrix_from_ortho6d` keep working as composable,
differentiable building blocks like their reference namesakes.
"""
import numpy as np
import torch

from . import _lib, ops


def _to_soa(x):
    """(B, C, T) -> [C*T, bp]"""
    B, Cn, T = x.shape
    bp = ops.pad_batch(B)
    out = torch.empty(Cn * T, bp, device=x.device, dtype=torch.float32)
    ops.aos_to_soa(x.contiguous().float(), out, B, 1, Cn * T, bp)
    return out, bp


def _from_soa(s, B, Cn, T):
    out = torch.empty(B, Cn * T, device=s.device, dtype=torch.float32)
    ops.soa_to_aos(s, out, B, 1, Cn * T, s.shape[1])
    return out.reshape(B, Cn, T)


class SkelConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, op):
        _lib.require_gpu()
        B = x.shape[0]
        cout, cin, kt = op.w_shape
        op.pack(weight.detach().contiguous())
        xs, bp = _to_soa(x.detach())
        ys = torch.empty(cout * op.t_out, bp, device=x.device)
        op.forward(xs, ys, bp, bias=None if bias is None else bias.detach().contiguous())
        ctx.op, ctx.xs, ctx.bp, ctx.B, ctx.has_bias = op, xs, bp, B, bias is not None
        return _from_soa(ys, B, cout, op.t_out)

    @staticmethod
    def backward(ctx, gy):
        op, xs, bp, B = ctx.op, ctx.xs, ctx.bp, ctx.B
        cout, cin, kt = op.w_shape
        gys, _ = _to_soa(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if op.bwd is None:
                raise _lib.HmvaeError("this SkeletonConv was planned without a data gradient")
            dxs = torch.empty(cin * op.t_in, bp, device=gy.device)
            op.backward_data(gys, dxs, bp)
            gx = _from_soa(dxs, B, cin, op.t_in)
        if ctx.needs_input_grad[1]:
            gw = torch.zeros(cout, cin, kt, device=gy.device)
            slab = torch.empty(op.slab_floats(bp), device=gy.device)
            op.backward_weight(gys, xs, gw, bp, slab)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = torch.empty(cout, device=gy.device)
            rows = (torch.arange(cout, dtype=torch.int32) * op.t_out).to(gy.device)
            ops.row_sums(gys, gb, rows, cout, op.t_out, 1, bp)
        return gx, gw, gb, None


class PoolFn(torch.autograd.Function):
    """SkeletonPool (mean); `slope` = 1 gives the plain pool, 0.2 fuses the LeakyReLU that follows it."""

    @staticmethod
    def forward(ctx, x, op, slope):
        B, Cn, T = x.shape
        xs, bp = _to_soa(x.detach())
        ys = torch.empty(op.n_out * op.ch * T, bp, device=x.device)
        op.forward(xs, ys, T, bp, slope)
        ctx.op, ctx.ys, ctx.geom, ctx.slope = op, ys, (B, T, bp), slope
        return _from_soa(ys, B, op.n_out * op.ch, T)

    @staticmethod
    def backward(ctx, gy):
        op, (B, T, bp) = ctx.op, ctx.geom
        gys, _ = _to_soa(gy)
        dxs = torch.empty(op.n_in * op.ch * T, bp, device=gy.device)
        op.backward(gys, ctx.ys, dxs, T, bp, ctx.slope)
        return _from_soa(dxs, B, op.n_in * op.ch, T), None, None


class UnpoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        B, Cn, T = x.shape
        xs, bp = _to_soa(x.detach())
        T2 = 2 * T if op.up else T
        ys = torch.empty(op.n_out * op.ch * T2, bp, device=x.device)
        op.forward(xs, ys, T, bp)
        ctx.op, ctx.geom = op, (B, T, T2, bp)
        return _from_soa(ys, B, op.n_out * op.ch, T2)

    @staticmethod
    def backward(ctx, gy):
        op, (B, T, T2, bp) = ctx.op, ctx.geom
        gys, _ = _to_soa(gy)
        dxs = torch.empty(op.n_in * op.ch * T, bp, device=gy.device)
        op.backward(gys, dxs, T, bp)
        return _from_soa(dxs, B, op.n_in * op.ch, T), None


class FkFn(torch.autograd.Function):
    """rot (N,J,6) or (N,J,3,3) -> (R (N,J,3,3), pos (N,J,3)); offsets (J,3) shared or (N,J,3) per frame."""

    @staticmethod
    def forward(ctx, rot, offsets, parents, want_R):
        _lib.require_gpu()
        n, J = rot.shape[0], rot.shape[1]
        rd = 6 if rot.shape[-1] == 6 else 9
        r = rot.detach().reshape(n, J, rd).contiguous().float()
        off = offsets.detach().contiguous().float()
        stride = 0 if off.dim() == 2 else J * 3
        R = torch.empty(n, J, 9, device=rot.device) if want_R else None
        P = torch.empty(n, J, 3, device=rot.device)
        ops.fk_fwd(r, rd, off, stride, parents, n, R, P)
        ctx.save = (r, rd, off, stride, tuple(parents), rot.shape)
        ctx.want_R = want_R
        if want_R:
            return R.reshape(n, J, 3, 3), P
        return P.new_zeros(0), P

    @staticmethod
    def backward(ctx, gR, gP):
        r, rd, off, stride, parents, shape = ctx.save
        n, J = r.shape[0], r.shape[1]
        dR = gR.reshape(n, J, 9).contiguous().float() if ctx.want_R and gR is not None and gR.numel() else None
        dP = gP.contiguous().float() if gP is not None else None
        d = torch.empty(n, J, rd, device=r.device)
        ops.fk_bwd(r, rd, off, stride, list(parents), n, dR, dP, d)
        return d.reshape(shape), None, None, None
