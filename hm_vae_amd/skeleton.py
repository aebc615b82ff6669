"""Skeleton-aware layers with the reference's class surface (skeleton.py), computed by HIP kernels.

Same constructor signatures, attribute names, parameter names and initialisation distributions as the
reference classes, so state_dicts interchange.  The arithmetic is NOT torch's: SkeletonConv runs the fp32-MFMA
gather-GEMM on neighbour-channel lists (no `weight * mask` tensor, no padded copy, no dense conv), SkeletonPool /
SkeletonUnpool are index gathers (no dense constant-matrix matmul).  Inputs must live on a HIP device.
"""
import math

import torch
import torch.nn as nn

from . import ops
from . import functional as Fn
from .topology import (get_edges, calc_edge_mat, find_neighbor, derive_pooling)  # noqa: F401  (re-exported)


def _norm_pad_mode(padding_mode):
    if padding_mode in ("zeros", "constant"):
        return "constant"
    if padding_mode in ("reflection", "reflect"):
        return "reflect"
    raise ValueError("unsupported padding_mode %r" % (padding_mode,))


class SkeletonConv(nn.Module):
    """reference: skeleton.py:9-105"""

    def __init__(self, neighbour_list, in_channels, out_channels, kernel_size, joint_num, stride=1, padding=0,
                 bias=True, padding_mode='zeros', add_offset=False, in_offset_channel=0):
        if in_channels % joint_num != 0 or out_channels % joint_num != 0:
            raise Exception('BAD')
        super().__init__()
        if add_offset:
            raise NotImplementedError("add_offset=True is unreachable from every shipped config (SURVEY.md §2 row 7)")
        self.in_channels_per_joint = in_channels // joint_num
        self.out_channels_per_joint = out_channels // joint_num
        self.neighbour_list = neighbour_list
        self.joint_num = joint_num
        self.add_offset = False
        self.stride, self.dilation, self.groups = stride, 1, 1
        self.padding = padding
        self.padding_mode = _norm_pad_mode(padding_mode)
        self._padding_repeated_twice = (padding, padding)
        self.kernel_size = kernel_size
        self.in_channels, self.out_channels = in_channels, out_channels
        self.expanded_neighbour_list = [[k * self.in_channels_per_joint + i for k in nb
                                         for i in range(self.in_channels_per_joint)] for nb in neighbour_list]
        weight = torch.zeros(out_channels, in_channels, kernel_size)
        mask = torch.zeros_like(weight)
        b = torch.zeros(out_channels) if bias else None
        co = self.out_channels_per_joint
        for i, cols in enumerate(self.expanded_neighbour_list):
            mask[co * i:co * (i + 1), cols, :] = 1
            # per-block kaiming_uniform(a=sqrt(5)) on the neighbour slice only (skeleton.py:70-85)
            blk = torch.empty(co, len(cols), kernel_size)
            nn.init.kaiming_uniform_(blk, a=math.sqrt(5))
            weight[co * i:co * (i + 1), cols, :] = blk
            if b is not None:
                bound = 1.0 / math.sqrt(len(cols) * kernel_size)
                b[co * i:co * (i + 1)] = torch.empty(co).uniform_(-bound, bound)
        self.mask = nn.Parameter(mask, requires_grad=False)
        self.weight = nn.Parameter(weight)
        if b is not None:
            self.bias = nn.Parameter(b)
        else:
            self.register_parameter('bias', None)
        self.description = 'SkeletonConv(in_channels_per_armature={}, out_channels_per_armature={}, kernel_size={}, ' \
                           'joint_num={}, stride={}, padding={}, bias={})'.format(
            in_channels // joint_num, out_channels // joint_num, kernel_size, joint_num, stride, padding, bias)
        self._ops = {}

    def _op(self, t_in, device):
        key = (t_in, str(device))
        if key not in self._ops:
            self._ops[key] = ops.conv_gemm(device, self.neighbour_list, self.in_channels, self.out_channels,
                                           self.kernel_size, self.stride, self.padding, self.padding_mode, t_in)
        return self._ops[key]

    def forward(self, input):
        return Fn.SkelConvFn.apply(input, self.weight, self.bias, self._op(input.shape[2], input.device))


class SkeletonLinear(nn.Module):
    """reference: skeleton.py:108-156.  Only reachable with add_offset=True, which no config sets; the name stays
    importable, the layer itself is out of scope."""

    def __init__(self, neighbour_list, in_channels, out_channels, extra_dim1=False):
        super().__init__()
        raise NotImplementedError("SkeletonLinear is dead code in the reference (add_offset is always False)")


class SkeletonPool(nn.Module):
    """reference: skeleton.py:159-231"""

    def __init__(self, edges, pooling_mode, channels_per_edge, last_pool=False):
        super().__init__()
        if pooling_mode != 'mean':
            raise Exception('Unimplemented pooling mode in matrix_implementation')
        self.channels_per_edge = channels_per_edge
        self.pooling_mode = pooling_mode
        self.edge_num = len(edges)
        self.seq_list, self.pooling_list, self.new_edges = derive_pooling(edges, last_pool)
        self.description = 'SkeletonPool(in_edge_num={}, out_edge_num={})'.format(len(edges), len(self.pooling_list))
        w = torch.zeros(len(self.pooling_list) * channels_per_edge, self.edge_num * channels_per_edge)
        for i, pair in enumerate(self.pooling_list):
            for j in pair:
                for c in range(channels_per_edge):
                    w[i * channels_per_edge + c, j * channels_per_edge + c] = 1.0 / len(pair)
        # kept only because the reference serialises it (checkpoint key ...1.weight); the kernel uses index lists
        self.weight = nn.Parameter(w, requires_grad=False)
        self._ops = {}

    def _op(self, device):
        key = str(device)
        if key not in self._ops:
            self._ops[key] = ops.PoolOp(device, self.pooling_list, self.edge_num, self.channels_per_edge)
        return self._ops[key]

    def forward(self, input):
        return Fn.PoolFn.apply(input, self._op(input.device), 1.0)


class SkeletonUnpool(nn.Module):
    """reference: skeleton.py:234-261"""

    def __init__(self, pooling_list, channels_per_edge):
        super().__init__()
        self.pooling_list = pooling_list
        self.input_edge_num = len(pooling_list)
        self.output_edge_num = sum(len(t) for t in pooling_list)
        self.channels_per_edge = channels_per_edge
        self.description = 'SkeletonUnpool(in_edge_num={}, out_edge_num={})'.format(self.input_edge_num,
                                                                                  self.output_edge_num)
        w = torch.zeros(self.output_edge_num * channels_per_edge, self.input_edge_num * channels_per_edge)
        for i, pair in enumerate(self.pooling_list):
            for j in pair:
                for c in range(channels_per_edge):
                    w[j * channels_per_edge + c, i * channels_per_edge + c] = 1
        self.weight = nn.Parameter(w, requires_grad=False)
        self._ops = {}

    def _op(self, device):
        key = str(device)
        if key not in self._ops:
            self._ops[key] = ops.UnpoolOp(device, self.pooling_list, self.channels_per_edge, up=False)
        return self._ops[key]

    def forward(self, input):
        return Fn.UnpoolFn.apply(input, self._op(input.device))
