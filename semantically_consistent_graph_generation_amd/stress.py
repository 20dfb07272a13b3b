"""Stress configuration of BASELINE.json (configs[4]): the reference's ``NodeEdgeBlock.forward``
(src/ConStruct/models/transformer_model.py:165-248) on a bf16 edge tensor at de = 128, n up to 1024, as ONE fused HIP kernel
over the n^2 pairs (``scgg_edge_block_bf16``) plus the row-wise pieces around it.

``NodeEdgeBlockBF16(block)`` wraps the parameter holder of a ``NodeEdgeBlock`` (this package's or the reference's: same
attribute names) and returns ``(newX, newE)``: newE as a bfloat16 tensor, newX fp32.  The y update (:239-246) is not part of
it (it pools the fp32 tensors: the layer kernels' job).  The node-sized Linears (q, k, v, the four FiLM vectors, x_out) are
device GEMMs through torch (rocBLAS): O(n) work beside the O(n^2) kernel this configuration is about.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib


class NodeEdgeBlockBF16:
    def __init__(self, block):
        self.block = block
        self.dx, self.de, self.n_head = block.q.in_features, block.e_mul.in_features, block.n_head
        dev = block.q.weight.device
        if dev.type != "cuda":
            raise _lib.ScggError("the bf16 NodeEdgeBlock runs on the MI355X only; move the module to the GPU")
        nbytes = int(_lib.lib().scgg_edge_block_bf16_image_bytes(self.dx, self.de))
        if nbytes == 0:
            raise _lib.ScggError(f"edge_block_bf16: dims dx={self.dx} de={self.de} are not built (256/128, 256/64 and 128/64 are)")
        self.image = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.refresh()

    def refresh(self):
        """Re-pack the weight image after the parameters changed."""
        b = self.block
        f = lambda t: t.detach().float().contiguous()
        self._keep = [f(b.e_mul.weight), f(b.e_mul.bias), f(b.e_add.weight), f(b.e_add.bias), f(b.e_out.weight)]
        _lib.check(_lib.lib().scgg_edge_block_bf16_pack(*[_lib.ptr(t) for t in self._keep], self.dx, self.de, _lib.ptr(self.image),
                                                        _lib.stream_ptr()), "edge_block_bf16_pack")

    @staticmethod
    def work_lists(node_mask):
        """(rows, row_count, units, unit_count): flat indices of the valid rows and of the (valid row, 16-key chunk holding a
        valid key) units, as device int32 tensors (built with tensor ops: this path is a benchmark, not the sampling loop)."""
        B, n = node_mask.shape
        nch = (n + 15) // 16
        m = node_mask.bool()
        pad = nch * 16 - n
        chunk_any = F.pad(m, (0, pad)).view(B, nch, 16).any(dim=2)                    # [B, nch]
        unit_ok = m[:, :, None] & chunk_any[:, None, :]                               # [B, n, nch]
        units = unit_ok.reshape(-1).nonzero().flatten().to(torch.int32)
        rows = m.reshape(-1).nonzero().flatten().to(torch.int32)
        dev = node_mask.device
        return (rows, torch.tensor([rows.numel()], dtype=torch.int32, device=dev), units,
                torch.tensor([units.numel()], dtype=torch.int32, device=dev))

    @torch.no_grad()
    def __call__(self, X, E_bf16, y, node_mask, lists=None, out=None, attn=None):
        """X [B,n,dx] fp32, E_bf16 [B,n,n,de] bfloat16, y [B,dy] fp32 -> (newX fp32 [B,n,dx], newE bfloat16 [B,n,n,de])."""
        b = self.block
        B, n = node_mask.shape
        if E_bf16.dtype != torch.bfloat16 or not E_bf16.is_contiguous():
            raise _lib.ScggError("E must be a contiguous bfloat16 tensor")
        m = node_mask.bool()
        xm = m[:, :, None].float()
        qkv = torch.cat([F.linear(X, b.q.weight, b.q.bias) * xm, F.linear(X, b.k.weight, b.k.bias) * xm,
                         F.linear(X, b.v.weight, b.v.bias) * xm], dim=-1).contiguous()
        film = torch.cat([F.linear(y, b.y_e_add.weight, b.y_e_add.bias), F.linear(y, b.y_e_mul.weight, b.y_e_mul.bias) + 1.0,
                          F.linear(y, b.y_x_add.weight, b.y_x_add.bias), F.linear(y, b.y_x_mul.weight, b.y_x_mul.bias) + 1.0],
                         dim=-1).contiguous()
        rows, row_count, units, unit_count = lists if lists is not None else self.work_lists(m)
        nch = (n + 15) // 16
        newE = out if out is not None else torch.zeros(B, n, n, self.de, dtype=torch.bfloat16, device=X.device)
        attn = attn if attn is not None else torch.empty(B * n * nch, 3, self.dx, device=X.device)
        b_out = b.e_out.bias.detach().float().contiguous()
        dx = self.dx
        ev = getattr(self, "kernel_events", None)          # (start, end) torch events: the bench times the pair kernel alone
        if ev is not None:
            ev[0].record()
        _lib.check(_lib.lib().scgg_edge_block_bf16(
            _lib.ptr(E_bf16), _lib.ptr(newE), _lib.ptr(m.view(torch.uint8)), _lib.ptr(qkv), C.c_void_p(qkv.data_ptr() + 4 * dx),
            C.c_void_p(qkv.data_ptr() + 8 * dx), 3 * dx, _lib.ptr(film), _lib.ptr(self.image), _lib.ptr(b_out), _lib.ptr(attn), _lib.ptr(units),
            _lib.ptr(unit_count), B, n, dx, self.de, self.n_head, _lib.stream_ptr()), "edge_block_bf16")
        if ev is not None:
            ev[1].record()
        pre = torch.zeros(B, n, dx, device=X.device)
        _lib.check(_lib.lib().scgg_attn_merge(_lib.ptr(attn), _lib.ptr(m.view(torch.uint8)), _lib.ptr(film), _lib.ptr(pre), _lib.ptr(rows),
                                              _lib.ptr(row_count), max(int(rows.numel()), 1), n, dx, _lib.stream_ptr()), "attn_merge")
        newX = F.linear(pre, b.x_out.weight, b.x_out.bias) * xm
        return newX, newE
