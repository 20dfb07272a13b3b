"""Stress configuration of BASELINE.json (configs[4]): the reference's ``NodeEdgeBlock.forward``
(src/ConStruct/models/transformer_model.py:165-248) on a bf16 edge tensor at de = 128, n up to 1024, as ONE fused HIP kernel
over the n^2 pairs (``scgg_edge_block_bf16``) plus the row-wise pieces around it.

``NodeEdgeBlockBF16(block)`` wraps the parameter holder of a ``NodeEdgeBlock`` (this package's or the reference's: same
attribute names) and returns ``(newX, newE)``: newE as a bfloat16 tensor, newX fp32.  The y update (:239-246) is not part of
it (it pools the fp32 tensors: the layer kernels' job).  The node-sized Linears (q | k | v, the four FiLM vectors, x_out) run
on the library's row-list Linear (``scgg_linear_rows``: fp32 results from three-term bf16 products, valid rows only).
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
        dy = block.y_e_add.in_features
        L = _lib.lib()
        self._lin = {}                                     # name -> (N, K, packed image)
        for name, N, K in (("qkv", 3 * self.dx, self.dx), ("film", 4 * self.dx, dy), ("x_out", self.dx, self.dx)):
            nb = int(L.scgg_linear_rows_image_bytes(N, K))
            if nb == 0:
                raise _lib.ScggError(f"linear_rows: K={K} is not built (32, 64, 128, 256 are)")
            self._lin[name] = (N, K, torch.empty(nb, dtype=torch.uint8, device=dev))
        self.refresh()

    def refresh(self):
        """Re-pack the weight image after the parameters changed."""
        b = self.block
        f = lambda t: t.detach().float().contiguous()
        self._keep = [f(b.e_mul.weight), f(b.e_mul.bias), f(b.e_add.weight), f(b.e_add.bias), f(b.e_out.weight)]
        _lib.check(_lib.lib().scgg_edge_block_bf16_pack(*[_lib.ptr(t) for t in self._keep], self.dx, self.de, _lib.ptr(self.image),
                                                        _lib.stream_ptr()), "edge_block_bf16_pack")
        one = lambda lin: f(lin.bias) + 1.0                # the "+ 1" of the multiplicative FiLM vectors folded into the bias
        packs = {"qkv": (torch.cat([f(b.q.weight), f(b.k.weight), f(b.v.weight)]), torch.cat([f(b.q.bias), f(b.k.bias), f(b.v.bias)])),
                 "film": (torch.cat([f(b.y_e_add.weight), f(b.y_e_mul.weight), f(b.y_x_add.weight), f(b.y_x_mul.weight)]),
                          torch.cat([f(b.y_e_add.bias), one(b.y_e_mul), f(b.y_x_add.bias), one(b.y_x_mul)])),
                 "x_out": (f(b.x_out.weight), f(b.x_out.bias))}
        for name, (W, bias) in packs.items():
            N, K, img = self._lin[name]
            W, bias = W.contiguous(), bias.contiguous()
            self._keep += [W, bias]
            _lib.check(_lib.lib().scgg_linear_rows_pack(_lib.ptr(W), _lib.ptr(bias), N, K, _lib.ptr(img), _lib.stream_ptr()),
                       "linear_rows_pack")

    def _linear(self, name, A, out, rows, row_count, max_rows):
        N, K, img = self._lin[name]
        _lib.check(_lib.lib().scgg_linear_rows(_lib.ptr(A), A.shape[-1], _lib.ptr(img), N, K, _lib.ptr(out), N, _lib.ptr(rows),
                                               _lib.ptr(row_count), max_rows, 0, _lib.stream_ptr()), "linear_rows")
        return out

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
        X, y = X.float().contiguous(), y.float().contiguous()
        rows, row_count, units, unit_count = lists if lists is not None else self.work_lists(m)
        n_rows = max(int(rows.numel()), 1)
        # q | k | v of the valid rows (the rows of masked nodes stay 0 = the reference's mask, :181-183), FiLM vectors of every graph
        qkv = self._linear("qkv", X.view(B * n, -1), torch.zeros(B * n, 3 * self.dx, device=X.device), rows, row_count, n_rows)
        if getattr(self, "_grows", None) is None or self._grows[0].numel() != B:
            self._grows = (torch.arange(B, dtype=torch.int32, device=X.device), torch.tensor([B], dtype=torch.int32, device=X.device))
        film = self._linear("film", y, torch.empty(B, 4 * self.dx, device=X.device), self._grows[0], self._grows[1], B)
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
        newX = self._linear("x_out", pre.view(B * n, dx), torch.zeros(B * n, dx, device=X.device), rows, row_count, n_rows)
        return newX.view(B, n, dx), newE
