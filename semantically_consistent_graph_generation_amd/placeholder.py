"""Dense batch container of the hot path: same fields and masking rules as the reference's
``PlaceHolder`` (/root/reference/src/ConStruct/utils.py:110-235).  Pure tensor plumbing (torch ops on whatever
device the tensors live on); the O(n^2) masking inside the denoiser / sampler is done by the HIP kernels."""
from __future__ import annotations

import torch


class PlaceHolder:
    def __init__(self, X, E, y, charges=None, t_int=None, t=None, node_mask=None, pos=None, radius=None,
                 extra_info=None):
        self.X = X
        self.charges = charges
        self.E = E
        self.y = y
        self.t_int = t_int
        self.t = t
        self.node_mask = node_mask
        self.pos = pos
        self.radius = radius
        self.extra_info = extra_info

    def device_as(self, x: torch.Tensor):
        """utils.py:123-132."""
        for f in ("X", "charges", "E", "y", "pos", "radius", "extra_info"):
            v = getattr(self, f)
            if v is not None:
                setattr(self, f, v.to(x.device))
        return self

    def mask(self, node_mask=None, check_symmetry: bool = False):
        """utils.py:134-167: X*m, E*m_i*m_j*(1-I), radius/extra_info likewise, pos masked and re-centred over ALL
        n slots (padding included, quirk 6).  The reference's symmetric ``assert`` (a host sync) is opt-in here."""
        if node_mask is None:
            assert self.node_mask is not None
            node_mask = self.node_mask
        n = node_mask.shape[1]
        xm = node_mask.unsqueeze(-1)
        pm = (xm.unsqueeze(2) * xm.unsqueeze(1)) * (~torch.eye(n, dtype=torch.bool, device=node_mask.device))[None, :, :, None]
        if self.X is not None:
            self.X = self.X * xm
        if self.charges is not None and self.charges.numel() > 0:
            self.charges = self.charges * xm
        if self.E is not None:
            self.E = self.E * pm
            if check_symmetry:
                assert torch.allclose(self.E, self.E.transpose(1, 2))
        if self.radius is not None:
            self.radius = self.radius * pm
        if self.pos is not None:
            self.pos = self.pos * xm
            self.pos = self.pos - self.pos.mean(dim=1, keepdim=True)
        if self.extra_info is not None:
            self.extra_info = self.extra_info * pm
        return self

    def collapse(self, collapse_charges=None):
        """utils.py:169-195: argmax labels; padding -> X -1, charges 1000, E -1."""
        out = self.copy()
        out.X = torch.argmax(self.X, dim=-1)
        if self.charges is not None and self.charges.numel() > 0:
            out.charges = collapse_charges.to(self.charges.device)[torch.argmax(self.charges, dim=-1)]
        else:
            out.charges = self.X.new_zeros(self.X.shape[:-1])
        out.E = torch.argmax(self.E, dim=-1)
        m = self.node_mask
        out.X[~m] = -1
        out.charges[~m] = 1000
        out.E[~(m[:, :, None] & m[:, None, :])] = -1
        out.radius = self.radius.squeeze(-1) if self.radius is not None else None
        return out

    def split(self):
        """utils.py:197-211."""
        graphs = []
        for i in range(self.X.shape[0]):
            k = int(self.node_mask[i].sum())
            graphs.append(PlaceHolder(
                X=self.X[i, :k], charges=self.charges[i, :k] if self.charges is not None else None,
                E=self.E[i, :k, :k], y=self.y[i] if self.y is not None else None, node_mask=None,
                pos=self.pos[i, :k] if self.pos is not None else None,
                radius=self.radius[i, :k, :k] if self.radius is not None else None))
        return graphs

    def copy(self):
        """utils.py:223-234 (extra_info is dropped, as in the reference)."""
        return PlaceHolder(X=self.X, charges=self.charges, E=self.E, y=self.y, t_int=self.t_int, t=self.t,
                           node_mask=self.node_mask, pos=self.pos, radius=self.radius)

    def __repr__(self):
        def s(v):
            return tuple(v.shape) if isinstance(v, torch.Tensor) else v
        return (f"X: {s(self.X)} -- charges: {s(self.charges)} -- E: {s(self.E)} -- y: {s(self.y)} -- "
                f"pos: {s(self.pos)} -- radius: {s(self.radius)}")
