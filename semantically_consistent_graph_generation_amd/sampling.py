"""The reverse-sampling driver of the hot path: ``forward`` (reference diffusion_model_discrete_3d.py:184-201),
``sample_zs_from_zt`` (:745-751) and the T-step loop of ``sample_batch`` (:583-684), without the Lightning /
visualisation glue around them.

Two equivalent routes through the same kernels:
  * PlaceHolder route (``forward`` / ``sample_zs_from_zt``): reference types in, reference types out (one-hot float E,
    int64 extra_info) -- what an unmodified caller or a host-side projector sees;
  * state route (``sample_batch`` with no projector): the graph state stays int8 on the device, the denoiser input
    ``[one-hot | extra features]`` is assembled by the feature kernel straight into a persistent buffer, and no step
    synchronises with the host (the reference forces ~30 syncs per step).
Graph chains are independent: a multi-GPU run shards the list of graphs and passes ``graph_id_base`` so that every
graph draws the same Philox stream whatever the sharding.
"""
from __future__ import annotations

import math
import os

import torch

from . import _lib
from .extra_features import DummyExtraFeatures, ExtraFeatures
from .noise_model import NoiseModel, Rng
from .placeholder import PlaceHolder
from .transformer_model import model_input


class DiscreteSampler:
    def __init__(self, model, noise_model: NoiseModel, extra_features=None, faster_sampling: int = 1):
        self.model = model
        self.noise_model = noise_model
        self.extra_features = extra_features if extra_features is not None else DummyExtraFeatures()
        self.T = noise_model.T
        self.faster_sampling = faster_sampling
        # position-aware denoisers (pos_transformer_model.py) also read z_t.pos and, for the radius head, z_t.radius
        self.is3d = hasattr(model, "mlp_in_pos_high_dim")

    # ---------------------------------------------------------------------------------------- PlaceHolder route
    def forward(self, z_t):
        """reference :184-201: concatenate [z_t | extra features | t] and run the denoiser."""
        extra = self.extra_features(z_t)
        X = torch.cat((z_t.X, z_t.charges, extra.X), dim=2).float()
        E = torch.cat((z_t.E, extra.E), dim=3).float()
        y = torch.hstack((z_t.y, extra.y, z_t.t)).float()
        inp = model_input(X, E, y, z_t.node_mask)
        if self.is3d:
            inp.pos, inp.radius = z_t.pos, z_t.radius
        return self.model(inp)

    def sample_zs_from_zt(self, z_t, s_int, rng: Rng = None):
        """reference :745-751."""
        pred = self.forward(z_t)
        return self.noise_model.sample_zs_from_zt_and_pred(z_t=z_t, pred=pred, s_int=s_int, rng=rng)

    # --------------------------------------------------------------------------------------------- state route
    def _alloc_state(self, node_mask):
        B, n = node_mask.shape
        dev = node_mask.device
        nm, m = self.noise_model, self.model
        C_e = nm.E_classes
        top_k = nm.top_k or 0
        st = {
            "E_in": torch.zeros(B, n, n, m.in_dim_E, device=dev),
            "X_in": torch.zeros(B, n, m.in_dim_X, device=dev),
            "y_in": torch.zeros(B, m.in_dim_y, device=dev),
            "labels": [torch.empty(B, n, n, dtype=torch.int8, device=dev) for _ in range(2)],
            "cands": [torch.empty(B, n, n, max(top_k, 1), dtype=torch.int8, device=dev) for _ in range(2)],
            "t_all": torch.arange(self.T + 1, dtype=torch.int32, device=dev)[:, None].expand(-1, B).contiguous(),
        }
        st["X_in"][..., 0] = node_mask.float()            # single node class: masked one-hot
        if isinstance(self.extra_features, ExtraFeatures):
            st["y_in"][:, 0:1] = node_mask.sum(dim=1, keepdim=True) / self.extra_features.max_n_nodes
        return st

    def _build_inputs(self, st, labels, node_mask, t):
        """Assemble the denoiser input of step t from the int8 state (reference forward :190-199)."""
        nm = self.noise_model
        B, n = node_mask.shape
        C_e = nm.E_classes
        ef = self.extra_features
        if isinstance(ef, ExtraFeatures):
            # X_in = [node one-hot | 3 cycle feats | LCC flag + eigenvectors], y_in = [n/n_max | graph feats |
            # eigenvalue feats | t], E_in = [edge one-hot | reachability + Adamic-Adar]: two kernels, no copies
            ky = 4 + ef.num_degree + 2 + 1 + C_e
            # The eigen features feed only X and y (extra_features.py:41-60): Laplacian first, then the eigensolver on the
            # denoiser's second stream (ahead of its consumers of X / y: scgg_denoiser_side_stream) while the caller's
            # stream computes the rest of the structural features and the denoiser's edge input MLP.
            if os.environ.get("SCGG_NO_SIDE_STREAM") or not hasattr(self.model, "side_stream"):   # A/B switch; 3D engine: one stream
                _, _, _, lap = ef.adjacency_features(labels, node_mask, C_e, e_dst=st["E_in"], e_off=C_e, want_onehot=True,
                                                     x_dst=st["X_in"], x_off=1, y_dst=st["y_in"], y_off=1)
                ef.eigen_features(lap, node_mask, evl_dst=st["y_in"], evl_off=1 + ky, evc_dst=st["X_in"], evc_off=4)
            else:
                lap = ef.laplacian(labels, node_mask)
                side = st.get("side_stream")
                if side is None:
                    side = st["side_stream"] = self.model.side_stream(node_mask.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    lap.record_stream(side)
                    ef.eigen_features(lap, node_mask, evl_dst=st["y_in"], evl_off=1 + ky, evc_dst=st["X_in"], evc_off=4)
                ef.adjacency_features(labels, node_mask, C_e, e_dst=st["E_in"], e_off=C_e, want_onehot=True, want_lap=False,
                                      x_dst=st["X_in"], x_off=1, y_dst=st["y_in"], y_off=1)
            st["y_in"][:, -1] = t / self.T
        else:
            _lib.check(_lib.lib().scgg_labels_to_onehot(_lib.ptr(labels), None, B * n * n, C_e, 0, _lib.ptr(st["E_in"]), None,
                                                        _lib.stream_ptr()), "labels_to_onehot")
            st["y_in"][:, -1] = t / self.T
        return model_input(st["X_in"], st["E_in"], st["y_in"], node_mask)

    def sample_batch(self, n_nodes, seed: int = 0, graph_id_base: int = 0, projector=None, n_max=None,
                     steps=None, return_state=False, keep_chain: int = 0, number_chain_steps: int = 50):
        """reference sample_batch :583-684 (loop :653-663, collapse :684) for one batch of graphs.

        n_nodes: list / tensor of node counts.  Returns the collapsed PlaceHolder (X, E integer labels, -1 padding,
        pos) -- or the raw int8 state with ``return_state``.  ``steps`` limits the number of reverse steps (tests).
        ``projector``: a class / factory ``projector(z_T, device)`` with the reference's interface (projector.py, or the
        reference's own classes).  ``keep_chain`` > 0: the first ``keep_chain`` graphs are also recorded along the chain
        as the reference does (:622-631, :665-677, :689-699): ``number_chain_steps`` frames of collapsed X / E / pos, frame
        ``N-1-(s*N)//T`` taken after every step s with ``(s*N) % T == 0``, the last frame overwritten with the result; they
        come back as ``.chains`` of the returned PlaceHolder (with ``return_state``: as a fifth element)."""
        return self.sample_batches([dict(n_nodes=n_nodes, seed=seed, graph_id_base=graph_id_base, projector=projector,
                                         n_max=n_max, steps=steps, return_state=return_state, keep_chain=keep_chain,
                                         number_chain_steps=number_chain_steps)])[0]

    @torch.no_grad()
    def sample_batches(self, requests):
        """Several independent batches (each a dict of ``sample_batch`` arguments) advanced in turn, one reverse step at a
        time: while the host half of one batch's projector step runs (projector.py: wait for the compacted new edges,
        per-graph validity work, edits back), the device is busy with the next batch's step.  The reference's loop has
        a true dependence inside a batch (step s-1 needs the projected z_s), so the overlap is across batches only.
        Without a projector there is nothing to wait for and the batches simply run one after the other."""
        chains = [self._chain(**r) for r in requests]
        results = [None] * len(chains)
        live = list(range(len(chains)))
        while live:
            for k in list(live):
                try:
                    next(chains[k])
                except StopIteration as done:
                    results[k] = done.value
                    live.remove(k)
        return results

    @staticmethod
    def _collapse_labels(labels, node_mask):
        """collapse (utils.py:169-195) of the int8 state: X = 0 on nodes / -1 on padding; E = class on live pairs, 0 on the
        diagonal of valid nodes (argmax of a zero row), -1 on padded pairs."""
        B, n = node_mask.shape
        E = labels.long()
        eye = torch.eye(n, dtype=torch.bool, device=labels.device)[None]
        valid = node_mask[:, :, None] & node_mask[:, None, :]
        E = torch.where(valid & eye, torch.zeros_like(E), E)
        E = torch.where(valid, E, torch.full_like(E, -1))
        X = torch.where(node_mask, 0, -1).long()
        return X, E

    def _chain(self, n_nodes, seed=0, graph_id_base=0, projector=None, n_max=None, steps=None, return_state=False,
               keep_chain=0, number_chain_steps=50):
        """One reverse chain as a generator: yields wherever the host would have to wait for the device."""
        dev = next(self.model.parameters()).device
        if dev.type != "cuda":
            raise _lib.ScggError("sampling runs on the MI355X only; move the model to the GPU")
        n_nodes = torch.as_tensor(n_nodes, device=dev).long()
        B = n_nodes.numel()
        n = int(n_max) if n_max is not None else int(n_nodes.max())
        node_mask = (torch.arange(n, device=dev)[None, :] < n_nodes[:, None]).contiguous()
        nm = self.noise_model
        gen = torch.Generator(device=dev).manual_seed(seed)
        on_state = hasattr(projector, "project_labels")      # our projector.py classes take the int8 state directly
        z_T = nm.sample_limit_dist(node_mask, generator=gen, with_onehot=projector is not None and not on_state,
                                   rng=Rng(seed=seed, graph_id_base=graph_id_base))
        st = self._alloc_state(node_mask)
        cur = 0
        st["labels"][0].copy_(z_T._labels)
        pos, pos_raw = z_T.pos, z_T._pos_raw     # pos follows the reference's per-step re-centring (pos_after_masks)
        s_values = list(reversed(range(0, self.T, self.faster_sampling)))
        if steps is not None:
            s_values = s_values[:steps]
        proj = projector(z_T, dev) if projector is not None else None
        split = on_state and hasattr(proj, "project_labels_begin")
        chains = None
        if keep_chain > 0:
            assert 0 < number_chain_steps < self.T                                        # :620
            N, kc = number_chain_steps, min(int(keep_chain), B)
            chains = PlaceHolder(X=torch.zeros(N, kc, n, dtype=torch.long, device=dev),
                                 E=torch.zeros(N, kc, n, n, dtype=torch.long, device=dev), y=None,
                                 charges=torch.zeros(N, kc, n, dtype=torch.long, device=dev),
                                 pos=torch.zeros(N, kc, n, 3, device=dev))
        t = self.T
        n_done = 0
        for s in s_values:
            labels_t = st["labels"][cur]
            inp = self._build_inputs(st, labels_t, node_mask, t)
            if self.is3d:
                # z_t.pos went through one mask() per completed step (noise_model.py:446 of the reference).  The radius the
                # chain hands on is overwritten every step (:415-444) and never read by the model, so only the last step
                # runs the radius head.
                inp.pos = nm.pos_after_masks(pos_raw, node_mask, 1 + n_done)
                inp.radius = st["E_in"][..., :1] if s == s_values[-1] else None
            pred = self.model(inp)
            rng = Rng(seed=seed, step=s, graph_id_base=graph_id_base)
            nm.reverse_step_state(pred.E, labels_t, node_mask, st["t_all"][t], st["t_all"][s], rng,
                                  labels_out=st["labels"][cur ^ 1], cands_out=st["cands"][cur ^ 1])
            cur ^= 1
            if proj is not None and split:         # projector.py: compacted new edges to the host, edits scattered back
                proj.project_labels_begin(st["labels"][cur], st["cands"][cur])
                yield                               # the step is enqueued; the host half needs its result
                proj.project_labels_end()
            elif proj is not None and on_state:
                proj.project_labels(st["labels"][cur], st["cands"][cur])
            elif proj is not None:   # any other host-side projector: reference types in, in-place edit of z_s.E
                s_int = torch.full((B, 1), s, dtype=torch.long, device=dev)
                z_s = nm._to_placeholder(st["labels"][cur], st["cands"][cur], node_mask, s_int,
                                         nm.pos_after_masks(pos_raw, node_mask, 2 + n_done), B)
                proj.project(z_s)
                z_s._labels = None
                st["labels"][cur].copy_(NoiseModel._labels_of(z_s, nm.E_classes))
            t = s
            n_done += 1
            if chains is not None and (s * N) % self.T == 0:                               # :666-677
                pos = nm.pos_after_masks(pos_raw, node_mask, 1 + n_done)
                w = N - 1 - (s * N) // self.T
                cX, cE = self._collapse_labels(st["labels"][cur][:kc], node_mask[:kc])
                chains.X[w], chains.E[w], chains.pos[w] = cX, cE, pos[:kc]
        labels = st["labels"][cur]
        pos = nm.pos_after_masks(pos_raw, node_mask, 1 + n_done)
        radius = None
        if self.is3d and s_values and getattr(pred, "radius", None) is not None:          # noise_model.py:415-420
            radius = (labels > 0) * pred.radius.squeeze(-1)
        self.last_radius = radius
        if return_state and chains is None:
            return labels, st["cands"][cur], node_mask, pos
        X, E = self._collapse_labels(labels, node_mask)
        if chains is not None:                                                             # :689-699
            chains.X[-1], chains.E[-1], chains.pos[-1] = X[:kc], E[:kc], pos[:kc]
        if return_state:
            return labels, st["cands"][cur], node_mask, pos, chains
        out = PlaceHolder(X=X, charges=torch.where(node_mask, 0, 1000), E=E, y=torch.zeros(B, 0, device=dev),
                          node_mask=node_mask, pos=pos, radius=radius)           # radius as collapse() leaves it: [B,n,n]
        out.chains = chains
        return out

    # ------------------------------------------------------------------------------------------ driver-level API
    def sample_n_graphs(self, n_nodes_first, n_nodes_rest=(), batch_size: int = 128, adaptive_loader: bool = False,
                        reference_batch_size: int = None, seed: int = 0, projector=None, pipeline: int = 2, **kw):
        """reference sample_n_graphs (diffusion_model_discrete_3d.py:753-843): cut the requested node counts into
        batches by the reference's rule (``plan_sample_batches``) and sample every batch.  The batches are independent,
        which is what ``sample_batches`` needs: ``pipeline`` of them are advanced in turn (useful with a projector).
        ``n_nodes_first`` / ``n_nodes_rest``: what ``nodes_dist.sample_n`` returned for the unsorted first part
        (:767-769) and for the remainder (:816-818).  Global graph ids follow the order of the batches, so the result
        does not depend on ``pipeline``.  Returns one ``sample_batch`` result per batch, in the reference's order."""
        plan = plan_sample_batches(n_nodes_first, n_nodes_rest, batch_size, adaptive_loader, reference_batch_size)
        reqs, base = [], 0
        for k, ns in enumerate(plan):
            reqs.append(dict(n_nodes=ns, seed=seed + k, graph_id_base=base, projector=projector, **kw))
            base += len(ns)
        out = []
        step = max(1, int(pipeline))
        for lo in range(0, len(reqs), step):
            out.extend(self.sample_batches(reqs[lo:lo + step]))
        return out


def effective_batch_size(max_size, reference_batch_size, reference_size=20, sampling=False):
    """datasets/adaptive_loader.py:18-22."""
    x = reference_batch_size * (reference_size / max_size) ** 2
    return math.floor(1.8 * x) if sampling else math.floor(x)


def plan_sample_batches(n_nodes_first, n_nodes_rest=(), batch_size: int = 128, adaptive_loader: bool = False,
                        reference_batch_size: int = None):
    """The batching rule of the reference's sample_n_graphs, as lists of node counts (one list per sample_batch call).

    First part (:767-813, the graphs that are visualised; sizes in the order drawn): a graph joins the current batch while
    ``ebs > len(batch)`` with ``ebs = int(1.8 * batch_size)`` (:782-785; no backward pass at sampling time), or
    ``effective_batch_size(max size so far incl. this graph, reference_batch_size, sampling=True)`` with the adaptive
    loader (:776-781).  Remainder (:816-842): sorted by decreasing size when adaptive (:820-822); the limit there is
    the FLOAT ``1.8 * batch_size`` (:832-833, no int()), and the running maximum is updated before the test (:827)."""
    def as_list(x):
        return [int(v) for v in (x.tolist() if hasattr(x, "tolist") else x)]
    first, rest = as_list(n_nodes_first), as_list(n_nodes_rest)
    if adaptive_loader and reference_batch_size is None:
        raise ValueError("adaptive_loader needs reference_batch_size (cfg.train.reference_batch_size)")
    plan = []
    if first:
        cur, cur_max = [], 0
        for n in first:
            pot_max = max(cur_max, n)
            ebs = effective_batch_size(pot_max, reference_batch_size, sampling=True) if adaptive_loader else int(1.8 * batch_size)
            if ebs > len(cur) or len(cur) == 0:
                cur.append(n)
                cur_max = pot_max
            else:
                plan.append(cur)
                cur, cur_max = [n], n
        plan.append(cur)
    if rest:
        if adaptive_loader:
            rest = sorted(rest, reverse=True)
        cur, max_size = [], 0
        for n in rest:
            max_size = max(max_size, n)
            ebs = effective_batch_size(max_size, reference_batch_size, sampling=True) if adaptive_loader else 1.8 * batch_size
            if ebs > len(cur) or len(cur) == 0:
                cur.append(n)
            else:
                plan.append(cur)
                cur, max_size = [n], n
        plan.append(cur)
    return plan
