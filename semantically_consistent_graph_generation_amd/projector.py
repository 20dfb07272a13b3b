"""Host-side validity projectors of the reverse loop.

Mirrors /root/reference/src/ConStruct/projector/projector_3d_utils.py: ``CoWProjector`` (:213-387),
``CoWVanillaProjector`` (:464-598), ``ATMTreeProjector`` (:662-840), ``ATMVanillaTreeProjector`` (:917-1052),
``TreeProjector`` (:157-164 over ``AbstractProjector`` :16-147) and ``BaseValidProjector`` (:168-183): same
constructor arguments, same ``project(z_s)`` (in-place edit of ``z_s.E``), same ``blocked_edges`` /
``nx_graphs_list`` / ``counter`` attributes, same use of the GLOBAL ``np.random`` stream (one
``np.random.shuffle`` of an ``[m, 2]`` array per graph that falls back to one-by-one insertion), so a run under
``np.random.seed(k)`` gives the reference's result bit for bit (tests/golden/f12_projector_*.npz).

What is different is the machinery:
  * no networkx: every sample is a ``LabelGraph`` (adjacency dict + per-node label multiset) and the reference's
    line-graph test (``is_line_graph_valid`` :431-461: every two edges that share a node must carry classes that
    are equal or adjacent in the 13x13 / 4x4 table) is a bitmask test per node;
  * ``project_labels(labels, cands)``: the state route of ``sampling.Sampler`` keeps the graph state as int8 labels
    on the device, so the projector takes that tensor directly: the new edges (label differs from the stored
    z_t, upper triangle, row-major = the order of ``Tensor.nonzero`` in the reference) are compacted on the
    device, one small copy brings (b, i, j, label, candidates) to the host, and the host's edits go back as one
    scatter.  The one-hot E and the int64 extra_info are never built.

The reference's behaviour is kept where it is surprising (each is covered by the golden trajectories):
  * one-by-one insertion shuffles the edges but not their labels / candidates (:330-331), so an edge can enter the
    graph with another new edge's label;
  * the all-at-once attempt is evaluated twice (:318 and again in the condition :327), the second time re-adding
    every edge with its original label on top of the first pass's fixes;
  * the vanilla variants do not write the graph's label back, so graph and tensor labels can differ;
  * a candidate label stays on the edge when no candidate repairs it and later edges of the same attempt are
    judged with it present (:412-426).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

# projector_3d_utils.py:193-205 (TopCoW segment classes 1..13) and :645-648 (ATM generations 1..4)
COW_ADJACENCY = (
    (0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0),
    (1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0),
    (1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0),
    (0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0),
    (0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0),
    (0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 1, 0),
    (0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0),
    (0, 1, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0),
    (0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0),
    (0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1),
    (0, 0, 0, 1, 1, 0, 0, 1, 0, 1, 0, 1, 1),
    (0, 0, 0, 0, 0, 1, 1, 0, 1, 1, 1, 0, 1),
    (0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0),
)
ATM_ADJACENCY = ((0, 1, 0, 0), (1, 0, 1, 0), (0, 1, 0, 1), (0, 0, 1, 0))


def _allowed_masks(table):
    """allowed[c] = bit set of the classes that may sit beside class c at a node (c itself included: the reference
    clears the diagonal, :457).  Bit c <-> class c, classes are 1-based, bit 0 unused."""
    rows = [0]
    for a, row in enumerate(table):
        assert all(row[b] == table[b][a] for b in range(len(table))), "Not symmetric"        # check_symmetric :207
        rows.append((1 << (a + 1)) | sum(1 << (b + 1) for b, v in enumerate(row) if v))
    return tuple(rows)


class LabelGraph:
    """One sample's graph on the host: adj[u][v] = class of edge {u,v}; cnt[u][c] = edges of class c at node u."""

    __slots__ = ("n", "adj", "cnt", "n_edges", "pending")

    def __init__(self, n: int):
        self.n = n
        self.adj = [dict() for _ in range(n)]
        self.cnt = [dict() for _ in range(n)]
        self.n_edges = 0
        self.pending = None            # the reference's nx_graph.graph['all_possible_new_edges_to_add']

    def copy(self) -> "LabelGraph":
        g = LabelGraph.__new__(LabelGraph)
        g.n, g.n_edges, g.pending = self.n, self.n_edges, self.pending
        g.adj = [d.copy() for d in self.adj]
        g.cnt = [d.copy() for d in self.cnt]
        return g

    def _count(self, u, c, d):
        k = self.cnt[u].get(c, 0) + d
        if k:
            self.cnt[u][c] = k
        else:
            del self.cnt[u][c]

    def add_edge(self, u: int, v: int, label=None):
        """nx.Graph.add_edge: an existing edge keeps its place and takes the new label."""
        old = self.adj[u].get(v, self)
        if old is self:
            self.n_edges += 1
        else:
            self._count(u, old, -1)
            self._count(v, old, -1)
        self.adj[u][v] = label
        self.adj[v][u] = label
        self._count(u, label, 1)
        self._count(v, label, 1)

    def remove_edge(self, u: int, v: int):
        c = self.adj[u].pop(v)
        del self.adj[v][u]
        self._count(u, c, -1)
        self._count(v, c, -1)
        self.n_edges -= 1

    def label(self, u: int, v: int):
        return self.adj[u][v]

    def number_of_nodes(self) -> int:
        return self.n

    def number_of_edges(self) -> int:
        return self.n_edges

    def edges(self):
        return [(u, v, c) for u in range(self.n) for v, c in self.adj[u].items() if u < v]

    def to_numpy_array(self) -> np.ndarray:
        a = np.zeros((self.n, self.n))
        for u, v, _ in self.edges():
            a[u, v] = a[v, u] = 1.0
        return a


def line_graph_valid(g: LabelGraph, allowed) -> bool:
    """is_line_graph_valid (:431-461, :881-912) / is_valid_graph (:605-637, :1060-1092): the line graph joins two
    edges iff they share a node, so the set of class pairs it marks is the union over nodes of the pairs of
    classes present at that node; invalid iff some pair of different classes is not in the table."""
    n_cls = len(allowed)
    for cnt in g.cnt:
        if len(cnt) < 2:
            continue
        present = 0
        for c in cnt:
            if not 0 < c < n_cls:          # the reference indexes table[c-1, ...] here and fails the same way
                raise IndexError(f"edge class {c} outside the validity table (1..{n_cls - 1})")
            present |= 1 << c
        for c in cnt:
            if present & ~allowed[c]:
                return False
    return True


def has_no_cycles(g: LabelGraph) -> bool:
    """has_no_cycles (:150-154): fewer edges than nodes and nx.is_forest."""
    if g.n_edges >= g.n:
        return False
    parent = list(range(g.n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for u, v, _ in g.edges():
        ru, rv = find(u), find(v)
        if ru == rv:
            return False
        parent[ru] = rv
    return True


class _Fixer:
    """check_validity_and_fix_if_needed (:395-428 with the rebuttal counter, :847-879 without)."""

    def __init__(self, table, counted: bool):
        self.allowed = _allowed_masks(table)
        self.valid_adjacency = torch.as_tensor(table)
        self.counted = counted
        self._count = 0
        self._owner = None             # set by a projector whose engine is native: the count lives in libscgg.so

    @property
    def counter(self):
        if self._owner is not None and self._owner._native is not None:
            return int(_lib.lib().scgg_projector_counter(self._owner._native))
        return self._count

    @counter.setter
    def counter(self, v):
        self._count = v

    def is_line_graph_valid(self, g: LabelGraph) -> bool:
        return line_graph_valid(g, self.allowed)

    is_valid_graph = is_line_graph_valid

    def check_validity_and_fix_if_needed(self, g: LabelGraph) -> bool:
        if g.pending is None and g.n_edges == 0:
            return True
        all_ok = True
        for (u, v), label, cx in g.pending:
            g.add_edge(u, v, label)
            ok = True
            if not line_graph_valid(g, self.allowed):
                ok = False
                if self.counted:
                    self.counter -= 1
                for cand in cx:
                    if self.counted:
                        self.counter += 1
                    cand = int(cand)
                    if cand == 0:          # class 0 would delete the edge
                        continue
                    g.remove_edge(u, v)
                    g.add_edge(u, v, cand)
                    if line_graph_valid(g, self.allowed):
                        ok = True
                        break
            all_ok = all_ok and ok
        return all_ok


def _adj_of(z, binary: bool) -> torch.Tensor:
    """get_multi_class_adj_matrix (:185-187) / get_adj_matrix (:10-13)."""
    a = torch.argmax(z.E, dim=3)
    return (a != 0).long() if binary else a


class _Projector:
    """Shared driver.  ``engine="native"`` (default): the per-graph work runs in libscgg.so (csrc/projector_host.hip,
    scgg_projector_*), the new edges of a device state are compacted by scgg_new_edges and the edits go back through
    scgg_apply_edge_edits.  ``engine="python"``: the same algorithm on the ``LabelGraph`` classes of this module (the
    readable restatement the tests hold the native engine against)."""

    binary = False           # TreeProjector compares 0/1 adjacency (:59-62), the others the class labels
    uses_cands = False
    can_block_edges = True
    kind = None              # SCGG_PROJ_* of include/scgg.h

    def __init__(self, z_t, device=None, engine="native", rng=None):
        if engine not in ("native", "python"):
            raise ValueError(f"engine {engine!r}")
        self.engine = engine
        self.rng = rng           # None: numpy's global stream, as the reference; else a np.random.RandomState of this chain
        self.batch_size = z_t.X.shape[0]
        self.device = torch.device(device) if device is not None else z_t.X.device
        if getattr(z_t, "E", None) is not None:
            self.z_t_adj = _adj_of(z_t, self.binary)
        else:                                  # our sampler's z_T without the one-hot: int8 labels, -1 = zero row
            self.z_t_adj = z_t._labels.clone()
        assert bool((self.z_t_adj <= 0).all()), "the limit distribution has no edges"            # :234 / :37
        self._n_nodes = [int(k) for k in z_t.node_mask.sum(dim=1).tolist()]
        self._native = None
        self._dev_buf = None
        self._in_flight = None
        if engine == "native":
            L = _lib.lib()
            nn_ = np.asarray(self._n_nodes, dtype=np.int32)
            h = C.c_void_p()
            _lib.check(L.scgg_projector_create(self.kind, self.batch_size, nn_.ctypes.data, C.byref(h)), "projector_create")
            self._native = h
            self._fb_graph = np.empty(max(self.batch_size, 1), dtype=np.int32)
            self._fb_count = np.empty(max(self.batch_size, 1), dtype=np.int32)
        else:
            self._graphs = [LabelGraph(k) for k in self._n_nodes]
            for g in self._graphs:
                assert self._valid_empty(g)
            self._blocked = {b: {(i, j): False for i in range(k) for j in range(i + 1, k)}
                             for b, k in enumerate(self._n_nodes)}

    def __del__(self):
        h, self._native = getattr(self, "_native", None), None
        if h:
            try:
                _lib.lib().scgg_projector_destroy(h)
            except Exception:          # interpreter shutdown: the library object may be gone already
                pass

    # ---- the reference's attributes ---------------------------------------------------------------------------
    def _export(self, b):
        k = self._n_nodes[b]
        lab, blk = np.zeros((k, k), dtype=np.int8), np.zeros((k, k), dtype=np.uint8)
        _lib.check(_lib.lib().scgg_projector_export(self._native, b, lab.ctypes.data, blk.ctypes.data), "projector_export")
        return lab, blk

    @property
    def nx_graphs_list(self):
        """The stored graph of every sample (the reference keeps nx.Graph objects; here ``LabelGraph``)."""
        if self._native is None:
            return self._graphs
        out = []
        for b, k in enumerate(self._n_nodes):
            lab, _ = self._export(b)
            g = LabelGraph(k)
            for i, j in zip(*np.nonzero(np.triu(lab, 1))):
                g.add_edge(int(i), int(j), None if self.binary else int(lab[i, j]))
            out.append(g)
        return out

    @property
    def blocked_edges(self):
        if self._native is None:
            return self._blocked
        out = {}
        for b, k in enumerate(self._n_nodes):
            _, blk = self._export(b)
            out[b] = {(i, j): bool(blk[i, j]) for i in range(k) for j in range(i + 1, k)}
        return out

    # ---- per-class pieces (python engine) -----------------------------------------------------------------------
    def _valid_empty(self, g) -> bool:
        return True

    def _project_graph(self, b, edges, attrs, cands, edits):
        raise NotImplementedError

    # ---- reference interface --------------------------------------------------------------------------------
    def project(self, z_s):
        """project(z_s) of the reference: one-hot ``z_s.E`` (and int ``z_s.extra_info``) in, ``z_s.E`` edited in place."""
        z_s_adj = _adj_of(z_s, self.binary)
        diff = z_s_adj - self.z_t_adj.clamp(min=0).to(z_s_adj.dtype)
        assert bool((diff >= 0).all()), "Once unmasked, the edges cannot be removed."
        new = diff.nonzero(as_tuple=False)
        new = new[new[:, 1] < new[:, 2]]
        rec = [new, z_s_adj[new[:, 0], new[:, 1], new[:, 2]][:, None]]
        if self.uses_cands:
            rec.append(z_s.extra_info[new[:, 0], new[:, 1], new[:, 2]].long())
        edits = self._run(torch.cat(rec, dim=1).to(torch.int32).cpu().numpy())
        if len(edits):
            e = torch.from_numpy(edits).long().to(z_s.E.device)
            rows = torch.nn.functional.one_hot(e[:, 3], z_s.E.shape[-1]).to(z_s.E.dtype)
            z_s.E[e[:, 0], e[:, 1], e[:, 2]] = rows
            z_s.E[e[:, 0], e[:, 2], e[:, 1]] = rows
            # our sampler's PlaceHolder also carries the int8 state the next step reads (noise_model._labels_of): keep it
            # in step with the edited one-hot, or the next reverse step would start from the pre-projection graph
            lab = getattr(z_s, "_labels", None)
            if lab is not None:
                v = e[:, 3].to(lab.dtype)
                lab[e[:, 0], e[:, 1], e[:, 2]] = v
                lab[e[:, 0], e[:, 2], e[:, 1]] = v
        self.z_t_adj = _adj_of(z_s, self.binary)

    # ---- state route ----------------------------------------------------------------------------------------
    def project_labels(self, labels: torch.Tensor, cands: torch.Tensor = None):
        """Same step on the int8 state of ``Sampler.sample_batch``: labels [B,n,n] (-1 on padding and diagonal; never a
        new edge), cands [B,n,n,k] int8.  Edits ``labels`` in place; returns the number of edited pairs."""
        if self.uses_cands and cands is None:
            raise _lib.ScggError(f"{type(self).__name__}.project_labels needs the top-k candidates")
        if labels.dtype != torch.int8 or labels.dim() != 3 or not labels.is_contiguous():
            raise _lib.ScggError("project_labels: labels must be a contiguous int8 [B,n,n] tensor")
        if labels.is_cuda:
            return self._project_labels_device(labels, cands if self.uses_cands else None)
        prev = self.z_t_adj.to(labels.dtype)                     # host tensors: same step with torch ops
        if self.binary:
            changed, lost = (labels > 0) & ~(prev > 0), (prev > 0) & ~(labels > 0)
        else:
            changed, lost = labels > prev, labels < prev
        assert not bool(lost.any()), "Once unmasked, the edges cannot be removed."
        new = changed.triu(1).nonzero(as_tuple=False)
        rec = [new, labels[new[:, 0], new[:, 1], new[:, 2]].long()[:, None]]
        if self.uses_cands:
            rec.append(cands[new[:, 0], new[:, 1], new[:, 2]].long())
        edits = self._run(torch.cat(rec, dim=1).to(torch.int32).numpy())
        if len(edits):
            e = torch.from_numpy(edits).long()
            v = e[:, 3].to(labels.dtype)
            labels[e[:, 0], e[:, 1], e[:, 2]] = v
            labels[e[:, 0], e[:, 2], e[:, 1]] = v
        self.z_t_adj = labels.clone()
        return len(edits)

    def _project_labels_device(self, labels, cands):
        self.project_labels_begin(labels, cands)
        return self.project_labels_end()

    HOST_CHUNK = 8192        # records copied to the host together with the count; a step with more takes a second copy

    def project_labels_begin(self, labels, cands=None):
        """Device half of a step, enqueued on the current stream without waiting: compaction of the new edges, then an
        asynchronous copy of the count and the first HOST_CHUNK records into pinned memory, then an event.
        ``project_labels_end`` waits for that event only, so work enqueued behind it (another batch's step) keeps the
        device busy during the host half."""
        if self.uses_cands and cands is None:
            raise _lib.ScggError(f"{type(self).__name__}.project_labels needs the top-k candidates")
        if not labels.is_cuda or labels.dtype != torch.int8 or labels.dim() != 3 or not labels.is_contiguous():
            raise _lib.ScggError("project_labels_begin: labels must be a contiguous int8 [B,n,n] tensor on the GPU")
        cands = cands if self.uses_cands else None
        L = _lib.lib()
        B, n, _ = labels.shape
        k = cands.shape[-1] if cands is not None else 0
        if cands is not None and (cands.dtype != torch.int8 or not cands.is_contiguous() or cands.shape[:3] != labels.shape):
            raise _lib.ScggError("project_labels: cands must be a contiguous int8 [B,n,n,k] tensor")
        buf = self._dev_buf
        if buf is None or buf["key"] != (B, n, k, labels.device):
            cap = B * (n * (n - 1) // 2)                         # every pair of every graph new at once
            chunk = max(1, min(cap, self.HOST_CHUNK))
            dev = labels.device
            buf = self._dev_buf = dict(key=(B, n, k, dev), cap=cap, chunk=chunk,
                                       meta=torch.zeros(B + 2, dtype=torch.int32, device=dev),
                                       rec=torch.empty(max(cap, 1), 4 + k, dtype=torch.int32, device=dev),
                                       prev=self.z_t_adj.to(device=dev, dtype=torch.int8).contiguous(),
                                       meta_host=torch.zeros(2, dtype=torch.int32).pin_memory(),
                                       rec_host=torch.empty(chunk, 4 + k, dtype=torch.int32).pin_memory(),
                                       edits_host=torch.empty(chunk, 4, dtype=torch.int32).pin_memory(),
                                       edits=torch.empty(chunk, 4, dtype=torch.int32, device=dev),
                                       event=torch.cuda.Event())
        meta, rec = buf["meta"], buf["rec"]
        status_ptr = C.c_void_p(meta.data_ptr() + 4 * (B + 1))
        _lib.check(L.scgg_new_edges(_lib.ptr(labels), _lib.ptr(buf["prev"]), _lib.ptr(cands), B, n, k, int(self.binary),
                                    _lib.ptr(meta), _lib.ptr(rec), buf["cap"], status_ptr, _lib.stream_ptr()), "new_edges")
        buf["meta_host"].copy_(meta[B:], non_blocking=True)
        buf["rec_host"].copy_(rec[:buf["chunk"]], non_blocking=True)
        buf["event"].record()
        self._in_flight = labels

    def project_labels_end(self):
        """Host half: wait for the step's records, decide per graph, enqueue the edits and the new z_t."""
        labels, self._in_flight = self._in_flight, None
        if labels is None:
            raise _lib.ScggError("project_labels_end without project_labels_begin")
        buf = self._dev_buf
        B, n, k, _ = buf["key"]
        buf["event"].synchronize()                               # this batch's step only, not the whole stream
        m, status = buf["meta_host"].tolist()
        assert status == 0, "Once unmasked, the edges cannot be removed."
        if m <= buf["chunk"]:
            rec = buf["rec_host"][:m].numpy()
        else:                                                    # rare: a step with more new edges than the pinned chunk
            rec = buf["rec"][:m].cpu().numpy()
        edits = self._run(rec)
        ne = len(edits)
        if ne:
            if ne > buf["edits_host"].shape[0]:
                buf["edits_host"] = torch.empty(ne, 4, dtype=torch.int32).pin_memory()
                buf["edits"] = torch.empty(ne, 4, dtype=torch.int32, device=labels.device)
            buf["edits_host"][:ne].numpy()[...] = edits
            buf["edits"][:ne].copy_(buf["edits_host"][:ne], non_blocking=True)
            _lib.check(_lib.lib().scgg_apply_edge_edits(_lib.ptr(labels), _lib.ptr(buf["edits"]), ne, B, n, _lib.stream_ptr()),
                       "apply_edge_edits")
        buf["prev"].copy_(labels)
        self.z_t_adj = buf["prev"]
        return ne

    # ---- host part ------------------------------------------------------------------------------------------
    def _run(self, rec: np.ndarray) -> np.ndarray:
        """rec int32 [m, 4+k]: (graph, i, j, label, candidates...) sorted by (graph, i, j), i < j.
        Returns the step's edits int32 [e, 4] = (graph, i, j, class), one per pair (the reference's last write)."""
        rec = np.ascontiguousarray(rec, dtype=np.int32)
        m, k = rec.shape[0], rec.shape[1] - 4
        if self._native is not None:
            L = _lib.lib()
            n_fb = C.c_int32(0)
            _lib.check(L.scgg_projector_begin(self._native, m, rec.ctypes.data, k, self._fb_graph.ctypes.data,
                                              self._fb_count.ctypes.data, C.byref(n_fb)), "projector_begin")
            counts = self._fb_count[:n_fb.value].astype(np.int64)
            ends = np.cumsum(counts)
            perm = (np.arange(int(ends[-1]) if len(ends) else 0) - np.repeat(ends - counts, counts)).astype(np.int32)
            shuffle = self._shuffle
            for lo, hi in zip((ends - counts)[counts > 1].tolist(), ends[counts > 1].tolist()):
                shuffle(perm[lo:hi])                 # np.random.shuffle of the reference, in graph order (:330); a
                                                     # 1-element shuffle draws nothing from the stream and is skipped
            out = np.empty((max(m, 1), 4), dtype=np.int32)
            ne = C.c_int64(0)
            _lib.check(L.scgg_projector_finish(self._native, perm.ctypes.data, perm.size, out.ctypes.data, out.shape[0],
                                               C.byref(ne)), "projector_finish")
            return out[:ne.value]
        edits = []
        if m:
            starts = np.searchsorted(rec[:, 0], np.arange(self.batch_size + 1))
            for b in range(self.batch_size):
                lo, hi = starts[b], starts[b + 1]
                if hi > lo:
                    self._project_graph(b, rec[lo:hi, 1:3], rec[lo:hi, 3], rec[lo:hi, 4:] if k else None, edits)
        last = {}
        for b, u, v, c in edits:
            last[(int(b), int(u), int(v))] = int(c)
        return np.asarray([(b, u, v, c) for (b, u, v), c in last.items()], dtype=np.int32).reshape(-1, 4)

    def _shuffle(self, x):
        (np.random if self.rng is None else self.rng).shuffle(x)

    def _drop_blocked(self, b, edges, edits):
        """:282-301: an edge blocked in an earlier step is deleted again at once; returns the indices kept."""
        blocked = self._blocked[b]
        keep = []
        for idx, (u, v) in enumerate(edges):
            if blocked[(int(u), int(v))]:
                edits.append((b, u, v, 0))
            else:
                keep.append(idx)
        return keep


class _FixingProjector(_Projector):
    """project() of CoWProjector (:256-377) and ATMTreeProjector (:697-825)."""

    uses_cands = True

    def _try(self, g) -> bool:
        raise NotImplementedError

    def _valid_empty(self, g):
        return self._try(g)

    def try_fix_and_validate_fn(self, g) -> bool:
        return self._try(g)

    def _project_graph(self, b, edges, attrs, cands, edits):
        g = self._graphs[b]
        old = g.copy()
        keep = self._drop_blocked(b, edges, edits)
        if not keep:
            return
        edges, attrs, cands = np.array(edges[keep]), attrs[keep], cands[keep]
        if len(edges) > 1:
            g.pending = [((int(u), int(v)), int(label), cx) for (u, v), label, cx in zip(edges, attrs, cands)]
            if self._try(g):
                for u, v in edges:
                    edits.append((b, u, v, g.label(int(u), int(v))))
        if len(edges) == 1 or not self._try(g):
            g = old
            self._shuffle(edges)                   # edges only: labels and candidates keep their order (:330-331)
            for (u, v), label, cx in zip(edges, attrs, cands):
                u, v = int(u), int(v)
                old = g.copy()
                g.pending = [((u, v), int(label), cx)]
                if not self._try(g):
                    g = old
                    edits.append((b, u, v, 0))
                    self._blocked[b][(u, v)] = True
                else:
                    edits.append((b, u, v, g.label(u, v)))
            self._graphs[b] = g



class _VanillaProjector(_Projector):
    """project() of CoWVanillaProjector (:499-590), ATMVanillaTreeProjector (:952-1043) and AbstractProjector (:57-147)."""

    def valid_graph_fn(self, g) -> bool:
        raise NotImplementedError

    def _valid_empty(self, g):
        return self.valid_graph_fn(g)

    def _project_graph(self, b, edges, attrs, cands, edits):
        g = self._graphs[b]
        old = g.copy()
        keep = self._drop_blocked(b, edges, edits)
        if not keep:
            return
        edges, attrs = np.array(edges[keep]), attrs[keep]
        if len(edges) > 1:
            for (u, v), label in zip(edges, attrs):
                g.add_edge(int(u), int(v), self._label(label))
        if not self.valid_graph_fn(g) or len(edges) == 1:
            g = old
            self._shuffle(edges)
            for (u, v), label in zip(edges, attrs):
                u, v = int(u), int(v)
                old = g.copy()
                g.add_edge(u, v, self._label(label))
                if not self.valid_graph_fn(g):
                    g = old
                    edits.append((b, u, v, 0))
                    self._blocked[b][(u, v)] = True
            self._graphs[b] = g

    @staticmethod
    def _label(label):
        return int(label)


class CoWProjector(_FixingProjector):
    """:213-387 -- TopCoW class adjacency with repair from the top-k candidates."""

    kind = 0

    def __init__(self, z_t, device=None, engine="native", rng=None):
        self.top_cow_validity_check_and_fix_obj = _Fixer(COW_ADJACENCY, counted=True)
        self.top_cow_validity_check_and_fix_obj._owner = self
        super().__init__(z_t, device, engine, rng)

    def _try(self, g):
        return self.top_cow_validity_check_and_fix_obj.check_validity_and_fix_if_needed(g)


class ATMTreeProjector(_FixingProjector):
    """:662-840 -- ATM generation adjacency with repair, and no cycles."""

    kind = 2

    def __init__(self, z_t, device=None, engine="native", rng=None):
        self.atm_validity_check_and_fix_obj = _Fixer(ATM_ADJACENCY, counted=False)
        self.atm_validity_check_and_fix_obj._owner = self
        super().__init__(z_t, device, engine, rng)

    def _try(self, g):
        return self.atm_validity_check_and_fix_obj.check_validity_and_fix_if_needed(g) and has_no_cycles(g)

    def valid_graph_fn_lite(self, g) -> bool:
        return g.pending is None and self.atm_validity_check_and_fix_obj.is_line_graph_valid(g)


class CoWVanillaProjector(_VanillaProjector):
    """:464-598 -- TopCoW class adjacency, violating edges deleted."""

    kind = 1

    def __init__(self, z_t, device=None, engine="native", rng=None):
        self.top_cow_validity_check = _Fixer(COW_ADJACENCY, counted=False)
        self.top_cow_validity_check._owner = self
        super().__init__(z_t, device, engine, rng)

    def valid_graph_fn(self, g):
        return self.top_cow_validity_check.is_valid_graph(g)


class ATMVanillaTreeProjector(_VanillaProjector):
    """:917-1052 -- ATM generation adjacency and no cycles, violating edges deleted."""

    kind = 3

    def __init__(self, z_t, device=None, engine="native", rng=None):
        self.atm_validity_check = _Fixer(ATM_ADJACENCY, counted=False)
        self.atm_validity_check._owner = self
        super().__init__(z_t, device, engine, rng)

    def valid_graph_fn(self, g):
        return self.atm_validity_check.is_valid_graph(g) and has_no_cycles(g)


class TreeProjector(_VanillaProjector):
    """:157-164 over AbstractProjector :16-147 -- unlabelled forest constraint on the 0/1 adjacency."""

    binary = True
    kind = 4

    def valid_graph_fn(self, g):
        return has_no_cycles(g)

    @staticmethod
    def _label(label):
        return None


class BaseValidProjector:
    """:168-183 -- checks only that no edge disappeared; stateless (its z_t_adj is never advanced in the reference
    either, so the check is against the empty limit graph)."""

    can_block_edges = True

    def __init__(self, z_t, device=None):
        self.batch_size = z_t.X.shape[0]

    def project(self, z_s):
        pass

    def project_labels(self, labels, cands=None):
        return 0

    def valid_graph_fn(self, g) -> bool:
        return True


def projector_for(rev_proj):
    """cfg.model.rev_proj -> class (diffusion_model_discrete_3d.py:636-650); falsy -> None (the loop skips it, :660)."""
    if not rev_proj:
        return None
    return {"cow_proj": CoWProjector, "cow_proj_vanilla": CoWVanillaProjector, "atm_proj": ATMTreeProjector,
            "atm_proj_vanilla": ATMVanillaTreeProjector}.get(rev_proj, BaseValidProjector)
